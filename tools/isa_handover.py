#!/usr/bin/env python3
"""tools/isa_handover.py [libyacrd_hip.so] — the ISA the device-side hand-overs rest on (VERDICT r4
item 4).  one_batch_kernel, finish_compact_kernel and scan_compact_kernel pass verdicts, regions, scan words and
counters between wavefronts on different XCDs INSIDE one launch with relaxed agent-scope accesses and s_waitcnt — no
release / acquire (a fence at agent scope writes back / invalidates a whole L2 per use on gfx950).  That is a data
race in the HSA memory model that works because of how gfx950 executes what the compiler emits.  This script
disassembles the gfx950 code objects of the built library and checks that what it emits is still that:
  1. agent-scope atomic stores / loads are global_store / global_load with the sc1 bit (write-through to / read from
     memory past the XCD's L2), system-scope ones sc0 sc1;
  2. in one_batch_kernel every arrival (global_atomic_add_x2 ... sc0, returning) has an `s_waitcnt vmcnt(0)` in front of
     it with no store in between: the wavefront's verdicts are acknowledged before it counts as arrived;
  3. in finish_compact_kernel the slab's counter atomics (returning: sc0) are waited for (s_waitcnt vmcnt(0)) before the
     next agent-scope store (the scan word that publishes the slab);
  4. none of the three kernels holds a cache write-back / invalidate (buffer_wbl2 / buffer_inv): nobody has turned the
     hand-overs into fences without looking at what they cost.
Exit status 0 and a summary on stdout, or an AssertionError naming the line.  tests/test_isa_handover.py runs it on CPU."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    """The device code objects of an HIP shared library: entries of its clang offload bundles for `arch`."""
    d = open(path, "rb").read()
    out, j = [], d.find(MAGIC)
    while j != -1:
        n = struct.unpack_from("<Q", d, j + len(MAGIC))[0]
        p = j + len(MAGIC) + 8
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + ts].decode()
            p += ts
            if arch in triple and size:
                out.append(d[j + off:j + off + size])
        j = d.find(MAGIC, j + 1)
    return out


class Ins(str):
    """An instruction's text, with its address and — for a branch — its target address."""
    addr = 0
    target = None


def disassemble(path):
    """{kernel symbol: [Ins, ...]} over every gfx950 code object of the library."""
    funcs = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur, start = None, 0
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
            if m:
                cur, start = funcs.setdefault(m.group(2), []), int(m.group(1), 16)
                continue
            t = line.strip()
            if cur is None or not t or t.startswith(("/", ";")):
                continue
            m = re.match(r"^(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<[^>+]+(?:\+0x([0-9a-f]+))?>)?\s*$", t)
            ins = Ins(m.group(1).strip() if m else t)
            if m:
                ins.addr = int(m.group(2), 16)
                if ins.startswith(("s_cbranch", "s_branch")):
                    ins.target = start + (int(m.group(3), 16) if m.group(3) else 0)
            cur.append(ins)
    return funcs


def dominated_by_wait(ins, i, what):
    """Every path to ins[i] passes an `s_waitcnt vmcnt(0)` with no store / atomic behind it: the nearest one in front of
    it in layout order, nothing but conditional forward skips in between, and no branch from outside into that stretch."""
    j = i - 1
    while j >= 0 and not ins[j].startswith("s_waitcnt vmcnt(0)"):
        assert not is_store(ins[j]) and not ins[j].startswith("global_atomic"), \
            "%s: `%s` between the last s_waitcnt vmcnt(0) and `%s`" % (what, ins[j], ins[i])
        assert not ins[j].startswith(("s_branch", "s_setpc", "s_endpgm")), \
            "%s: `%s` is not reached by falling through from an s_waitcnt vmcnt(0)" % (what, ins[i])
        j -= 1
    assert j >= 0, "%s: no s_waitcnt vmcnt(0) in front of `%s`" % (what, ins[i])
    lo, hi = ins[j].addr, ins[i].addr
    for k, t in enumerate(ins):
        if t.target is not None and lo < t.target <= hi:
            assert j <= k < i, "%s: `%s` at %#x jumps in between the wait and `%s`" % (what, t, t.addr, ins[i])
    return i - j


def kernels(funcs, name):
    """every instantiation of the kernel `name` (scan_compact_kernel is a template over the reads per thread)"""
    hits = sorted(k for k in funcs if name in k and not k.endswith(".kd"))
    assert hits, name
    return [(k, funcs[k]) for k in hits]


def kernel(funcs, name):
    hits = kernels(funcs, name)
    assert len(hits) == 1, (name, [k for k, _ in hits])
    return hits[0][1]


def is_store(t):
    return t.startswith(("global_store", "buffer_store", "flat_store", "scratch_store"))


def check(path):
    funcs = disassemble(path)
    report = {}
    for name, sym, ins in [(n, k, i) for n in ("one_batch_kernel", "finish_compact_kernel", "scan_compact_kernel") for k, i in kernels(funcs, n)]:
        tmpl = re.search(r"ILi(\d+)E", sym)
        name = name + ("<%s>" % tmpl.group(1) if tmpl else "")
        fences = [t for t in ins if t.startswith(("buffer_wbl2", "buffer_inv"))]
        assert not fences, "%s holds cache maintenance (%s): the hand-overs were measured WITHOUT fences" % (name, fences[:3])
        st = [t for t in ins if t.startswith("global_store")]
        ld = [t for t in ins if t.startswith("global_load")]
        st_agent = [t for t in st if re.search(r"\bsc1\b", t)]
        ld_agent = [t for t in ld if re.search(r"\bsc1\b", t)]
        report[name] = {"instructions": len(ins), "stores": len(st), "stores_sc1": len(st_agent), "loads": len(ld),
                        "loads_sc1": len(ld_agent)}
    ob = kernel(funcs, "one_batch_kernel")
    # 1. the verdicts (closed form: dwordx2 + the count word), the sorted reads' regions and counts, the scan words: agent scope
    assert report["one_batch_kernel"]["stores_sc1"] >= 8 and report["one_batch_kernel"]["loads_sc1"] >= 10, report["one_batch_kernel"]
    pairs = sum(1 for i, t in enumerate(ob) if t.startswith("global_store_dwordx2") and "sc1" in t and
                any(u.startswith("global_store_dword ") and "sc1" in u for u in ob[i + 1:i + 24]))
    assert pairs >= 2, "the closed form's (a, b) store and its count word are no longer both sc1 stores"
    assert any("sc0 sc1" in t for t in ob if t.startswith("global_store")), "the give-up flag for the host is no system-scope store"
    # 2. arrivals
    arrivals = [i for i, t in enumerate(ob) if t.startswith("global_atomic_add_x2") and "sc0" in t]
    assert arrivals, "no returning 64-bit arrival atomic in one_batch_kernel"
    for i in arrivals:
        dominated_by_wait(ob, i, "one_batch_kernel")
    report["one_batch_kernel"]["arrivals_checked"] = len(arrivals)
    # 3. finish_compact_kernel: counter atomics (returning) -> s_waitcnt vmcnt(0) -> the slab's scan word (sc1 store)
    fc = kernel(funcs, "finish_compact_kernel")
    ctr = [i for i, t in enumerate(fc) if t.startswith("global_atomic_add_x2") and "sc0" in t]
    assert ctr, "finish_compact_kernel: the deferred-interval counter is no returning atomic any more"
    for i in ctr:
        j = i + 1
        while j < len(fc) and not fc[j].startswith("s_waitcnt vmcnt(0)"):
            assert not (fc[j].startswith("global_store") and "sc1" in fc[j]), \
                "finish_compact_kernel: `%s` before the counter atomics were waited for" % fc[j]
            j += 1
        assert j < len(fc)
    report["finish_compact_kernel"]["counter_atomics_checked"] = len(ctr)
    assert report["finish_compact_kernel"]["stores_sc1"] >= 2 and report["finish_compact_kernel"]["loads_sc1"] >= 2
    scans = [k for k in report if k.startswith("scan_compact_kernel")]
    assert scans
    for k in scans:
        assert report[k]["stores_sc1"] >= 2 and report[k]["loads_sc1"] >= 1, (k, report[k])
    return report


def check_litmus(path):
    """libyacrd_litmus.so (csrc/litmus.hip, the -m gpu litmus test) exercises the same instruction forms."""
    ins = kernel(disassemble(path), "litmus_kernel")
    forms = {"store_sc1": any(t.startswith("global_store_dword ") and re.search(r"\bsc1\b", t) and "sc0" not in t for t in ins),
             "load_sc1": any(t.startswith("global_load_dword ") and re.search(r"\bsc1\b", t) for t in ins),
             "returning_atomic": any(t.startswith("global_atomic_add ") and "sc0" in t for t in ins),
             "plain_store": any(t.startswith("global_store_dword ") and "sc1" not in t for t in ins),  # (the control, kind 2)
             "no_cache_maintenance": not any(t.startswith(("buffer_wbl2", "buffer_inv")) for t in ins)}
    assert all(forms.values()), forms
    arr = [i for i, t in enumerate(ins) if t.startswith("global_atomic_add ") and "sc0" in t]
    waited = 0
    for i in arr:  # kind 0 / 2's arrival sits behind the explicit wait; kind 1's counter atomic behind nothing
        try:
            dominated_by_wait(ins, i, "litmus_kernel")
            waited += 1
        except AssertionError:
            pass
    assert waited >= 1, "litmus_kernel: no arrival atomic behind an s_waitcnt vmcnt(0)"
    return forms


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "yacrd_amd", "lib", "libyacrd_hip.so")
    for k, v in check(lib).items():
        print(k, v)
    lit = os.path.join(os.path.dirname(lib), "libyacrd_litmus.so")
    if os.path.exists(lit):
        print("litmus_kernel", check_litmus(lit))
    print("hand-over ISA: ok")
