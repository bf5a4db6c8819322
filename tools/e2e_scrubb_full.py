#!/usr/bin/env python3
"""tools/e2e_scrubb_full.py [reads overlaps] — BASELINE configs[4] as the config says it: a 5 M-read synthetic FASTQ +
a 500 M-overlap PAF through the drop-in CLI with its DEFAULT flags (no -t),
    yacrd -i s.paf -o r.yacrd -c 3 -n 0.4 scrubb -i s.fastq -o o.fastq
on one MI355X, wall clock per stage (YACRD_CLI_TIMING), and the result checked against the CPU restatements
(reference: src/editor/scrubbing.rs:156-236, tests/run.rs:254-300) — round 5: over the WHOLE output, not its head:
  * the report: every read's length, type and regions against the oracle's (all reads, arrays compared);
  * totals: the scrubbed file's record count and byte count against what the oracle's regions imply — every piece
    of every read (names `<id>_<b>_<e>`, dropped NotCovered reads, untouched unknown reads) summed in numpy;
  * windows: N_WINDOWS stretches of records spread over the whole input (the first and the last records among them),
    each through oracle/editors.py and compared byte for byte with the output at the place it must be found
    (chimeric, NotCovered and unmentioned reads among the records looked at, asserted).
Sizes are cut down when /dev/shm cannot hold the three files.  `verify_scrubb` is what tests/test_gpu_cli.py runs at
1/100 of the size."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
from oracle import editors as oed  # noqa: E402
from yacrd_amd import host  # noqa: E402

N_WINDOWS = 240
NOT_BAD, CHIMERIC, NOT_COVERED = 0, 1, 2


def digits(x):
    """decimal digits of the non-negative integers in x"""
    x = np.asarray(x, dtype=np.uint64)
    d = np.ones(x.shape, dtype=np.int64)
    p = np.uint64(10)
    for _ in range(19):
        d += x >= p
        p = np.uint64(int(p) * 10) if int(p) * 10 < 2 ** 64 else np.uint64(2 ** 64 - 1)
    return d


def expected_totals(ln, bo, br, rt):
    """(records, bytes) that scrubb writes for the generator's reads r%09d (header `@<name> synthetic len=<L>`), from
    the oracle's regions: scrubbing.rs:183-233 — NotCovered: dropped; no region: the record as it is; else the pieces
    between the regions ((0, 0) in front dropped with the first region at 0, nothing behind a region that ends at the
    read's length), each `@<name>_<b>_<e> <desc>`."""
    ln = np.asarray(ln, dtype=np.int64)
    bo = np.asarray(bo, dtype=np.int64)
    br = np.asarray(br, dtype=np.int64).reshape(-1, 2)
    rt = np.asarray(rt)
    R = len(ln)
    k = np.diff(bo)
    name_len, desc_len = 10, 14 + digits(ln)  # "r%09d", "synthetic len=%d"
    rec_whole = name_len + desc_len + 2 * ln + 7  # @name desc\nseq\n+\nqual\n
    keep = rt != NOT_COVERED
    whole = keep & (k == 0)
    records = int(whole.sum())
    nbytes = int(rec_whole[whole].sum())
    # pieces of the reads with regions: in front of region j: (end of region j - 1 | 0, begin of region j), unless j is the
    # read's first region and begins at 0; behind the read's last region: (its end, L) unless it ends at L
    rid = np.repeat(np.arange(R), k)
    live = keep[rid]
    first = np.zeros(len(rid), dtype=bool)
    first[bo[:-1][k > 0]] = True
    prev_end = np.zeros(len(rid), dtype=np.int64)
    prev_end[1:] = br[:-1, 1]
    prev_end[first] = 0
    p0, p1 = prev_end, br[:, 0]
    ok = live & ~(first & (br[:, 0] == 0))
    assert np.all(p1[ok] >= p0[ok]), "regions out of order"
    piece = name_len + 1 + digits(p0) + 1 + digits(p1) + desc_len[rid] + 2 * (p1 - p0) + 7
    records += int(ok.sum())
    nbytes += int(piece[ok].sum())
    last = bo[1:][k > 0] - 1
    lr = rid[last]
    e = br[last, 1]
    ok2 = keep[lr] & (e != ln[lr])
    piece2 = name_len + 1 + digits(e) + 1 + digits(ln[lr]) + desc_len[lr] + 2 * (ln[lr] - e) + 7
    records += int(ok2.sum())
    nbytes += int(piece2[ok2].sum())
    return records, nbytes, int(rec_whole.sum())


def records_at(f, size, at, max_bytes):
    """Whole FASTQ records of the generator's file starting at the first record boundary at or behind byte `at`
    (quality is '?' only, so a line that begins with '@' is a header): (offset, bytes)."""
    f.seek(at)
    buf = f.read(min(max_bytes + (1 << 20), size - at))
    if at == 0:
        s = 0
    else:
        s = buf.find(b"\n@") + 1
        if s == 0:
            return None
    e = s
    while True:
        nxt = e
        for _ in range(4):
            nxt = buf.find(b"\n", nxt) + 1
            if nxt == 0:
                break
        if nxt == 0 or nxt - s > max_bytes:
            break
        e = nxt
        if e == len(buf):
            break
    return (at + s, buf[s:e]) if e > s else None


def verify_scrubb(fq, out, rep, off, iv, ln, cov, nc, n_extras, n_windows=N_WINDOWS, window_bytes=600_000, log=print):
    """The CLI's report `rep` and scrubbed FASTQ `out` of the generator's FASTQ `fq` against the oracle over the CSR
    (off, iv, ln): every read's report line, the output's totals, n_windows byte-exact windows.  Returns a dict."""
    R = len(ln)
    t0 = time.perf_counter()
    bo, br, rt = oracle.run(off, iv, ln.astype(np.uint64), cov, nc, n_threads=max(1, (os.cpu_count() or 2) // 2))
    bo, br = np.asarray(bo, dtype=np.int64), np.asarray(br, dtype=np.int64).reshape(-1, 2)
    log("oracle over all %d reads: %.1f s" % (R, time.perf_counter() - t0))
    res = {}
    # ---- the report, every line of it
    names, rl, rbo, rbr = host.report_read(rep)
    idx = np.fromiter((int(n[1:]) for n in names), dtype=np.int64, count=len(names))
    assert all(n[0] == "r" for n in names[:1000]) and len(np.unique(idx)) == len(idx)
    mentioned = np.diff(np.asarray(off, dtype=np.int64)) > 0
    assert len(idx) == int(mentioned.sum()), (len(idx), int(mentioned.sum()))
    rk = np.diff(rbo.astype(np.int64))
    same_len = bool(np.array_equal(rl.astype(np.int64), np.asarray(ln, dtype=np.int64)[idx]))
    same_cnt = bool(np.array_equal(rk, np.diff(bo)[idx]))
    gi = np.repeat(bo[:-1][idx] - rbo[:-1].astype(np.int64), rk) + np.arange(int(rbo[-1])) if same_cnt else None
    same_reg = bool(same_cnt and np.array_equal(rbr.astype(np.int64).reshape(-1, 2), br[gi]))
    with open(rep, "rb") as f:
        tl = np.fromiter((len(l.split(b"\t", 1)[0]) for l in f), dtype=np.int64, count=len(names))  # NotBad 6, Chimeric 8, NotCovered 10
    same_type = bool(np.array_equal((tl - 6) // 2, np.asarray(rt, dtype=np.int64)[idx]))
    res["report"] = {"lines": len(names), "lengths": same_len, "region_counts": same_cnt, "regions": same_reg, "types": same_type}
    log("report: %d lines; lengths %s, region counts %s, regions %s, types %s" % (len(names), same_len, same_cnt, same_reg, same_type))
    # ---- totals
    in_size, out_size = os.path.getsize(fq), os.path.getsize(out)
    want_rec, want_bytes, in_reads_bytes = expected_totals(ln, bo, br, rt)
    extras_bytes = in_size - in_reads_bytes  # the unmentioned reads x%09d are copied through as they are
    want_rec += n_extras
    want_bytes += extras_bytes
    t0 = time.perf_counter()
    got_lines = int(subprocess.run(["wc", "-l", out], capture_output=True, text=True, check=True).stdout.split()[0])
    res["totals"] = {"records": got_lines // 4, "records_expected": want_rec, "bytes": out_size, "bytes_expected": want_bytes,
                     "types": np.bincount(np.asarray(rt), minlength=3).tolist()}
    log("totals: %d records (expected %d), %d bytes (expected %d); wc -l took %.0f s; oracle types NotBad / Chimeric / NotCovered %s" % (
        got_lines // 4, want_rec, out_size, want_bytes, time.perf_counter() - t0, res["totals"]["types"]))
    # ---- windows
    seen = {"chimeric": 0, "not_covered": 0, "unmentioned": 0, "records": 0, "bytes_in": 0, "bytes_out": 0}
    bad = []
    off64 = np.asarray(off, dtype=np.int64)
    with open(fq, "rb") as fi, open(out, "rb") as fo:
        starts = [int(k * (in_size - 1) / max(1, n_windows - 2)) for k in range(n_windows - 1)]
        starts[-1] = max(0, in_size - window_bytes)  # (the last records: the window runs to the end of the file)
        pos_hint = 0.0
        for w, at in enumerate(starts):
            got = records_at(fi, in_size, at, window_bytes)
            if got is None:
                continue
            a0, chunk = got
            ids, table = [], {}
            for l in chunk.split(b"\n")[0::4]:
                if l.startswith(b"@r"):
                    ids.append(int(l[2:11]))
                elif l.startswith(b"@x"):
                    seen["unmentioned"] += 1
            for r in ids:
                regs = [tuple(int(x) for x in br[k]) for k in range(int(bo[r]), int(bo[r + 1]))]
                table["r%09d" % r] = (regs, int(ln[r]))
                seen["chimeric"] += int(rt[r] == CHIMERIC)
                seen["not_covered"] += int(rt[r] == NOT_COVERED)
            want = oed.edit_fastq("scrubb", chunk, table, nc)
            seen["records"] += chunk.count(b"\n") // 4
            seen["bytes_in"] += len(chunk)
            seen["bytes_out"] += len(want)
            if not want:
                continue
            if a0 == 0:
                fo.seek(0)
                okw = fo.read(len(want)) == want
            elif a0 + len(chunk) == in_size:
                fo.seek(max(0, out_size - len(want)))
                okw = fo.read() == want
            else:  # where the output must hold it: near the same fraction of the file
                guess = int(a0 / in_size * out_size)
                okw = False
                for slack in (64 << 20, 1 << 30, 8 << 30):
                    lo = max(0, guess - slack)
                    fo.seek(lo)
                    hay = fo.read(min(out_size - lo, 2 * slack + len(want)))
                    p = hay.find(want[:4096])
                    if p >= 0:
                        okw = hay[p:p + len(want)] == want if p + len(want) <= len(hay) else (fo.seek(lo + p) or fo.read(len(want)) == want)
                        pos_hint = (lo + p) / out_size - a0 / in_size
                        break
            if not okw:
                bad.append((w, a0, ids[:2]))
    res["windows"] = dict(seen, windows=len(starts), mismatches=len(bad), first_bad=bad[:3], drift=pos_hint)
    log("windows: %d of them, %d records (%.1f MB in -> %.1f MB out) byte for byte; chimeric %d, NotCovered %d, unmentioned %d; mismatches %d %s" % (
        len(starts), seen["records"], seen["bytes_in"] / 1e6, seen["bytes_out"] / 1e6, seen["chimeric"], seen["not_covered"],
        seen["unmentioned"], len(bad), bad[:3]))
    res["ok"] = bool(same_len and same_cnt and same_reg and same_type and got_lines == 4 * want_rec and out_size == want_bytes
                     and not bad and seen["chimeric"] > 0 and seen["not_covered"] > 0 and seen["unmentioned"] > 0)
    return res


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    O = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000_000
    d = os.environ.get("YACRD_E2E_DIR", "/dev/shm")
    st = os.statvfs(d)
    free = st.f_bavail * st.f_frsize
    need = O * 75 + 2 * R * 21000  # PAF + FASTQ in + FASTQ out
    while need > 0.8 * free and R > 1000:
        R //= 2
        O //= 2
        need = O * 75 + 2 * R * 21000
    print("free on %s: %.0f GB; running %d reads / %d overlaps" % (d, free / 1e9, R, O), flush=True)
    seed = 20241108 + 5
    paf, fq, rep, out = (os.path.join(d, "yacrd_full_%d.%s" % (os.getpid(), x)) for x in ("paf", "fastq", "yacrd", "out.fastq"))
    exe = os.path.join(ROOT, "yacrd_amd", "bin", "yacrd")
    n_extras = R // 200
    try:
        t0 = time.perf_counter()
        host.synth_paf(host.SYNTH_SEQUEL, R, O, seed, paf)
        t1 = time.perf_counter()
        host.synth_fastq(host.SYNTH_SEQUEL, R, O, seed, n_extras, fq)
        t2 = time.perf_counter()
        print("generated PAF %.1f GB in %.0f s, FASTQ %.1f GB in %.0f s" % (os.path.getsize(paf) / 1e9, t1 - t0, os.path.getsize(fq) / 1e9, t2 - t1), flush=True)
        time.sleep(5)  # (the generators' burst on all CPUs: let the cgroup quota recover)
        for rep_no in range(2):
            for x in (rep, out):  # (a run over the outputs of the run before frees their 99 GB inside the editor's open: 5-6 s)
                if os.path.exists(x):
                    os.remove(x)
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-i", paf, "-o", rep, "-c", "3", "-n", "0.4", "scrubb", "-i", fq, "-o", out],
                               env=dict(os.environ, YACRD_CLI_TIMING="1"), capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr
            stages = {l.split()[1]: float(l.split()[2]) for l in p.stderr.splitlines() if l.startswith("[timing]")}
            print("run %d: %.2f s wall; stages %s; scrubb %.1f GB/s of FASTQ in; report %d MB, scrubbed %.1f GB" % (
                rep_no, dt, stages, os.path.getsize(fq) / max(stages.get("edit", dt), 1e-9) / 1e9, os.path.getsize(rep) >> 20,
                os.path.getsize(out) / 1e9), flush=True)
            time.sleep(3)
        # the same report from TWO engines on this device (yacrd_engines_ingest_overlaps: a byte range of the text each, a
        # range of the reads each): byte for byte the one-engine report
        rep2 = rep + ".gpus2"
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", paf, "-o", rep2, "-c", "3", "-n", "0.4", "--gpus", "2"],
                           env=dict(os.environ, YACRD_CLI_TIMING="1", YACRD_GPUS_ON_DEVICE="0"), capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0 and "device parser: 2 engine(s)" in p.stderr, p.stderr
        stages = {l.split()[1]: float(l.split()[2]) for l in p.stderr.splitlines() if l.startswith("[timing]")}
        same2 = subprocess.run(["cmp", "-s", rep, rep2]).returncode == 0
        print("--gpus 2 on one device: %.2f s wall; stages %s; report %s" % (dt, stages, "byte-identical to one engine's" if same2 else "DIFFERS"), flush=True)
        os.remove(rep2)
        assert same2
        os.remove(paf)  # (room for the checker's arrays)
        off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, R, O, seed)
        res = verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, n_extras, log=lambda s: print(s, flush=True))
        print("VERDICT:", "everything identical" if res["ok"] else "MISMATCH", flush=True)
        assert res["ok"], res
    finally:
        for x in (paf, fq, rep, out):
            if os.path.exists(x):
                os.remove(x)


if __name__ == "__main__":
    main()
