"""The ISA the device-side hand-overs rest on (one_batch.h, finish_compact.h: relaxed agent-scope accesses + s_waitcnt,
no release / acquire — DESIGN.md §3.10): tools/isa_handover.py on the built library, on CPU (llvm-objdump of the
gfx950 code objects inside libyacrd_hip.so).  VERDICT r4 item 4: "nothing asserts the ISA it depends on"."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "yacrd_amd", "lib", "libyacrd_hip.so")
LITMUS = os.path.join(ROOT, "yacrd_amd", "lib", "libyacrd_litmus.so")

pytestmark = pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"),
                                reason="needs the built library and llvm-objdump")


def test_hand_over_isa_of_the_built_library():
    import isa_handover
    rep = isa_handover.check(LIB)
    assert rep["one_batch_kernel"]["arrivals_checked"] >= 1 and rep["finish_compact_kernel"]["counter_atomics_checked"] >= 1
    scans = [k for k in rep if k.startswith("scan_compact_kernel")]  # (a template over the reads per thread: every instantiation)
    assert scans
    for k in ["one_batch_kernel", "finish_compact_kernel"] + scans:
        assert rep[k]["stores_sc1"] >= 2 and rep[k]["loads_sc1"] >= 1, (k, rep[k])


def test_litmus_library_uses_the_same_instruction_forms():
    import isa_handover
    forms = isa_handover.check_litmus(LITMUS)
    assert forms["store_sc1"] and forms["load_sc1"] and forms["returning_atomic"] and forms["no_cache_maintenance"]


def test_the_dominance_check_catches_what_it_is_there_for():
    from isa_handover import Ins, dominated_by_wait

    def prog(lines):
        out = []
        for k, (t, tgt) in enumerate(lines):
            i = Ins(t)
            i.addr, i.target = 0x100 + 4 * k, (0x100 + 4 * tgt if tgt is not None else None)
            out.append(i)
        return out
    good = prog([("global_store_dword v[0:1], v2, off sc1", None), ("s_waitcnt vmcnt(0)", None), ("v_mov_b32 v0, 0", None),
                 ("s_cbranch_execz 2", 5), ("global_atomic_add_x2 v[2:3], v0, v[2:3], s[0:1] sc0", None), ("s_nop 0", None)])
    assert dominated_by_wait(good, 4, "t") == 3
    late_store = prog([("s_waitcnt vmcnt(0)", None), ("global_store_dword v[0:1], v2, off sc1", None),
                       ("global_atomic_add_x2 v[2:3], v0, v[2:3], s[0:1] sc0", None)])
    with pytest.raises(AssertionError, match="between the last s_waitcnt"):
        dominated_by_wait(late_store, 2, "t")
    jump_in = prog([("s_cbranch_scc1 3", 3), ("global_store_dword v[0:1], v2, off sc1", None), ("s_waitcnt vmcnt(0)", None),
                    ("v_mov_b32 v0, 0", None), ("global_atomic_add_x2 v[2:3], v0, v[2:3], s[0:1] sc0", None)])
    with pytest.raises(AssertionError, match="jumps in"):
        dominated_by_wait(jump_in, 4, "t")
    no_wait = prog([("v_mov_b32 v0, 0", None), ("global_atomic_add_x2 v[2:3], v0, v[2:3], s[0:1] sc0", None)])
    with pytest.raises(AssertionError, match="no s_waitcnt"):
        dominated_by_wait(no_wait, 1, "t")
