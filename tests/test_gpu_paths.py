"""Which path a short batch takes (yacrd_timing, ABI 7: predicted / prediction_misses / fused_reruns / build_switches):
VERDICT r5 weak #5 — one box measured configs[1] at sigma = 300 at 0.130 ms per batch where every other measured 0.059, and
nothing in the line said whether a batch had been run again or the launch's build had flipped.  The counters say; this test
pins what a steady stream of such batches must look like."""
import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import host
from cases import assert_same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engines", [1, 3])
def test_sigma300_batches_take_a_steady_path(engines):
    """40 batches of configs[1] at sigma = 300 (100 000 reads / 5 M overlaps, -c 4: a tenth of the reads is left to the sort,
    so the engine alternates 15 batches of the build with the second looks and one probe of the default build) on fresh
    engines, one at a time and three in flight: no batch is run again, every batch but an engine's first is predicted, the
    build flips at most twice per 16 batches and engine, and the results are the oracle's (src/stack.rs:61-139)."""
    import torch
    off, iv, ln = host.synth_csr(host.SYNTH_ONT, 100_000, 5_000_000, 20241108 + 2, flags=host.SYNTH_F_JITTER | host.synth_f_sigma(300))
    want = oracle.run(off, iv, ln.astype(np.uint64), 4, 0.4, n_threads=16)
    d = [torch.from_numpy(x).cuda() for x in (off.view(np.int64), iv.view(np.int32).reshape(-1), ln.view(np.int32))]
    torch.cuda.synchronize()
    ptrs = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), len(ln), int(off[-1]), 4, 0.4)
    engs = [yacrd_amd.Engine() for _ in range(engines)]
    try:
        K = 40 * engines
        if engines == 1:
            for _ in range(K):
                engs[0].run_device(*ptrs)
        else:
            yacrd_amd.run_device_batches(engs, [ptrs] * K)
        for e in engs:
            t, n = e.timing_total()
            assert n == 40, n
            assert t["fused_reruns"] == 0 and t["prediction_misses"] == 0, t
            assert t["predicted"] >= 38, t              # (an engine's first batch waits for the plan's counts)
            assert t["sorting_build"] == 0, t           # (11 % deferred: never the sorting build)
            assert t["screened"] == 40 and 30 <= t["screen_wide"] <= 39, t  # 15 of 16 batches in the build with the second looks
            assert t["build_switches"] <= 2 * (40 // 16 + 1), t
            assert_same(e.fetch(), want, "sigma 300, %d engine(s)" % engines)
    finally:
        for e in engs:
            e.close()
