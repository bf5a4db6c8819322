#!/usr/bin/env python3
"""tools/gpu_fuzz.py [seconds] — randomized batches through the engine vs the oracle (GPU box).
Every size class, random coverage thresholds, read lengths from 1 bp up, all pile-up modes of
tests/cases.py, wavefronts that are homogeneous (so the pre-filter fires) and mixed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (checker)
import yacrd_amd  # noqa: E402
from cases import assert_same, make_csr  # noqa: E402

MODES = ("regular", "abutting", "dups", "beyond", "sparse", "zero_len", "degenerate")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()) & 0xFFFF)
t0, rounds, reads, ones = time.time(), 0, 0, 0
cflags = (yacrd_amd.F_ALWAYS_DEFER | (yacrd_amd.F_SCREEN_ITEMS_2 if os.environ.get("YACRD_FUZZ_ITEMS2") else 0)
          | (yacrd_amd.F_SCREEN_WIDE if os.environ.get("YACRD_FUZZ_WIDE") else 0))
if os.environ.get("YACRD_FUZZ_ONE_LAUNCH"):  # the one-launch form of short batches (csrc/one_batch.h) where it applies
    cflags = yacrd_amd.F_ONE_LAUNCH
with yacrd_amd.Engine(flags=cflags) as e, yacrd_amd.Engine(flags=yacrd_amd.F_NO_PREFILTER) as ref:
    while time.time() - t0 < budget:
        med = bool(os.environ.get("YACRD_FUZZ_MED"))  # mostly the workgroup classes (513 .. 16 384 intervals): screen_stream.h / screen_wg.h
        R = int(rng.integers(1, 400 if med else 3000))
        hi = int(rng.choice([4200, 9000, 17000] if med else [8, 40, 130, 260, 520, 4200, 17000]))
        sizes = rng.integers(400 if med else 0, hi + 1, size=R)
        if hi > 4200 and not med:
            sizes[rng.random(R) < 0.97] //= 64  # a few big reads only
        kind = rng.integers(0, 3)
        if kind == 0:
            lengths = rng.integers(1, 300, size=R)
        elif kind == 1:
            lengths = rng.integers(300, 20000, size=R)
        else:
            lengths = rng.integers(1000, 2000000, size=R)
        modes = tuple(rng.permutation(MODES)[: int(rng.integers(1, len(MODES) + 1))])
        block = int(rng.choice([1, 64, 512]))
        csr = make_csr(int(rng.integers(1 << 30)), sizes, modes, lengths=lengths, mode_block=block)
        cov = int(rng.choice([0, 1, 2, 3, 4, 7, 20, 100, 2**32 - 1]))
        nc = float(rng.choice([0.0, 0.4, 0.8, 1.0]))
        want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), cov, nc, n_threads=8)
        ctx = "round %d R=%d hi=%d kind=%d modes=%s block=%d cov=%d" % (rounds, R, hi, kind, modes, block, cov)
        assert_same(e.run(*csr, cov, nc), want, ctx)
        assert_same(ref.run(*csr, cov, nc), want, ctx + " (no prefilter)")
        rounds += 1
        reads += R
        ones += int(e.timing().get("one_launch", 0))
print("gpu_fuzz: %d rounds, %d reads, all bit-exact%s" % (rounds, reads, " (%d batches as one launch)" % ones if ones else ""))
