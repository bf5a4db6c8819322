#!/usr/bin/env python3
"""tools/scale_check.py — BASELINE.json configs at FULL size on the GPU box: bit-exact comparison
of the engine against the CPU oracle (all host cores) plus size-independent properties
(read-partition invariance, interval-order invariance).  Prints one JSON line per config.

  python tools/scale_check.py 2 3 4        # config numbers (1-based, BASELINE.json order)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
import yacrd_amd  # noqa: E402
from yacrd_amd import host  # noqa: E402

CONFIGS = {
    2: dict(profile=host.SYNTH_ONT, reads=100_000, overlaps=5_000_000, cov=4, nc=0.4),
    3: dict(profile=host.SYNTH_SEQUEL, reads=2_000_000, overlaps=200_000_000, cov=3, nc=0.4),
    4: dict(profile=host.SYNTH_SKEWED, reads=10_000, overlaps=30_000_000, cov=4, nc=0.4),
    5: dict(profile=host.SYNTH_SEQUEL, reads=5_000_000, overlaps=500_000_000, cov=3, nc=0.4),
}


def same(a, b):
    return bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]))


def main():
    which = [int(x) for x in sys.argv[1:]] or [2]
    scale = float(os.environ.get("YACRD_SCALE", "1"))
    ncores = os.cpu_count() or 1
    for c in which:
        cfg = CONFIGS[c]
        R, O = int(cfg["reads"] * scale), int(cfg["overlaps"] * scale)
        t0 = time.perf_counter()
        sflags = int(os.environ.get("YACRD_SYNTH_FLAGS", "0"))  # 1 = no abutting/degenerate injection
        offsets, intervals, lengths = host.synth_csr(cfg["profile"], R, O, 20241108 + c, sflags)
        t_gen = time.perf_counter() - t0
        out = {"config": c, "synth_flags": sflags, "reads": R, "overlaps": O, "intervals": int(offsets[-1]),
               "max_intervals_per_read": int(np.diff(offsets.astype(np.int64)).max()),
               "gen_s": round(t_gen, 2)}
        with yacrd_amd.Engine(device_id=0, flags=yacrd_amd.F_TIMING_FULL) as e:
            t0 = time.perf_counter()
            got = e.run(offsets, intervals, lengths, cfg["cov"], cfg["nc"])
            out["gpu_run_s_incl_pcie"] = round(time.perf_counter() - t0, 3)
            out["timing"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.timing().items()}
            t0 = time.perf_counter()
            want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cfg["cov"], cfg["nc"],
                              n_threads=ncores)
            out["oracle_s_%d_threads" % ncores] = round(time.perf_counter() - t0, 3)
            out["bit_exact_all_reads"] = same(got, want)
            out["regions"] = int(want[0][-1])
            out["types"] = np.bincount(want[2], minlength=3).tolist()
            # property: 8-way read partition (what 8 GPUs would each get) == whole
            # (eight engines on this one device; an engine serves one call at a time)
            engines = [yacrd_amd.Engine(device_id=0) for _ in range(8)]
            try:
                parts = yacrd_amd.run_partitioned(engines, offsets, intervals, lengths,
                                                  cfg["cov"], cfg["nc"])
            finally:
                for x in engines:
                    x.close()
            out["partition8_invariant"] = same(parts, want)
            # property: interval order inside reads is irrelevant (the reference sorts)
            rng = np.random.default_rng(c)
            n_per = np.diff(offsets.astype(np.int64))
            keys = np.repeat(np.arange(R, dtype=np.int64), n_per) * (1 << 32) + rng.integers(0, 1 << 32, size=int(offsets[-1]))
            perm = np.argsort(keys, kind="stable")
            shuffled = e.run(offsets, intervals[perm], lengths, cfg["cov"], cfg["nc"])
            out["order_invariant"] = same(shuffled, want)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
