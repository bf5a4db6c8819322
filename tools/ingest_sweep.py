#!/usr/bin/env python3
"""tools/ingest_sweep.py [reads] [overlaps] — PAF text -> CSR / -> read types on the GPU box, by
thread count, file access (pread copies vs slices of a mapping) and destination (host CSR vs records
streamed to HBM during the parse + CSR build on the GPU + engine run).  One JSON line per point."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
O = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
THREADS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 8, 16, 32]
MODE = os.environ.get("SWEEP_CHILD")

if MODE is None:  # parent: generate once, one child process per file-access mode (env var is read at load)
    import yacrd_amd  # noqa: F401
    from yacrd_amd import host
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(d, "sweep_%d_%d.paf" % (R, O))
    t0 = time.perf_counter()
    host.synth_paf(host.SYNTH_ONT, R, O, 20241110, path)
    print(json.dumps({"paf_bytes": os.path.getsize(path), "generate_s": round(time.perf_counter() - t0, 2),
                      "dir": d}), flush=True)
    for mode in ("pread", "mmap"):
        env = dict(os.environ, SWEEP_CHILD=mode, SWEEP_PATH=path)
        if mode == "mmap":
            env["YACRD_INGEST_MMAP"] = "1"
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=True)
    os.remove(path)
    sys.exit(0)

import yacrd_amd
from yacrd_amd import host
path = os.environ["SWEEP_PATH"]
size = os.path.getsize(path)
lib = host.load_library()
have_gpu = True
try:
    eng = yacrd_amd.Engine()
except yacrd_amd.EngineError:
    have_gpu = False
for th in THREADS:
    best = None
    for rep in range(3):
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = lib.yacrd_csr_from_file(path.encode(), 0, th, ctypes.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0, lib.yacrd_host_last_error()
        lib.yacrd_csr_free(h)
        best = dt if best is None else min(best, dt)
    print(json.dumps({"file": MODE, "dest": "host_csr", "threads": th, "s": round(best, 4),
                      "M_overlaps_per_s": round(O / best / 1e6, 2), "GB_per_s": round(size / best / 1e9, 2)}), flush=True)
    if not have_gpu:
        continue
    with yacrd_amd.Stream(eng) as st:
        best, keep = None, None
        for rep in range(3):
            sink = st.sink()
            h = ctypes.c_void_p()
            t0 = time.perf_counter()
            rc = lib.yacrd_ingest_stream(path.encode(), 0, th, ctypes.addressof(sink), ctypes.byref(h))
            t1 = time.perf_counter()
            assert rc == 0, lib.yacrd_host_last_error()
            v = host._View()
            lib.yacrd_csr_get(h, ctypes.byref(v))
            mp = ctypes.POINTER(ctypes.c_uint32)()
            nh = ctypes.c_uint64()
            lib.yacrd_csr_handle_map(h, ctypes.byref(mp), ctypes.byref(nh))
            res = yacrd_amd.engine._Result()
            rc = yacrd_amd.load_library().yacrd_stream_finish(st._h, mp, nh, v.lengths, v.n_reads, 4, 0.4,
                                                             ctypes.byref(res))
            t2 = time.perf_counter()
            assert rc == 0
            yacrd_amd.load_library().yacrd_result_free(ctypes.byref(res))
            lib.yacrd_csr_free(h)
            if best is None or t2 - t0 < best:
                best, keep = t2 - t0, (t1 - t0, t2 - t1, st.stats())
        stats = keep[2]
        print(json.dumps({"file": MODE, "dest": "stream_to_hbm+run", "threads": th, "s": round(best, 4),
                          "M_overlaps_per_s": round(O / best / 1e6, 2), "parse_s": round(keep[0], 4),
                          "finish_s": round(keep[1], 4),
                          "h2d_GB_per_s_while_busy": round(stats["h2d_bytes"] / max(stats["h2d_busy_ms"], 1e-6) / 1e6, 1),
                          "build_ms": round(stats["build_ms"], 3), "run_ms": round(stats["run_ms"], 3),
                          "d2h_ms": round(stats["d2h_ms"], 3)}), flush=True)
