#!/usr/bin/env python3
"""tools/ingest_bench.py — overlaps/s ingested: synthetic PAF text -> CSR (host parser), by threads."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yacrd_amd import host  # noqa: E402

reads, overlaps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
path = "/tmp/ingest_%d_%d.paf" % (reads, overlaps)
t0 = time.perf_counter()
host.synth_paf(host.SYNTH_ONT, reads, overlaps, 20241110, path)
size = os.path.getsize(path)
res = {"paf_bytes": size, "write_s": round(time.perf_counter() - t0, 2), "runs": []}
import ctypes
lib = host.load_library()
for th in (1, 8, 32, 64, 128, 0):
    best = None
    for rep in range(3):  # the raw C call (no numpy copies, no name decoding), best of 3
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = lib.yacrd_csr_from_file(path.encode(), 0, th, ctypes.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0
        lib.yacrd_csr_free(h)
        best = dt if best is None else min(best, dt)
    res["runs"].append({"threads": th or (os.cpu_count() or 0), "s": round(best, 4),
                        "overlaps_per_s": round(overlaps / best), "GB_per_s": round(size / best / 1e9, 3)})
os.remove(path)
print(json.dumps(res))
