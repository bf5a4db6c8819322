#!/usr/bin/env python3
"""Phase breakdown of the PAF -> CSR ingest (YACRD_INGEST_TIMING) by thread count."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YACRD_INGEST_TIMING"] = "1"
from yacrd_amd import host
reads, overlaps = int(sys.argv[1]), int(sys.argv[2])
path = "/tmp/ingest_%d_%d.paf" % (reads, overlaps)
host.synth_paf(host.SYNTH_ONT, reads, overlaps, 20241110, path)
lib = host.load_library()
for th in [int(x) for x in sys.argv[3:]] or [16, 32, 64, 128]:
    for rep in range(2):
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        lib.yacrd_csr_from_file(path.encode(), 0, th, ctypes.byref(h))
        dt = time.perf_counter() - t0
        lib.yacrd_csr_free(h)
        sys.stderr.write("threads %d rep %d total %.1f ms  %.1f M overlaps/s\n" % (th, rep, dt * 1e3, overlaps / dt / 1e6))
os.remove(path)
