#!/usr/bin/env python3
"""tools/paf_matrix.py — yacrd_engine_ingest_paf on the configs[1] PAF text: threads (x segment size when the library is built with a YACRD_PAF_SEG
override: the run logged in profiles/r03/paf_matrix.log; 128 MiB is compiled in now), best and median of 5 (GPU box)."""
import ctypes, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yacrd_amd
from yacrd_amd import host
paf = "/dev/shm/yacrd_matrix.paf"
host.synth_paf(host.SYNTH_ONT, 100000, 5000000, 20241110, paf)
el = yacrd_amd.load_library()
try:
    with yacrd_amd.Engine() as eng:
        for seg in (sys.argv[1:] or ["8", "32", "100000"]):
            os.environ["YACRD_PAF_SEG"] = seg
            for th in (4, 6, 8):
                ts, last = [], None
                for rep in range(6):
                    res, rd, stt = yacrd_amd.engine._Result(), yacrd_amd.engine._Reads(), yacrd_amd.engine._IngestStats()
                    t0 = time.perf_counter()
                    rc = el.yacrd_engine_ingest_paf(eng._h, paf.encode(), th, 4, 0.4, ctypes.byref(res), ctypes.byref(rd), ctypes.byref(stt))
                    dt = time.perf_counter() - t0
                    assert rc == 0, rc
                    el.yacrd_result_free(ctypes.byref(res)); el.yacrd_reads_free(ctypes.byref(rd))
                    if rep:
                        ts.append(dt * 1e3)
                    last = (round(stt.text_ms, 2), round(stt.parse_ms, 2), round(stt.build_ms, 2))
                print("seg %6s threads %d: best %.2f ms median %.2f ms  (last: text %.2f parse %.2f build %.2f)" % (seg, th, min(ts), statistics.median(ts), *last))
finally:
    os.remove(paf)
