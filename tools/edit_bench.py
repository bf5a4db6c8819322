#!/usr/bin/env python3
"""tools/edit_bench.py [reads overlaps] — the chunk-parallel scrubb alone (no GPU: the oracle makes the table) on a synthetic
FASTQ in /dev/shm, by threads, by the chunks' way into memory (YACRD_EDIT_IO=pread | mmap) and out of it (YACRD_EDIT_OUT=
turns | pwrite | map: one writer at a time in chunk order, all at once, a shared mapping): GB/s of FASTQ in.  (1 thread = the one-thread loop: stream in, stream out.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (makes the bad-region table here: no engine in this tool)
from yacrd_amd import host  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
O = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
d = os.environ.get("YACRD_EDIT_BENCH_DIR", "/dev/shm")
fq, out = os.path.join(d, "yacrd_eb_%d.fastq" % os.getpid()), os.path.join(d, "yacrd_eb_%d.out.fastq" % os.getpid())
try:
    t0 = time.perf_counter()
    host.synth_fastq(host.SYNTH_SEQUEL, R, O, 7, R // 200, fq)
    size = os.path.getsize(fq)
    print("FASTQ %.1f GB in %.1f s" % (size / 1e9, time.perf_counter() - t0), flush=True)
    off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, R, O, 7)
    bo, br, rt = oracle.run(off, iv, ln.astype(np.uint64), 3, 0.4, n_threads=16)
    del off, iv
    names = ["r%09d" % i for i in range(R)]
    time.sleep(5)
    ref = None
    ways = [tuple(w.split(":")) for w in os.environ.get("YACRD_EDIT_BENCH_WAYS", "pread:turns,pread:pwrite,pread:map,mmap:turns").split(",")]
    threads = [int(x) for x in os.environ.get("YACRD_EDIT_BENCH_THREADS", "1,4,8,16,32").split(",")]
    for io, oo in ways:
        os.environ["YACRD_EDIT_IO"] = io
        os.environ["YACRD_EDIT_OUT"] = oo
        io = "in " + io + " / out " + oo
        for th in threads:
            if th == 1 and oo != "turns":
                continue
            if os.path.exists(out):
                os.remove(out)  # (or the open's O_TRUNC frees the 20 GB of the run before inside the timed region: 1.5-1.9 s)
            t0 = time.perf_counter()
            host.edit_file(host.OP_SCRUBB, fq, out, names, ln, bo, br, rt, n_threads=th)
            dt = time.perf_counter() - t0
            sig = (os.path.getsize(out),)
            ref = ref or sig
            print("%s %2d threads: %.2f s = %.2f GB/s in (out %.1f GB) %s" % (io, th, dt, size / dt / 1e9, sig[0] / 1e9, "" if sig == ref else "SIZE DIFFERS"), flush=True)
            time.sleep(2)
finally:
    for x in (fq, out):
        if os.path.exists(x):
            os.remove(x)
