#!/usr/bin/env python3
"""tools/host_cost.py — host time of one yacrd_engine_submit_device call (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yacrd_amd
from yacrd_amd import host
yacrd_amd.load_library()
import torch
dev = torch.device("cuda", 0)
o, iv, ln = host.synth_csr(host.SYNTH_ONT, 100000, 5000000, 20241110)
d = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), iv.view(np.int32), ln.view(np.int32))]
ptrs = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 100000, int(o[-1]), 4, 0.4)
for flags in (0, yacrd_amd.F_NO_TIMING):
    e = yacrd_amd.Engine(device_id=0, flags=flags)
    for _ in range(20):
        e.run_device(*ptrs)
    sub = wait = 0.0
    N = 300
    for _ in range(N):
        t0 = time.perf_counter()
        e.submit_device(*ptrs)
        t1 = time.perf_counter()
        e.wait()
        t2 = time.perf_counter()
        sub += t1 - t0
        wait += t2 - t1
    print("flags %d: submit %.1f us, wait %.1f us per batch" % (flags, sub / N * 1e6, wait / N * 1e6))
