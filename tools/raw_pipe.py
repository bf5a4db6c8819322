#!/usr/bin/env python3
"""tools/raw_pipe.py — us per configs[1] batch through yacrd_engines_run_device_batches on three engines,
WITHOUT checking results (for experiments that break them on purpose)."""
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
engs = [yacrd_amd.Engine(device_id=0, flags=yacrd_amd.F_TIMING_SAMPLED) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
yacrd_amd.run_device_batches(engs, [ptrs] * 30)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    yacrd_amd.run_device_batches(engs, [ptrs] * 600)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 600)
print("%.2f us per batch" % (best * 1e6))
