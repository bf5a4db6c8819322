#!/bin/bash
# twenty-fifth GPU call of round 4: timestamps inside one_batch_kernel (-DYK_OB_STAMPS): what the last slabs wait for
out=gpurun_out/r04y; mkdir -p $out
cp variants/libob_stamps.so yacrd_amd/lib/libyacrd_hip.so
cat > /tmp/ob_stamps.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import yacrd_amd
from yacrd_amd import host
o, iv, ln = host.synth_csr(host.SYNTH_ONT, 100000, 5000000, 1)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), iv.view(np.int32).reshape(-1), ln.view(np.int32))]
torch.cuda.synchronize()
with yacrd_amd.Engine(flags=yacrd_amd.F_ONE_LAUNCH | yacrd_amd.F_NO_TIMING) as e:
    for i in range(4):
        print("== launch", i, flush=True)
        e.run_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(ln), int(o[-1]), 4, 0.4)
        torch.cuda.synchronize()
PY
timeout 300 python /tmp/ob_stamps.py > $out/stamps.log 2>&1; tail -22 $out/stamps.log
