#!/bin/bash
# twentieth GPU call of round 4: where one_batch_kernel's time goes (phases S + A alone / with the arrivals / whole), rocprofv3
out=gpurun_out/r04t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ob_prof.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import yacrd_amd
from yacrd_amd import host
o, iv, ln = host.synth_csr(host.SYNTH_ONT, 100000, 5000000, 1)
dev = torch.device("cuda", 0)
t = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), iv.view(np.int32).reshape(-1), ln.view(np.int32))]
torch.cuda.synchronize()
with yacrd_amd.Engine(flags=yacrd_amd.F_ONE_LAUNCH | yacrd_amd.F_NO_TIMING) as e:
    for _ in range(40):
        e.run_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(ln), int(o[-1]), 4, 0.4)
PY
for v in e1 e1occ8 e2 slab128 e1items2 e2items2 slab128items2; do cp /root/repo/variants/libob_$v.so /root/repo/yacrd_amd/lib/libyacrd_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$v -o s -- python /tmp/ob_prof.py > /root/repo/$out/prof_$v.log 2>&1
  echo -n "$v: "; find /root/repo/$out/prof_$v -name "*kernel_stats.csv" -exec grep one_batch {} \; ; rm -rf /root/repo/$out/prof_$v
done > /root/repo/$out/phases.log 2>&1
cat /root/repo/$out/phases.log
