#!/bin/bash
# thirty-fourth GPU call of round 4: one_batch_kernel at occupancy 6 / 5 / 4
out=gpurun_out/r04zh; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_one_launch.py -x -q > $out/pytest_one_launch.log 2>&1; tail -4 $out/pytest_one_launch.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; o=h["one_launch_single_batch"]; print("pipelined %.5f three-launch single %.5f one-launch single %.5f %s deferred %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], o["ms_per_batch"], o["ran_as_one_launch"], o["deferred_reads"], o["parity"][:9]))'
for v in occ6 occ5 occ4 occ5slab256; do cp variants/libob_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 100; do echo -n "== $v configs[1] jitter $j: "; timeout 600 python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>$out/bench_err.log | python -c "$Q"; done
done > $out/one_launch_single_batch.log 2>&1
cat $out/one_launch_single_batch.log
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
for v in occ6 occ5 occ4 occ5slab256; do cp /root/repo/variants/libob_$v.so /root/repo/yacrd_amd/lib/libyacrd_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$v -o s -- python /tmp/ob_prof.py > /root/repo/$out/prof_$v.log 2>&1
  echo -n "$v: "; find /root/repo/$out/prof_$v -name "*kernel_stats.csv" -exec grep one_batch {} \; ; rm -rf /root/repo/$out/prof_$v
done > /root/repo/$out/phases.log 2>&1
cat /root/repo/$out/phases.log
