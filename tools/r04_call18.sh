#!/bin/bash
# eighteenth GPU call of round 4: one_batch_kernel, phase B with every load in flight; slab sizes / occupancy / items
out=gpurun_out/r04s; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_one_launch.py -x -q > $out/pytest_one_launch.log 2>&1; tail -6 $out/pytest_one_launch.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; o=h["one_launch_single_batch"]; print("pipelined %.5f three-launch single %.5f one-launch single %.5f %s deferred %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], o["ms_per_batch"], o["ran_as_one_launch"], o["deferred_reads"], o["parity"][:9]))'
for v in slab128 slab256 slab128items2; do cp variants/libob_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 100; do echo -n "== $v configs[1] jitter $j: "; timeout 600 python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>$out/bench_err.log | python -c "$Q"; done
done > $out/one_launch_single_batch.log 2>&1
cat $out/one_launch_single_batch.log
cp variants/libob_slab128.so yacrd_amd/lib/libyacrd_hip.so
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
for flags in (yacrd_amd.F_ONE_LAUNCH | yacrd_amd.F_NO_TIMING, yacrd_amd.F_NO_PREDICTION | yacrd_amd.F_NO_TIMING):
    with yacrd_amd.Engine(flags=flags) as e:
        for _ in range(60):
            e.run_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(ln), int(o[-1]), 4, 0.4)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof -o s -- python /tmp/ob_prof.py > /root/repo/$out/prof.log 2>&1
find /root/repo/$out/prof -name "*kernel_stats.csv" -exec cp {} /root/repo/$out/kernel_stats_one_launch_vs_three.csv \;
rm -rf /root/repo/$out/prof; head -8 /root/repo/$out/kernel_stats_one_launch_vs_three.csv
