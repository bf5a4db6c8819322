#!/bin/bash
# thirtieth GPU call of round 4: the build as committed — the whole GPU suite, fuzz (default / second looks / one launch),
# the default bench line, kernel stats of one batch at a time on both paths
out=gpurun_out/r04zd; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
timeout 150 python tools/gpu_fuzz.py 90 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_FUZZ_WIDE=1 timeout 150 python tools/gpu_fuzz.py 90 > $out/fuzz_wide.log 2>&1; tail -1 $out/fuzz_wide.log
YACRD_FUZZ_ONE_LAUNCH=1 timeout 250 python tools/gpu_fuzz.py 180 > $out/fuzz_one_launch.log 2>&1; tail -1 $out/fuzz_one_launch.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json; echo
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04w/bench_default.json").read().strip().splitlines()[-1])
sb = d["small_batches"]
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "| configs[1] pipelined", sb["ms_per_step"], "single", sb["unpredicted_single_batch"]["ms_per_batch"], "one launch", sb["one_launch_single_batch"])
PY
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
find /root/repo/$out/prof -name "*kernel_stats.csv" -exec cp {} /root/repo/$out/kernel_stats_one_batch_at_a_time.csv \;
rm -rf /root/repo/$out/prof; head -6 /root/repo/$out/kernel_stats_one_batch_at_a_time.csv
