#!/bin/bash
# sixteenth GPU call of round 4: YACRD_F_ONE_LAUNCH (one_batch_kernel) — its tests, the whole GPU suite, one batch at a time
out=gpurun_out/r04p; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_one_launch.py -x -q > $out/pytest_one_launch.log 2>&1; tail -15 $out/pytest_one_launch.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_one_launch.py > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("pipelined %.5f three-launch single %.5f one-launch single %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], json.dumps(h["one_launch_single_batch"])))'
for j in 0 100; do echo -n "== configs[1] jitter $j: "; timeout 600 python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>$out/bench_err_$j.log | python -c "$Q"; done > $out/one_launch_single_batch.log 2>&1
cat $out/one_launch_single_batch.log
