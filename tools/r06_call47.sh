#!/bin/bash
# forty-seventh GPU call of round 6: windows that jump to the next event instead of sliding by W (the build with the second looks)
out=gpurun_out/r06Q; mkdir -p $out
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], "healthy", h["healthy_reads"], h["paths"]["screen_wide"], d["parity"][:9])'
for v in base jump base jump; do
  cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 300 100 30; do echo -n "== $v weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
done 2>&1 | tee $out/jump.log
cp variants/lib_jump.so yacrd_amd/lib/libyacrd_hip.so
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "deferred", h["deferred_reads_rank0"], d["parity"][:9])'
echo -n "== jump cfg2 jitter 300: "; timeout 900 python bench.py --config 2 --jitter 300 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q" | tee -a $out/jump.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_one_launch.py -x -q 2>&1 | tail -2 | tee -a $out/jump.log
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 150 2>&1 | tail -1 | tee -a $out/jump.log
timeout 200 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee -a $out/jump.log
cp variants/lib_base.so yacrd_amd/lib/libyacrd_hip.so
