#!/bin/bash
# forty-second GPU call of round 6: the ramp's mirror in the second looks (ends behind the largest start are in no block's count):
# parity, fuzz with the second looks forced, configs[1] at sigma 100 / 300, configs[2] at sigma 300
out=gpurun_out/r06L; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_one_launch.py -x -q 2>&1 | tail -3 | tee $out/parity.log
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 120 2>&1 | tail -1 | tee $out/fuzz_wide.log
timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee $out/fuzz.log
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], "healthy", h["healthy_reads"], h["paths"]["screen_wide"], d["parity"][:9])'
for j in 300 100 30 300 100; do echo -n "== weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done 2>&1 | tee $out/weak.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "deferred", h["deferred_reads_rank0"], h["paths"], d["parity"][:9])'
for j in 300 100; do echo -n "== cfg2 jitter $j: "; timeout 900 python bench.py --config 2 --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done | tee $out/cfg2.log
