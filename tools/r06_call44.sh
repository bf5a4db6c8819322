#!/bin/bash
# forty-fourth GPU call of round 6: the screen's windows at W = 64 against W = 32 where the dovetail ends are spread (the build with the second looks)
out=gpurun_out/r06N; mkdir -p $out
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], "healthy", h["healthy_reads"], h["paths"]["screen_wide"], d["parity"][:9])'
for v in base w64 base w64; do
  cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 300 100 30; do echo -n "== $v weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
done 2>&1 | tee $out/w64.log
cp variants/lib_w64.so yacrd_amd/lib/libyacrd_hip.so
YACRD_FUZZ_WIDE=1 timeout 200 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee -a $out/w64.log
cp variants/lib_base.so yacrd_amd/lib/libyacrd_hip.so
