#!/bin/bash
# twenty-eighth GPU call of round 6: the spot checks inline on the live registers (8 / 4 candidates / none): configs[1] at sigma 300 / 100 / 30
out=gpurun_out/r06B; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q 2>&1 | tail -3 | tee $out/parity.log
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 90 2>&1 | tail -1 | tee $out/fuzz_wide.log
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep spot4 nospot keep spot4 nospot; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 300 100; do echo -n "== $v jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
done 2>&1 | tee $out/weak.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
