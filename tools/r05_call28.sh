#!/bin/bash
# twenty-eighth GPU call of round 5: one store per decided read, the count 0 in four bytes for a read with no bad region — parity, fuzz,
# configs[4] clamped / sigma 100 against the build before (variants/lib_prev.so)
out=gpurun_out/r05za; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -1 $out/pytest_parity_split.log
timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for args in "--config 4" "--config 4 --jitter 100"; do for v in keep prev keep prev keep prev; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v [$args]: "; timeout 900 python bench.py $args --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_one_store_per_read_2.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
