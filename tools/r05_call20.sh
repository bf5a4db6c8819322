#!/bin/bash
# twentieth GPU call of round 5: screen_wg_fused_kernel with the next read's extent and length asked for a turn ahead AND kept per
# lane (variant `ahead`: one round trip a turn, not three) against the committed kernel — parity of the workgroup classes, configs[3]
out=gpurun_out/r05t; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],3), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
cp variants/lib_ahead.so yacrd_amd/lib/libyacrd_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "workgroup or screen or fallback or skew or medium or size_class" > $out/pytest_wg_ahead.log 2>&1; tail -2 $out/pytest_wg_ahead.log
for round in 1 2 3; do for v in keep ahead; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_screen_wg_ahead.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
