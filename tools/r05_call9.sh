#!/bin/bash
# ninth GPU call of round 5: screen_wg_fused_kernel with 16-byte pair loads (default build) and with the next read's intervals
# staged in LDS a turn ahead (global_load_lds_dwordx4, LDS-only barriers: variant `stage`) — parity, then configs[3]
out=gpurun_out/r05i; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],3), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep stage; do [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest_parity_$v.log 2>&1; tail -2 $out/pytest_parity_$v.log
  timeout 300 python tools/gpu_fuzz.py 60 > $out/fuzz_$v.log 2>&1; tail -1 $out/fuzz_$v.log
done
for round in 1 2; do for v in keep stage; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_screen_wg_stage.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
