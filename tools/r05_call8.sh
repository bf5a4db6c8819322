#!/bin/bash
# eighth GPU call of round 5: screen_wg_fused_kernel A/B on configs[3] — round 4's two loops (p0b0), + the extents asked for a
# turn ahead (p1b0), + the queue looked at between the turns (p1b1 = commit 694e5ea, 0.415 ms in call 7 against round 4's 0.32)
out=gpurun_out/r05h; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],3), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for round in 1 2; do for v in p0b0 p1b0 p1b1; do cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_screen_wg_fused.log
cp variants/lib_p1b0.so yacrd_amd/lib/libyacrd_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fallback_queue or workgroup or screen" > $out/pytest_wg.log 2>&1; tail -3 $out/pytest_wg.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
