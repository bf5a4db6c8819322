#!/bin/bash
# fifth GPU call of round 5: the litmus tests; deferred_sweep_kernel A/B: filtered (2048 / 512), filtered (1024 / 256), round 4's
out=gpurun_out/r05e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_litmus.py -x -q -s > $out/pytest_litmus.log 2>&1; tail -8 $out/pytest_litmus.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in base slab1024 nofilter; do cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for c in 2 4; do echo -n "== $v: "; timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done
done 2>&1 | tee $out/ab_deferred.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
