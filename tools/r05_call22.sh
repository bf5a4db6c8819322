#!/bin/bash
# twenty-second GPU call of round 5: deferred_list_kernel by workgroup size and occupancy (512 / 256 threads, 6 / 5 wavefronts per SIMD)
out=gpurun_out/r05v; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for c in 2 4; do for v in keep t256o6 t512o5 t256o5 keep t256o6; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v: "; timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_list_kernel_shapes.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
