#!/bin/bash
# eighteenth GPU call of round 5: the screen with THREE groups of list entries per wavefront (YACRD_SCREEN_ITEMS=3) at occupancy 6
# (44 bytes of scratch) and 5 (none) against the two-items build, configs[4] and configs[2]
out=gpurun_out/r05s; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for c in 4 2; do for v in "keep 2" "keep 3" "items3occ5 3" "keep 2" "keep 3" "items3occ5 3"; do set -- $v; cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $1 = keep ] || cp variants/lib_$1.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $1 items $2: "; YACRD_SCREEN_ITEMS=$2 timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_items3.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
