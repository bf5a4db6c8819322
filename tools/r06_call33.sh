#!/bin/bash
# thirty-third GPU call of round 6: finish_compact_kernel's slab (reads per workgroup = its threads): 1024 / 512 / 256 on configs[1]
# clamped and at sigma 100 / 300 — at 1024 a batch of 100 000 reads is 98 workgroups: 158 of the device's 256 CUs sort nothing
out=gpurun_out/r06G; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "compact us", round(h["phases_full_timing_ms"]["compact_ms"]*1e3,1), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep sb512 sb256 keep sb512 sb256; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 100 300; do echo -n "== $v jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
done 2>&1 | tee $out/slab.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
