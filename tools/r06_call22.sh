#!/bin/bash
# twenty-second GPU call of round 6: short batches with the long batches' follow-on (list + filtered sweep + scan: YACRD_SPLIT_MIN_READS=0)
out=gpurun_out/r06v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "phases", {k: round(v*1e3,1) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9])'
for sp in default 0 default 0; do for j in 0 300; do
  if [ $sp = default ]; then unset YACRD_SPLIT_MIN_READS; else export YACRD_SPLIT_MIN_READS=$sp; fi
  echo -n "== split $sp jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"
done; done 2>&1 | tee $out/split.log
