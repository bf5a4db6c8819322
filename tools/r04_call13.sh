#!/bin/bash
# thirteenth GPU call of round 4: release / acquire on the scan words of the SHORT batches' one-kernel follow-on, A/B
out=gpurun_out/r04m; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/batch %.5f single %.5f phases %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2 3; do for v in relaxed relacq; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[1]: "; python bench.py --weak --no-extras --no-cpu-baseline 2>/dev/null | python -c "$Q"
  echo -n "== $v 390k reads: "; python bench.py --weak --no-extras --no-cpu-baseline --reads 390000 --overlaps 19500000 --small-steps 100 2>/dev/null | python -c "$Q"
done; done > $out/ab_relacq_short.log 2>&1; cat $out/ab_relacq_short.log
cp variants/librelaxed.so yacrd_amd/lib/libyacrd_hip.so
