#!/bin/bash
# forty-fifth GPU call of round 6: one rank's share of the headline at N = 2 / 4 / 8 on one GPU (what strong scaling can reach), and the
# follow-on step's two forms at the 1/8 share
out=gpurun_out/r06O; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("reads", d["headline"].get("reads_rank0", "?"), "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "G reads/s", round(d["value"]/1e9,3), "follow_on_ms", d["headline"].get("follow_on_ms"), d["parity"][:9])'
for share in 1 2 4 8; do
  R=$((5000000/share)); O=$((500000000/share))
  echo -n "== share 1/$share: "; timeout 900 python bench.py --reads $R --overlaps $O --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/shares.log
for m in 0 400000 1000000; do
  echo -n "== share 1/8, YACRD_SPLIT_MIN_READS=$m: "; YACRD_SPLIT_MIN_READS=$m timeout 900 python bench.py --reads 625000 --overlaps 62500000 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
  echo -n "== share 1/4, YACRD_SPLIT_MIN_READS=$m: "; YACRD_SPLIT_MIN_READS=$m timeout 900 python bench.py --reads 1250000 --overlaps 125000000 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee -a $out/shares.log
