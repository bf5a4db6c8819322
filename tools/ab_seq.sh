#!/bin/bash
# usage: tools/ab_seq.sh v1 v2 ... : for each variants/lib<v>.so configs[2] resident (clamped and jittered generator)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("value %.4g ms/step %.4f screen %.4f ms frac %.3f finish %.4f %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms") or 0, d["parity"][:9]))'
for rep in 1 2; do for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30; do echo -n "== $v jitter $j: "; python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-north-star --jitter $j 2>/dev/null | python -c "$P"; done
done; done
