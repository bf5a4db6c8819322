#!/bin/bash
# usage: tools/ab_cfg3.sh v1 v2 ... : for each variants/lib<v>.so configs[3] (skewed) resident
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/step %.4f phases %s %s" % (d["ms_per_step"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2; do for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs3: "; python bench.py --config 3 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-north-star 2>/dev/null | python -c "$P"
done; done
