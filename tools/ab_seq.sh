#!/bin/bash
# usage: ab_seq.sh v...: configs[2] on one engine for each variants/lib<v>.so
S="python bench.py --profile sequel --reads 2000000 --overlaps 200000000 --steps 6 --warmup 2 --engines 1 --no-extras --no-cpu-baseline --time-every-launch"
for rep in 1 2; do for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; echo -n "== seq $v: "; $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"roofline\"][\"frac\"])"; done; done
