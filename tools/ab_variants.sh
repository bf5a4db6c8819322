#!/bin/bash
# usage: ab.sh v1 v2 ... : bench configs[1] twice and configs[2] once for each variants/lib<v>.so
B="python bench.py --no-cpu-baseline --no-extras --steps 500 --warmup 20"
S="python bench.py --profile sequel --reads 2000000 --overlaps 200000000 --steps 5 --warmup 2 --engines 1 --no-extras --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("value %.4g ms/step %.4f kernel_ms %.5f frac %.3f deferred %s %.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("deferred_reads"), r.get("deferred_kernel_ms",0)))'
for rep in 1 2; do for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; echo -n "== ont $v: "; $B | python -c "$P"; done; done
for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; echo -n "== seq $v: "; $S | python -c "$P"; done
