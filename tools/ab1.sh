#!/bin/bash
# usage: ab1.sh v1 v2 ... : configs[1] with ONE engine (kernels alone, no overlap) for each variants/lib<v>.so
B="python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 20 --engines 1"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("ms/step %.4f kernel_ms %.5f frac %.3f deferred %s %.4f  %s" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("deferred_reads"), r.get("deferred_kernel_ms",0), {k:round(v,4) for k,v in d["kernel_ms"].items() if not isinstance(v,dict)}))'
for v in "$@"; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; echo -n "== ont1 $v: "; $B | python -c "$P"; echo -n "== ont1 full $v: "; $B --full-timing | python -c "$P"; done
