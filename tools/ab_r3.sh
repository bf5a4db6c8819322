#!/bin/bash
# usage: tools/ab_r3.sh v1 v2 ... : for each variants/lib<v>.so — configs[1] batches (one engine, then pipelined over
# three) and configs[2], each from the clamped and the jittered generator
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("value %.4g ms/step %.4f kernel %s %.5f ms frac %.3f deferred %s finish+compact %s" % (d["value"], d["ms_per_step"], r["kernel"][-22:], r["kernel_ms"], r["frac"], r.get("deferred_reads"), r.get("finish_compact_kernel_ms")))'
B1="python bench.py --weak --no-cpu-baseline --no-extras --steps 300 --warmup 20 --engines 1 --time-every-launch"
B3="python bench.py --weak --no-cpu-baseline --no-extras --steps 500 --warmup 20"
S="python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline"
for v in "$@"; do [ "$v" = cur ] || cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30; do
    echo -n "== $v ont1 jitter $j: "; $B1 --jitter $j 2>/dev/null | python -c "$P"
    echo -n "== $v ont3 jitter $j: "; $B3 --jitter $j 2>/dev/null | python -c "$P"
    echo -n "== $v seq jitter $j: "; $S --jitter $j 2>/dev/null | python -c "$P"
  done
done
