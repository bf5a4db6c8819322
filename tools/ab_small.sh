#!/bin/bash
# usage: tools/ab_small.sh v1 v2 ... : for each variants/lib<v>.so the configs[1] batches pipelined over three engines and
# on one engine with every launch timed (A/B of the small kernels; run on the GPU box)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("value %.4g ms/step %.4f screen %.5f ms frac %.3f whole %.3f single %.4f finish %.5f %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], h.get("whole_path_frac_of_peak",0), (h.get("unpredicted_single_batch") or {}).get("ms_per_batch",0), r.get("finish_compact_kernel_ms") or 0, d["parity"][:9]))'
for rep in 1 2; do
for v in "$@"; do [ "$v" = cur ] || cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v pipelined: "; python bench.py --weak --no-cpu-baseline --no-extras --steps 500 --warmup 20 2>/dev/null | python -c "$P"
  echo -n "== $v one engine: "; python bench.py --weak --no-cpu-baseline --no-extras --steps 300 --warmup 20 --engines 1 --time-every-launch 2>/dev/null | python -c "$P"
done; done
