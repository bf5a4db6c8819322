#!/bin/bash
# fifteenth GPU call of round 4: the second looks in a build of their own (sweep_small_fused_defer_wide_kernel) —
# the whole GPU suite, fuzz (default flags / wide forced), configs[1] and configs[2] per sigma, the default bench line,
# and the profiles of configs[1] again (its dominant kernel changed)
out=gpurun_out/r04o; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 150 python tools/gpu_fuzz.py 100 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_FUZZ_WIDE=1 timeout 150 python tools/gpu_fuzz.py 100 > $out/fuzz_wide.log 2>&1; tail -1 $out/fuzz_wide.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; r=d["roofline"]; print("ms/step %.5f single %s screen %.4f %s healthy %s deferred %s %s" % (d["ms_per_step"], (h.get("unpredicted_single_batch") or {}).get("ms_per_batch"), r["kernel_ms"], r.get("kernel"), h["healthy_reads"], h["deferred_reads"], d["parity"][:9]))'
for rep in 1 2; do
  for j in 0 30 100 300; do echo -n "== configs[1] jitter $j: "; python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$Q"; done
done > $out/ab_wide_build.log 2>&1
for j in 30 300; do echo -n "== configs[2] jitter $j: "; python bench.py --config 2 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$Q"; done >> $out/ab_wide_build.log 2>&1
cat $out/ab_wide_build.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json
PROFILE_WORKLOADS="configs1" bash tools/profile_r04.sh $out/prof > $out/profile.log 2>&1; tail -3 $out/profile.log
