#!/bin/bash
# first GPU call of round 4: tests, the default bench line, size sweep of the headline profile, configs[4] profile, W = 64 A/B
out=gpurun_out/r04a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; tail -3 $out/bench_default.time
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %.4f ms frac %.3f finish %s plan %s healthy %s deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), (h.get("phases_full_timing_ms") or {}).get("plan_ms"), h.get("healthy_reads_rank0"), h.get("deferred_reads_rank0"), d["parity"][:9]))'
for m in 1 2 3 4 5; do
  python bench.py --config 4 --reads $((m*1000000)) --overlaps $((m*100000000)) --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/size_sweep.log 2>&1; cat $out/size_sweep.log
PROFILE_WORKLOADS=configs4 bash tools/profile_r04.sh $out/prof > $out/prof.log 2>&1; tail -3 $out/prof.log
for rep in 1 2; do for v in base w64; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30 100 300; do echo -n "== $v jitter $j: "; python bench.py --config 2 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$P"; done
done; done > $out/ab_w64.log 2>&1; cat $out/ab_w64.log
