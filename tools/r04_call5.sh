#!/bin/bash
# fifth GPU call of round 4: the hole closed form in the screen, editors' view path at full size
out=gpurun_out/r04e; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
YACRD_SPLIT_MIN_READS=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q > $out/pytest_split0.log 2>&1; tail -3 $out/pytest_split0.log
timeout 200 python tools/gpu_fuzz.py 120 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
YACRD_FUZZ_ITEMS2=1 timeout 200 python tools/gpu_fuzz.py 90 > $out/fuzz_items2.log 2>&1; tail -1 $out/fuzz_items2.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %.4f ms frac %.3f follow-on %s whole-path %.3f healthy %s deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("healthy_reads_rank0"), h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for v in hole nohole; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30; do echo -n "== $v configs[2] jitter $j: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$P"; done
done; done > $out/ab_hole.log 2>&1; cat $out/ab_hole.log
for v in hole nohole; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[4]: "; python bench.py --config 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/ab_hole_configs4.log 2>&1; cat $out/ab_hole_configs4.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/batch %.5f single %.5f healthy %s deferred %s screened %s phases %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], h["healthy_reads"], h["deferred_reads"], h["batches_through_the_screen"][:10], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2; do for v in hole nohole; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; for a in "--jitter 0" "--jitter 100" "--chimeras 10" "--chimeras 40"; do
  echo -n "== $v configs[1] $a: "; python bench.py --weak --no-extras --no-cpu-baseline $a 2>/dev/null | python -c "$Q"
done; done; done > $out/ab_hole_small.log 2>&1; cat $out/ab_hole_small.log
cp variants/libhole.so yacrd_amd/lib/libyacrd_hip.so
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -o s -- python $OLDPWD/bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1 )
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs2.csv \; ; rm -rf $out/stats; cat $out/kernel_stats_configs2.csv
timeout 1500 python tools/e2e_scrubb_full.py > $out/e2e_scrubb_full.log 2>&1; cat $out/e2e_scrubb_full.log
