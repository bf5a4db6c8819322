#!/bin/bash
# second GPU call of round 4: the split follow-on kernels — parity forced on every batch size, fuzz, A/B against the one-dispatch form
out=gpurun_out/r04b; mkdir -p $out
YACRD_SPLIT_MIN_READS=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_cli.py -x -q > $out/pytest_split0.log 2>&1; tail -3 $out/pytest_split0.log
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz_split0.log 2>&1; tail -2 $out/fuzz_split0.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %.4f ms frac %.3f follow-on %s whole-path %.3f deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for m in 0 99999999999; do for cfg in 4 2; do for j in 0 30; do
  echo -n "== split_min $m configs[$cfg] jitter $j: "; YACRD_SPLIT_MIN_READS=$m python bench.py --config $cfg --steps 10 --warmup 3 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$P"
done; done; done; done > $out/ab_split.log 2>&1; cat $out/ab_split.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/batch %.5f single %.5f phases %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2; do for m in 0 99999999999; do for j in 0 100; do
  echo -n "== split_min $m configs[1] jitter $j: "; YACRD_SPLIT_MIN_READS=$m python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$Q"
done; done; done > $out/ab_split_small.log 2>&1; cat $out/ab_split_small.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -o s -- python $OLDPWD/bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1; cd $OLDPWD
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs2_split.csv \; ; rm -rf $out/stats; cat $out/kernel_stats_configs2_split.csv
