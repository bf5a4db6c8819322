#!/bin/bash
# seventh GPU call of round 4: the trimming path of the deferred sweep
out=gpurun_out/r04g; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
YACRD_SPLIT_MIN_READS=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_cli.py -x -q > $out/pytest_split0.log 2>&1; tail -3 $out/pytest_split0.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_split0.log 2>&1; tail -1 $out/fuzz_split0.log
YACRD_SPLIT_MIN_READS=0 YACRD_FUZZ_ITEMS2=1 timeout 200 python tools/gpu_fuzz.py 100 > $out/fuzz_split0_items2.log 2>&1; tail -1 $out/fuzz_split0_items2.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %.4f ms frac %.3f follow-on %s whole-path %.3f healthy %s deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("healthy_reads_rank0"), h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for v in trim notrim; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30; do echo -n "== $v configs[2] jitter $j: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$P"; done
done; done > $out/ab_trim.log 2>&1; cat $out/ab_trim.log
for v in trim notrim; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[4]: "; python bench.py --config 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/ab_trim_configs4.log 2>&1; cat $out/ab_trim_configs4.log
cp variants/libtrim.so yacrd_amd/lib/libyacrd_hip.so
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -o s -- python $OLDPWD/bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1 )
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs2.csv \; ; rm -rf $out/stats; cat $out/kernel_stats_configs2.csv
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OLDPWD/$out/pmc2/sq -o p -- python $OLDPWD/bench.py --config 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1 )
python tools/pmc_summary.py $out/pmc2 > $out/pmc_sq_configs2.txt 2>&1; rm -rf $out/pmc2; grep -A9 "deferred_sweep" $out/pmc_sq_configs2.txt
