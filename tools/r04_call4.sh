#!/bin/bash
# fourth GPU call of round 4: fused workgroup screen, compressed inputs, ramp + slides, editors at full size
out=gpurun_out/r04d; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
YACRD_SPLIT_MIN_READS=0 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q > $out/pytest_split0.log 2>&1; tail -3 $out/pytest_split0.log
timeout 200 python tools/gpu_fuzz.py 120 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %.4f ms frac %.3f follow-on %s whole-path %.3f healthy %s deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("healthy_reads_rank0"), h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for v in cur d2occ5 noslide2; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 100; do echo -n "== $v configs[2] jitter $j: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$P"; done
done; done > $out/ab_screen.log 2>&1; cat $out/ab_screen.log
for v in cur d2occ5 noslide2; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[4]: "; python bench.py --config 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/ab_screen_configs4.log 2>&1; cat $out/ab_screen_configs4.log
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/batch %.5f single %.5f healthy %s deferred %s phases %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], h["healthy_reads"], h["deferred_reads"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2; do for v in cur slides6; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so; for j in 0 100 300; do
  echo -n "== $v configs[1] jitter $j: "; python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$Q"
done; done; done > $out/ab_slides_small.log 2>&1; cat $out/ab_slides_small.log
cp variants/libcur.so yacrd_amd/lib/libyacrd_hip.so
S='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; print("ms/step %.4f phases %s %s" % (d["ms_per_step"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items()}, d["parity"][:9]))'
for rep in 1 2; do for fl in 0 1048576; do echo -n "== configs[3] flags $fl: "; python bench.py --config 3 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --flags $fl 2>/dev/null | python -c "$S"; done; done > $out/ab_skewed.log 2>&1; cat $out/ab_skewed.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats3 -o s -- python $OLDPWD/bench.py --config 3 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1 )
find $out/stats3 -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \; ; rm -rf $out/stats3; cat $out/kernel_stats_configs3.csv
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OLDPWD/$out/pmc2/sq -o p -- python $OLDPWD/bench.py --config 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1 )
python tools/pmc_summary.py $out/pmc2 > $out/pmc_sq_configs2.txt 2>&1; rm -rf $out/pmc2; grep -A9 "deferred_sweep\|scan_compact" $out/pmc_sq_configs2.txt
timeout 1500 python tools/e2e_scrubb_full.py > $out/e2e_scrubb_full.log 2>&1; cat $out/e2e_scrubb_full.log
timeout 600 python tools/e2e_cli_paf.py > $out/e2e_cli_paf.log 2>&1; cat $out/e2e_cli_paf.log
