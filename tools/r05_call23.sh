#!/bin/bash
# twenty-third GPU call of round 5: the build with the list-driven follow-on as the only one (slab kernels gone) — the whole -m gpu
# suite, parity with the long-batch path forced, fuzz (default / forced split / one launch), every read of configs[2] and [4] at full
# size, the default bench line, kernel stats + PMC of configs[4] and configs[2]
out=gpurun_out/r05w; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -2 $out/pytest_parity_split.log
timeout 300 python tools/gpu_fuzz.py 150 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_SPLIT_MIN_READS=0 timeout 400 python tools/gpu_fuzz.py 240 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
YACRD_FUZZ_ONE_LAUNCH=1 timeout 300 python tools/gpu_fuzz.py 150 > $out/fuzz_one_launch.log 2>&1; tail -1 $out/fuzz_one_launch.log
timeout 1500 python tools/scale_check.py 3 5 > $out/scale_configs_2_4.jsonl 2> $out/scale.err; cut -c1-200 $out/scale_configs_2_4.jsonl
YACRD_SYNTH_FLAGS=19206 timeout 600 python tools/scale_check.py 3 > $out/scale_jitter300_configs_2.jsonl 2> $out/scale300.err; cut -c1-200 $out/scale_jitter300_configs_2.jsonl
timeout 1200 python bench.py > $out/bench_default.log 2>$out/bench_default.err; tail -c 2600 $out/bench_default.log; cp bench_extras.json $out/ 2>/dev/null
PROFILE_WORKLOADS="configs4 configs2" bash tools/profile_r05.sh $out/prof > $out/profile.log 2>&1; head -9 $out/prof/kernel_stats_configs4.csv | cut -c1-150
