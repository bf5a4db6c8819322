#!/bin/bash
# seventeenth GPU call of round 5: the FINAL build — smoke(), the whole -m gpu suite, a fuzz soak over the flag sets, every read of
# BASELINE configs[1..4] against the oracle at full size (clamped; sigma = 100 / 300 on configs[1..2]), the default bench line,
# kernel stats + PMC of configs[4] once more (plan / scan with eight reads per thread since the round's judged profiles)
out=gpurun_out/r05r; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest_gpu.log 2>&1; tail -14 $out/pytest_gpu.log
timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_wide.log 2>&1; tail -1 $out/fuzz_wide.log
YACRD_FUZZ_ITEMS2=1 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_items2.log 2>&1; tail -1 $out/fuzz_items2.log
YACRD_FUZZ_ONE_LAUNCH=1 timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz_one_launch.log 2>&1; tail -1 $out/fuzz_one_launch.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
timeout 1500 python tools/scale_check.py 2 3 4 5 > $out/scale_configs_1_2_3_4.jsonl 2> $out/scale.err; cut -c1-220 $out/scale_configs_1_2_3_4.jsonl
YACRD_SYNTH_FLAGS=25602 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter100_configs_1_2.jsonl 2> $out/scale100.err; cut -c1-220 $out/scale_jitter100_configs_1_2.jsonl
YACRD_SYNTH_FLAGS=19206 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter300_configs_1_2.jsonl 2> $out/scale300.err; cut -c1-220 $out/scale_jitter300_configs_1_2.jsonl
timeout 1200 python bench.py > $out/bench_default.log 2>$out/bench_default.err; tail -c 2600 $out/bench_default.log; cp bench_extras.json $out/ 2>/dev/null
PROFILE_WORKLOADS="configs4" bash tools/profile_r05.sh $out/prof > $out/profile.log 2>&1; head -8 $out/prof/kernel_stats_configs4.csv | cut -c1-150
