#!/bin/bash
# twenty-fourth GPU call of round 6, the final build: the whole -m gpu suite, smoke, the default bench line, the judged profiles
# (kernel stats + PMC), a fuzz soak over five flag sets, every read of configs[1..4] at full size (tools/scale_check.py)
out=gpurun_out/r06x; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -8 | tee $out/gpu_suite.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $out/smoke.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1300 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
bash tools/profile_r06.sh $out/profiles > $out/profiles.log 2>&1; ls $out/profiles | head -3
{ timeout 200 python tools/gpu_fuzz.py 100; YACRD_FUZZ_MED=1 timeout 200 python tools/gpu_fuzz.py 100; YACRD_FUZZ_WIDE=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_ITEMS2=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_ONE_LAUNCH=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_SPLIT_MIN_READS=0 timeout 200 python tools/gpu_fuzz.py 60; } 2>&1 | grep gpu_fuzz | tee $out/fuzz_soak.log
timeout 1500 python tools/scale_check.py 2 3 4 5 > $out/scale_configs_1_2_3_4.jsonl 2> $out/scale.err; cut -c1-260 $out/scale_configs_1_2_3_4.jsonl
