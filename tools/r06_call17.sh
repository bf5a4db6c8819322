#!/bin/bash
# seventeenth GPU call of round 6: the whole -m gpu suite, smoke, the default bench line, the judged profiles (kernel stats + PMC)
out=gpurun_out/r06q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -8 | tee $out/gpu_suite.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $out/smoke.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1500 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
bash tools/profile_r06.sh $out/profiles > $out/profiles.log 2>&1; ls $out/profiles
