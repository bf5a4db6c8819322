#!/bin/bash
# thirtieth GPU call of round 6, the final library (dead fused-lane host code removed; device code as in call 29): suite, smoke, fuzz, bench line
out=gpurun_out/r06D; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -6 | tee $out/gpu_suite.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
{ timeout 200 python tools/gpu_fuzz.py 90; YACRD_FUZZ_MED=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_WIDE=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_ONE_LAUNCH=1 timeout 200 python tools/gpu_fuzz.py 45; } 2>&1 | grep gpu_fuzz | tee $out/fuzz_soak.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1000 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
