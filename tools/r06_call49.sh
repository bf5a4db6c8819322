#!/bin/bash
# forty-ninth GPU call of round 6: the line with frac_alone in the pipelined configs[1] blocks (jitter[k][6]); the bench tests; the default line
out=gpurun_out/r06S; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3 | tee $out/bench_tests.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1200 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
