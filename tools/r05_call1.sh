#!/bin/bash
# first GPU call of round 5: the compact driver line (bench test + the default run as the driver runs it)
out=gpurun_out/r05a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q > $out/pytest_bench.log 2>&1; tail -5 $out/pytest_bench.log
SECONDS=0; timeout 1200 python bench.py > $out/bench_default.stdout 2> $out/bench_default.stderr
echo "wall ${SECONDS}s"; tail -3 $out/bench_default.stderr
wc -c $out/bench_default.stdout; cat $out/bench_default.stdout
cp bench_extras.json $out/bench_extras.json
