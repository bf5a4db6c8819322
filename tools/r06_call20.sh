#!/bin/bash
# twentieth GPU call of round 6: batches with device-wide reads predicted; the bench line with the box block
out=gpurun_out/r06t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q 2>&1 | tail -3 | tee $out/parity.log
timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee $out/fuzz.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee $out/fuzz_med.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), h["paths"], d["parity"][:9])'
for i in 1 2 3; do echo -n "== cfg3: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done | tee $out/cfg3.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1100 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
