#!/bin/bash
# fourteenth GPU call of round 5: screen_wg_read with five barriers a read — parity of the workgroup classes, configs[3]; then
# the whole -m gpu suite and the default bench line on this build
out=gpurun_out/r05n; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],3), d["parity"][:9])'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest_parity.log 2>&1; tail -2 $out/pytest_parity.log
for i in 1 2 3; do timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done | tee $out/configs3.log
timeout 300 python tools/gpu_fuzz.py 60 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 1200 python bench.py > $out/bench_default.log 2>$out/bench_default.err; tail -c 2500 $out/bench_default.log; cp bench_extras.json $out/ 2>/dev/null
