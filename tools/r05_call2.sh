#!/bin/bash
# second GPU call of round 5: the whole -m gpu suite on the ADVICE r4 fixes (queue without residency, one-launch gate) + a short fuzz
out=gpurun_out/r05b; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest_parity_split.log 2>&1; tail -3 $out/pytest_parity_split.log
timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz.log 2>&1; tail -3 $out/fuzz.log
timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("configs3 ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["parity"])'
