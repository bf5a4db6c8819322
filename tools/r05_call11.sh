#!/bin/bash
# eleventh GPU call of round 5: the N-engine device parser (yacrd_engines_ingest_overlaps) — its tests, the one-engine ingest
# tests on the refactored parse_range, the CLI tests (--gpus 2 / 3 take it now), configs[3] on the committed screen_wg kernel
out=gpurun_out/r05k; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ingest_group.py -x -q > $out/pytest_group.log 2>&1; tail -15 $out/pytest_group.log
timeout 1500 python -m pytest tests/test_gpu_ingest.py -x -q > $out/pytest_ingest.log 2>&1; tail -3 $out/pytest_ingest.log
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q > $out/pytest_cli.log 2>&1; tail -5 $out/pytest_cli.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],3), d["parity"][:9])'
timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q" | tee $out/configs3.log
