#!/bin/bash
# tenth GPU call of round 5: a long batch's screen in parts, the parts' deferred reads sorted on a second stream
# (deferred_list_kernel) — parity with the parts forced on small batches, then configs[4] / configs[2] by number of parts
out=gpurun_out/r05j; mkdir -p $out
YACRD_SPLIT_MIN_READS=0 YACRD_PART_MIN_BLOCKS=8 YACRD_SCREEN_PARTS=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parts.log 2>&1; tail -3 $out/pytest_parts.log
YACRD_SPLIT_MIN_READS=0 YACRD_PART_MIN_BLOCKS=8 YACRD_SCREEN_PARTS=3 timeout 300 python tools/gpu_fuzz.py 90 > $out/fuzz_parts.log 2>&1; tail -1 $out/fuzz_parts.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest_default.log 2>&1; tail -2 $out/pytest_default.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
for c in 4 2; do for k in 1 2 4 8 1 4; do echo -n "parts $k: "; YACRD_SCREEN_PARTS=$k timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done; done 2>&1 | tee $out/bench_parts.log
