#!/bin/bash
# third GPU call of round 5: the filtered exact sweep in deferred_sweep_kernel (two-kernel follow-on forced), the fused
# screen's give-up path, configs[3] back at its time
out=gpurun_out/r05c; mkdir -p $out
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -5 $out/pytest_parity_split.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gives_up or fallback_queue" > $out/pytest_giveup.log 2>&1; tail -5 $out/pytest_giveup.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 150 > $out/fuzz_split.log 2>&1; tail -2 $out/fuzz_split.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "frac", round(r["frac"],4), "follow_on", r.get("finish_compact_kernel_ms"), "phases", h.get("phases_full_timing_ms"), d["parity"])'
for c in 3 4 2; do timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done 2>&1 | tee $out/bench_configs.log
timeout 600 python bench.py --config 2 --jitter 100 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q" 2>&1 | tee -a $out/bench_configs.log
