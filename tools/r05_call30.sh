#!/bin/bash
# thirtieth GPU call of round 5: the other bench lines of the final build (--weak = configs[1] batches, --config 3, --config 2), for profiles/
out=gpurun_out/r05zc; mkdir -p $out
timeout 600 python bench.py --weak --no-cpu-baseline > $out/bench_weak_configs1.json 2>/dev/null; tail -c 400 $out/bench_weak_configs1.json; echo
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-extras > $out/bench_configs3.json 2>/dev/null; tail -c 400 $out/bench_configs3.json; echo
timeout 600 python bench.py --config 2 --no-cpu-baseline --no-extras > $out/bench_configs2.json 2>/dev/null; tail -c 400 $out/bench_configs2.json; echo
