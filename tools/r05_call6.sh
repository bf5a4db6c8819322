#!/bin/bash
# sixth GPU call of round 5: the restructured deferred_sweep_kernel (32-lane groups, 8-byte entries, 3 workgroups per CU),
# the new CLI tests (whole scrubb output at 1/100, overlap-file editors), the ISA + litmus pins
out=gpurun_out/r05f; mkdir -p $out
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -3 $out/pytest_parity_split.log
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q > $out/pytest_cli.log 2>&1; tail -5 $out/pytest_cli.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz_split.log 2>&1; tail -2 $out/fuzz_split.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
for c in 2 4 2 4; do timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done 2>&1 | tee $out/bench_configs.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/stats -o s -- python /root/repo/bench.py --config 4 --no-extras --no-cpu-baseline > /root/repo/$out/stats.log 2>&1
find /root/repo/$out/stats -name "*kernel_stats.csv" -exec cp {} /root/repo/$out/kernel_stats_configs4.csv \; ; rm -rf /root/repo/$out/stats; head -8 /root/repo/$out/kernel_stats_configs4.csv
