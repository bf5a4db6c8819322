#!/bin/bash
# twenty-first GPU call of round 5: the follow-on step of long batches over a LIST of the marked reads (mark_list_kernel +
# deferred_list_kernel: the list dealt out evenly over a resident grid) against slab by slab (YACRD_DEFER_LIST=0) — parity with
# the long-batch path forced on every batch, fuzz, configs[4] / configs[2], kernel stats
out=gpurun_out/r05u; mkdir -p $out
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -2 $out/pytest_parity_split.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 120 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), "deferred", r.get("deferred_reads"), d["parity"][:9])'
for c in 4 2; do for v in 1 0 1 0; do echo -n "== list $v: "; YACRD_DEFER_LIST=$v timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done; done 2>&1 | tee $out/ab_defer_list.log
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/stats -o s -- python /root/repo/bench.py --config 4 --no-extras --no-cpu-baseline > /root/repo/$out/stats.log 2>&1 )
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs4_defer_list.csv \; ; rm -rf $out/stats; head -9 $out/kernel_stats_configs4_defer_list.csv | cut -c1-150
