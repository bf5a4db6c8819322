#!/bin/bash
# sixteenth GPU call of round 6: the fallback launch with its sweep passes on registers; the resident blocks with 1 / 2 / 3
# passes in flight (--resident-engines) on configs[4], [2], [3]
out=gpurun_out/r06p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "workgroup or fused or skewed or screen or filtered" 2>&1 | tail -3 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee $out/fuzz_med.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "one engine", round(h.get("one_engine_one_pass_at_a_time_ms") or 0,4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
for c in 3 4 2; do for ne in 1 2 3 1 2; do
  echo -n "== cfg$c engines $ne: "; timeout 900 python bench.py --config $c --resident-engines $ne --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/engines.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \;
rm -rf $out/prof
head -4 $out/kernel_stats_configs3.csv | cut -c1-150
