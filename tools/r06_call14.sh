#!/bin/bash
# fourteenth GPU call of round 6: the workgroup classes as TWO launches (the screen alone, then table + filtered sweep + sort over what it left);
# side stream beside the workgroup classes': the whole parity file, fuzz, configs[3]
out=gpurun_out/r06n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -2 | tee $out/fuzz_med.log
timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -2 | tee $out/fuzz.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "phases", {k: round(v,4) for k,v in (h.get("phases_full_timing_ms") or {}).items()}, d["parity"][:9])'
for sh in 0 0 0; do
  echo -n "== cfg3: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \;
rm -rf $out/prof
head -8 $out/kernel_stats_configs3.csv | cut -c1-150
W='import sys,json; d=json.loads(sys.stdin.readline()); print("pipelined us", round(d["ms_per_step"]*1e3,2), "one at a time", round(d["headline"]["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "one launch", round(d["headline"]["one_launch_single_batch"]["ms_per_batch"]*1e3,2), d["parity"][:9])'
for j in 0 0; do
  echo -n "== weak configs1: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"
done 2>&1 | tee $out/weak.log
