#!/bin/bash
# second GPU call of round 6: the one-wavefront-per-read screen of the workgroup classes (screen_stream.h) — parity, fuzz,
# configs[3] against the workgroup screen, kernel stats
out=gpurun_out/r06b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 400 python tools/gpu_fuzz.py 150 2>&1 | tail -3 | tee $out/fuzz_med.log
timeout 300 python tools/gpu_fuzz.py 90 2>&1 | tail -3 | tee $out/fuzz.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), "phases", {k: round(v,4) for k,v in (h.get("phases_full_timing_ms") or {}).items()}, d["parity"][:9])'
for f in 0 65536 0 65536; do
  echo -n "== cfg3 flags $f: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras --flags $f 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3_stream.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \;
rm -rf $out/prof
head -12 $out/kernel_stats_configs3.csv | cut -c1-150
echo -n "== weak configs1: "; timeout 600 python bench.py --weak --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
