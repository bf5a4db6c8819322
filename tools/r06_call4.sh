#!/bin/bash
# fourth GPU call of round 6: the persistent workgroup screen with CLAIMED shares (two entries a claim; one: variant) on
# configs[3]; the workgroup tests; the whole default bench line with the interval loads non-temporal
out=gpurun_out/r06d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "workgroup or fused or skewed or screen" 2>&1 | tail -4 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -2 | tee $out/fuzz_med.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), "phases", {k: round(v,4) for k,v in (h.get("phases_full_timing_ms") or {}).items()}, d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep claim1 keep claim1; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== cfg3 $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3_claims.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \;
rm -rf $out/prof
head -5 $out/kernel_stats_configs3.csv | cut -c1-150
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 2500 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
