#!/bin/bash
# twenty-third GPU call of round 6: plan_kernel writes (extent, read) records for the workgroup classes; screen_wg_kernel reads one record instead of list entry + offsets
out=gpurun_out/r06w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "workgroup or fused or skewed or screen or filtered" 2>&1 | tail -3 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 90 2>&1 | tail -1 | tee $out/fuzz_med.log
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
prof() { local name=$1 flags=$2; shift 2
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 --flags $flags > $out/prof_$name.log 2>&1
  find $out/prof_$name -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_$name.csv \;
  rm -rf $out/prof_$name
  echo "== $name"; grep -E "screen_wg" $out/kernel_stats_$name.csv | cut -d, -f1,2,4,6,7 | cut -c1-140
}
for v in keep keep; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  prof $v 0 A=1
done
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), d["parity"][:9])'
for i in 1 2; do echo -n "== cfg3: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done | tee $out/cfg3.log
