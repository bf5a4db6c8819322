#!/bin/bash
# twenty-fifth GPU call of round 6: non-temporal interval loads only where the launch streams from HBM (the two-items build);
# the workgroup screen with / without them on configs[3]; configs[1]'s small batches; the default bench line
out=gpurun_out/r06y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "one launch", round(h["one_launch_single_batch"]["ms_per_batch"]*1e3,2), d["parity"][:9])'
for j in 0 300 0; do echo -n "== weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done 2>&1 | tee $out/weak.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep wgnt keep wgnt; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== cfg3 $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3_nt.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
for c in 4 2; do echo -n "== cfg$c: "; timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done | tee $out/cfg42.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1200 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
