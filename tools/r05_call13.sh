#!/bin/bash
# thirteenth GPU call of round 5: the filtered sweep sized by what it kept (1 / 2 / 4 keys per lane) against always 4 (ksel0);
# the group parser's last tests and the parser's sort on its own
out=gpurun_out/r05m; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ingest_group.py -x -q > $out/pytest_group.log 2>&1; tail -3 $out/pytest_group.log
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -2 $out/pytest_parity_split.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity.log 2>&1; tail -2 $out/pytest_parity.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 90 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for round in 1 2; do for v in keep ksel0; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for c in 2 4; do echo -n "== $v: "; timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"; done
done; done 2>&1 | tee $out/ab_filtered_ksel.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
