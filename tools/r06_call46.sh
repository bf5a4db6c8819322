#!/bin/bash
# forty-sixth GPU call of round 6: finish_compact_kernel (short batches' follow-on) with 128 registers (no spills, one workgroup per CU) and with its
# phase A through the filtered exact sweep, two marked reads per wavefront and turn
out=gpurun_out/r06P; mkdir -p $out
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], d["parity"][:9])'
for v in base occ4 filt4 filt8 base occ4 filt4 filt8; do
  cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 300; do echo -n "== $v weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
  echo -n "== $v weak 390000 reads: "; timeout 600 python bench.py --weak --reads 390000 --overlaps 19500000 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"
done 2>&1 | tee $out/finish.log
cp variants/lib_filt4.so yacrd_amd/lib/libyacrd_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q 2>&1 | tail -2 | tee -a $out/finish.log
timeout 200 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee -a $out/finish.log
YACRD_FUZZ_WIDE=1 timeout 200 python tools/gpu_fuzz.py 40 2>&1 | tail -1 | tee -a $out/finish.log
cp variants/lib_base.so yacrd_amd/lib/libyacrd_hip.so
