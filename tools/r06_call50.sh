#!/bin/bash
# fiftieth GPU call of round 6: window bins of 2 / 4 positions in the build with the second looks (YK_WIDE_WB: the windows reach 64 / 128
# positions with the same table; a and b resolved inside their bin by counting)
out=gpurun_out/r06T; mkdir -p $out
W='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(r["kernel_ms"]*1e3,2), "alone", round(r.get("kernel_alone_ms",0)*1e3,2), "frac alone", round(r.get("frac_alone",0),3), "one at a time", round(h["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), "deferred", h["deferred_reads"], h["paths"]["screen_wide"], d["parity"][:9])'
for v in base wb1 wb2 base wb2; do
  cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 300 100; do echo -n "== $v weak jitter $j: "; timeout 600 python bench.py --weak --jitter $j --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"; done
done 2>&1 | tee $out/wb.log
cp variants/lib_wb2.so yacrd_amd/lib/libyacrd_hip.so
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 120 2>&1 | tail -1 | tee -a $out/wb.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q 2>&1 | tail -2 | tee -a $out/wb.log
cp variants/lib_wb1.so yacrd_amd/lib/libyacrd_hip.so
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 60 2>&1 | tail -1 | tee -a $out/wb.log
cp variants/lib_base.so yacrd_amd/lib/libyacrd_hip.so
