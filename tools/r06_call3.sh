#!/bin/bash
# third GPU call of round 6: does the workgroup screen scale with the number of workgroups a CU holds?  screen_wg_kernel
# (the three-launch chain's, YACRD_F_NO_FUSED_SCREEN) at two (84 VGPRs) and three (80, YK_WGK_OCC=6) workgroups per CU
out=gpurun_out/r06c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep wgk6; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 --flags 1048576 > $out/prof_$v.log 2>&1
  find $out/prof_$v -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_chain_$v.csv \;
  rm -rf $out/prof_$v
  echo "== $v"; head -6 $out/kernel_stats_chain_$v.csv | cut -c1-120
done
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
