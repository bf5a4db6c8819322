#!/bin/bash
# eighth GPU call of round 6: what separates screen_wg_kernel (0.128 ms on configs[3]) from the fused launch's screening
# (0.19-0.21): the grid (YACRD_WGK_GRID on the chain's kernel), the footprint (the fused launch bare: no fallback code, 8 KB LDS)
out=gpurun_out/r06h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
prof() { # name flags env...
  local name=$1 flags=$2; shift 2
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 --flags $flags > $out/prof_$name.log 2>&1
  find $out/prof_$name -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_$name.csv \;
  rm -rf $out/prof_$name
  echo "== $name"; grep -E "screen_wg|sweep_lds" $out/kernel_stats_$name.csv | cut -d, -f1,2,4,6,7 | cut -c1-140
}
for g in 512 1024 2048 4096 9953; do prof chain_grid$g 1048576 YACRD_WGK_GRID=$g; done
for v in bare nofilt keep; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  for sh in 1 2 8; do prof ${v}_share$sh 0 YACRD_FUSED_SHARE=$sh; done
done
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
