#!/bin/bash
# nineteenth GPU call of round 6: screen_wg_kernel, one read per dispatcher-fed workgroup, at 2 / 3 workgroups per CU (YK_WGK_OCC)
out=gpurun_out/r06s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
prof() { local name=$1 flags=$2; shift 2
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 --flags $flags > $out/prof_$name.log 2>&1
  find $out/prof_$name -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_$name.csv \;
  rm -rf $out/prof_$name
  echo "== $name"; grep -E "screen_wg_kernel" $out/kernel_stats_$name.csv | cut -d, -f1,2,4,6,7 | cut -c1-140
}
for v in keep wgk6b wgk5 keep wgk6b; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  prof $v 0 A=1
done
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
