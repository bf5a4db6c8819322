#!/bin/bash
# tools/profile_r06.sh <outdir> — the round's judged profiles, on the GPU box:
#   * rocprofv3 --kernel-trace --stats of the commands the bench lines come from (headline configs[4], configs[2],
#     configs[3], the small batches of configs[1] on one engine); PROFILE_WORKLOADS="configs4 ..." picks a subset;
#   * PMC passes, each in a run of its own with --kernel-trace only (MI355X_MICROARCH.md, HBM section):
#     FETCH_SIZE, WRITE_SIZE (traffic) and two SQ sets, per workload;
#   * the per-kernel summaries (tools/pmc_summary.py).
# Copy what is to be judged from <outdir> into profiles/.
set -u
out=$(realpath -m "$1"); mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
common=(--no-cpu-baseline --no-extras)
declare -A W
W[configs4]="--config 4 --steps 20 --warmup 3"
W[configs2]="--config 2 --steps 20 --warmup 3"
W[configs1]="--weak --engines 1 --small-steps 100 --steps 100 --warmup 5"
W[configs3]="--config 3 --steps 20 --warmup 3"
for w in ${PROFILE_WORKLOADS:-configs4 configs2 configs3 configs1}; do
  read -r -a wa <<< "${W[$w]}"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$w" -o s -- \
      python "$root/bench.py" "${common[@]}" "${wa[@]}" > "$out/stats_$w.log" 2>&1
  cp "$out/stats_$w"/*/s_kernel_stats.csv "$out/kernel_stats_$w.csv" 2>/dev/null || \
      find "$out/stats_$w" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_$w.csv" \;
  pass() { local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$w/$name" -o p -- \
        python "$root/bench.py" "${common[@]}" "${wa[@]}" > "$out/pmc_${w}_$name.log" 2>&1; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  python3 "$root/tools/pmc_summary.py" "$out/pmc_$w" > "$out/pmc_summary_$w.txt" 2>&1
  rm -rf "$out/stats_$w" "$out/pmc_$w"
done
ls -la "$out"
