#!/bin/bash
# tools/pmc.sh <outdir> -- <bench args>: rocprofv3 PMC passes for the bench (run on the GPU box).
# Counters go in their own runs (no sys/hip traces), as the MI355X guide prescribes.
set -u
out=$(realpath -m "$1"); shift; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o p -- \
      python /root/repo/bench.py --no-cpu-baseline "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 /root/repo/tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
