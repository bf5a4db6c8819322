#!/bin/bash
# tools/pmc_quick3.sh <outdir> [bench args...] — two SQ counter passes (own runs, --kernel-trace only) over
# `bench.py --no-extras <bench args>` (default: the configs[2] headline; `--weak --engines 1` for configs[1]) and the
# per-kernel summary; run on the GPU box.
set -u
out=$(realpath -m "$1"); shift; mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
args=(--no-cpu-baseline --no-extras --steps 30 --warmup 3 "$@")
run() { local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o p -- \
      python "$root/bench.py" "${args[@]}" > "$out/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
python3 "$root/tools/pmc_summary.py" "$out" > "$out/pmc_summary.txt" 2>&1
grep -A 17 "fused_defer" "$out/pmc_summary.txt"
