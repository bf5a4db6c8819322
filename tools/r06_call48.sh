#!/bin/bash
# forty-eighth GPU call of round 6: rocprofv3 kernel stats + PMC (VALU / LDS / traffic) of configs[1]'s batches at sigma = 300 and 100 on one engine
# (the build with the second looks, the ramp's mirror in): the evidence behind the jitter block's numbers
out=$(realpath -m gpurun_out/r06R); mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for j in 300 100; do
  wa=(--no-cpu-baseline --no-extras --weak --jitter $j --engines 1 --small-steps 100 --steps 100 --warmup 5)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$j" -o s -- python "$root/bench.py" "${wa[@]}" > "$out/stats_$j.log" 2>&1
  find "$out/stats_$j" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_configs1_sigma$j.csv" \;
  pass() { local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$j/$name" -o p -- python "$root/bench.py" "${wa[@]}" > "$out/pmc_${j}_$name.log" 2>&1; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  python3 "$root/tools/pmc_summary.py" "$out/pmc_$j" > "$out/pmc_summary_configs1_sigma$j.txt" 2>&1
  rm -rf "$out/stats_$j" "$out/pmc_$j"
done
head -6 $out/kernel_stats_configs1_sigma300.csv | cut -c1-150; head -30 $out/pmc_summary_configs1_sigma300.txt | cut -c1-200
