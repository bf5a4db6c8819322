#!/bin/bash
# twenty-sixth GPU call of round 5: why is the screen 9 % slower on configs[4] with the dovetail ends spread (sigma = 100)?  The two SQ
# passes and the kernel stats of `bench.py --config 4 --jitter 100`, to set beside profiles/r05_pmc_summary_configs4.txt (clamped)
out=$(realpath -m gpurun_out/r05y); mkdir -p $out; root=$(pwd)
cd /tmp && export TMPDIR=/tmp
args="--config 4 --jitter 100 --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python $root/bench.py $args > $out/stats.log 2>&1
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs4_sigma100.csv \; ; rm -rf $out/stats
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc/sq1 -o p -- python $root/bench.py $args > $out/pmc_sq1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc/sq2 -o p -- python $root/bench.py $args > $out/pmc_sq2.log 2>&1
python3 $root/tools/pmc_summary.py $out/pmc > $out/pmc_summary_configs4_sigma100.txt 2>&1; rm -rf $out/pmc
head -5 $out/kernel_stats_configs4_sigma100.csv | cut -c1-140; grep -A 18 "sweep_small_fused" $out/pmc_summary_configs4_sigma100.txt | head -22
