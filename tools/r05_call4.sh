#!/bin/bash
# fourth GPU call of round 5: kernel stats + PMC of configs[2] with the filtered deferred sweep
PROFILE_WORKLOADS="configs2" bash tools/profile_r05.sh gpurun_out/r05d > gpurun_out/r05d_profile.log 2>&1
cat gpurun_out/r05d/kernel_stats_configs2.csv | head -12
grep -A12 "deferred_sweep" gpurun_out/r05d/pmc_summary_configs2.txt | head -40
