#!/bin/bash
# thirty-seventh GPU call of round 6 (host only): where the chunk-parallel scrubb's seconds go (YACRD_EDIT_STATS)
out=gpurun_out/r06I; mkdir -p $out
export YACRD_EDIT_STATS=1 YACRD_EDIT_BENCH_WAYS=pread:turns,pread:pwrite YACRD_EDIT_BENCH_THREADS=2,4,8,16
timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_stats_shm.log
YACRD_EDIT_BENCH_DIR=/tmp timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_stats_disk.log
free -g | tee -a $out/edit_stats_disk.log; cat /proc/sys/vm/dirty_ratio /proc/sys/vm/dirty_background_ratio | tee -a $out/edit_stats_disk.log
