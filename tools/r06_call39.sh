#!/bin/bash
# thirty-ninth GPU call of round 6: the scrubb with 4 MB chunks by threads (/dev/shm, disk), then configs[4] as the config says through the CLI
out=gpurun_out/r06I; mkdir -p $out
export YACRD_EDIT_STATS=1 YACRD_EDIT_BENCH_WAYS=pread:turns,pread:pwrite YACRD_EDIT_BENCH_THREADS=2,3,4,6,8
timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_4mb_shm.log
YACRD_EDIT_BENCH_DIR=/tmp timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_4mb_disk.log
unset YACRD_EDIT_STATS
timeout 1500 python tools/e2e_scrubb_full.py 2>&1 | tee $out/e2e_scrubb_full_turns.log
