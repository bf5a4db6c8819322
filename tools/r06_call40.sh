#!/bin/bash
# fortieth GPU call of round 6: the scrubb with the combining writer by threads (/dev/shm, disk), then configs[4] through the CLI once more
out=gpurun_out/r06J; mkdir -p $out
export YACRD_EDIT_STATS=1 YACRD_EDIT_BENCH_WAYS=pread:turns,pread:pwrite YACRD_EDIT_BENCH_THREADS=1,2,3,4,6,8,16
timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_writer_shm.log
YACRD_EDIT_BENCH_WAYS=pread:turns YACRD_EDIT_BENCH_DIR=/tmp timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_writer_disk.log
unset YACRD_EDIT_STATS
timeout 1500 python tools/e2e_scrubb_full.py 2>&1 | tee $out/e2e_scrubb_full_writer.log
