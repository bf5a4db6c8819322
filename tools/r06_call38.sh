#!/bin/bash
# thirty-eighth GPU call of round 6 (host only): the scrubb by threads, the old output removed BEFORE the clock starts; chunk sizes
out=gpurun_out/r06I; mkdir -p $out
export YACRD_EDIT_STATS=1 YACRD_EDIT_BENCH_WAYS=pread:turns,pread:pwrite YACRD_EDIT_BENCH_THREADS=1,2,3,4,6,8,16
timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_turns_shm.log
YACRD_EDIT_BENCH_DIR=/tmp timeout 900 python tools/edit_bench.py 2>&1 | tee $out/edit_turns_disk.log
export YACRD_EDIT_BENCH_WAYS=pread:turns YACRD_EDIT_BENCH_THREADS=4,6
for c in 1048576 4194304 67108864; do echo "== chunk $c"; YACRD_EDIT_CHUNK=$c timeout 600 python tools/edit_bench.py 2>&1; done | tee $out/edit_turns_chunks_shm.log
