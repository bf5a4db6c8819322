#!/bin/bash
# thirty-sixth GPU call of round 6 (host only): the chunk-parallel scrubb with its threads taking turns at the output file
out=gpurun_out/r06I; mkdir -p $out
timeout 1200 python tools/edit_bench.py 2>&1 | tee $out/edit_bench_shm.log
YACRD_EDIT_BENCH_DIR=/tmp timeout 1200 python tools/edit_bench.py 2>&1 | tee $out/edit_bench_disk.log
