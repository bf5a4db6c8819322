#!/bin/bash
# thirty-first GPU call of round 6: the bench tests (N = 2 ranks through the launcher and through bench.py itself) with the input's
# room check, the distributed CPU tests on the box
out=gpurun_out/r06E; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_distributed.py tests/test_bench_line.py -x -q 2>&1 | tail -3 | tee $out/bench_tests.log
df -h /dev/shm /tmp | tee $out/df.log
