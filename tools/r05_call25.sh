#!/bin/bash
# twenty-fifth GPU call of round 5: the last rebuild (64-bit share arithmetic in deferred_list_kernel) — smoke, the -m gpu suite, parity
# with the long-batch path forced, a fuzz soak over the flag sets
out=gpurun_out/r05x; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
YACRD_SPLIT_MIN_READS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_one_launch.py -x -q > $out/pytest_parity_split.log 2>&1; tail -1 $out/pytest_parity_split.log
timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_SPLIT_MIN_READS=0 timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
YACRD_FUZZ_WIDE=1 YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_wide_split.log 2>&1; tail -1 $out/fuzz_wide_split.log
YACRD_FUZZ_ITEMS2=1 YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_items2_split.log 2>&1; tail -1 $out/fuzz_items2_split.log
timeout 900 python bench.py --config 4 --no-extras --no-cpu-baseline > $out/bench_configs4.log 2>&1; tail -c 900 $out/bench_configs4.log
