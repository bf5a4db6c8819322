#!/bin/bash
# thirty-first GPU call of round 4: fuzz soak of the final build (default flags / second looks forced / two-items build /
# one launch / the two-kernel follow-on forced)
out=gpurun_out/r04ze; mkdir -p $out
timeout 400 python tools/gpu_fuzz.py 330 > $out/fuzz_default.log 2>&1; tail -1 $out/fuzz_default.log
YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 240 > $out/fuzz_wide.log 2>&1; tail -1 $out/fuzz_wide.log
YACRD_FUZZ_ITEMS2=1 timeout 300 python tools/gpu_fuzz.py 240 > $out/fuzz_items2.log 2>&1; tail -1 $out/fuzz_items2.log
YACRD_FUZZ_ONE_LAUNCH=1 timeout 400 python tools/gpu_fuzz.py 330 > $out/fuzz_one_launch.log 2>&1; tail -1 $out/fuzz_one_launch.log
YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 240 > $out/fuzz_split.log 2>&1; tail -1 $out/fuzz_split.log
