#!/bin/bash
# fifteenth GPU call of round 5: the group parser without the count pass (tests), then the round's judged profiles
# (tools/profile_r05.sh: kernel stats + PMC passes of configs[4], [2], [3], [1])
out=gpurun_out/r05p; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ingest_group.py tests/test_gpu_cli.py -x -q --durations=5 > $out/pytest_group_cli.log 2>&1; tail -12 $out/pytest_group_cli.log
bash tools/profile_r05.sh $out/prof > $out/profile.log 2>&1; tail -25 $out/profile.log
