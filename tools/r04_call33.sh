#!/bin/bash
# thirty-third GPU call of round 4: the CLI tests with and without the one-launch engines, same box, after a warm-up
out=gpurun_out/r04zg; mkdir -p $out
python -c "import torch, yacrd_amd" > /dev/null 2>&1
for rep in 1 2; do
YACRD_CLI_NO_ONE_LAUNCH=1 timeout 900 python -m pytest tests/test_gpu_cli.py -q 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_cli.py -q 2>&1 | tail -1
done > $out/cli_ab.log 2>&1; cat $out/cli_ab.log
