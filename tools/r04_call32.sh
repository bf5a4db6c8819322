#!/bin/bash
# thirty-second GPU call of round 4: the CLI with YACRD_F_ONE_LAUNCH on its engines — CLI / ingest / stream tests
out=gpurun_out/r04zf; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ingest.py tests/test_gpu_stream.py -q > $out/pytest_cli.log 2>&1; tail -3 $out/pytest_cli.log
