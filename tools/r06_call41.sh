#!/bin/bash
# forty-first GPU call of round 6: the -m gpu suite on the final host library (the CLI's editors go through the one-writer path), smoke
out=gpurun_out/r06K; mkdir -p $out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $out/gpu_suite.log
