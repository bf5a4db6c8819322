#!/bin/bash
# seventh GPU call of round 5 (first of the re-created container): the whole -m gpu suite on the rebuilt library, the default
# bench line, and an experiment: long batches pipelined over 1 / 2 / 3 engines on one device — how much does running one
# batch's plan / deferred sweep / scan beside another batch's screen buy (the bound of any second-stream overlap inside a batch)
out=gpurun_out/r05g; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for ne in 1 2 3; do
  echo -n "engines $ne: "
  timeout 600 python bench.py --weak --reads 1000000 --overlaps 100000000 --coverage 3 --engines $ne --small-steps 40 --small-warmup 4 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
def walk(o, p=""):
    if isinstance(o, dict):
        for k, v in o.items(): walk(v, p + "/" + k)
    elif isinstance(o, (int, float)) and ("ms" in p or "us" in p.split("/")[-1]) and "per" in p: print(p, round(o, 4), end="; ")
walk(d); print()'
done 2>&1 | tee $out/engines_long_batches.log
timeout 900 python bench.py > $out/bench_default.log 2>$out/bench_default.err; tail -c 3000 $out/bench_default.log; cp bench_extras.json $out/ 2>/dev/null
