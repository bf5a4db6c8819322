#!/bin/bash
# thirty-second GPU call of round 6: configs[1]'s pipelined batches over 3 / 4 / 5 / 6 engines (clamped, sigma 300)
out=gpurun_out/r06F; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
W='import sys,json; d=json.loads(sys.stdin.readline()); print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), d["parity"][:9])'
for ne in 3 4 5 6 3 4; do for j in 0 300; do
  echo -n "== engines $ne jitter $j: "; timeout 600 python bench.py --weak --jitter $j --engines $ne --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"
done; done 2>&1 | tee $out/engines.log
