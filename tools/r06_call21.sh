#!/bin/bash
# twenty-first GPU call of round 6: configs[1] at sigma 300 pipelined over 1 / 2 / 3 engines, by GPU_MAX_HW_QUEUES (is it the
# streams' mapping onto hardware queues that makes some boxes 2-5 x slower than others on this block?)
out=gpurun_out/r06u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
W='import sys,json; d=json.loads(sys.stdin.readline()); print("pipelined us", round(d["ms_per_step"]*1e3,2), "kernel us", round(d["roofline"]["kernel_ms"]*1e3,2), "one at a time", round(d["headline"]["unpredicted_single_batch"]["ms_per_batch"]*1e3,2), d["parity"][:9])'
for q in default 2 8; do for ne in 1 2 3; do for j in 300 0; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo -n "== queues $q engines $ne jitter $j: "; timeout 600 python bench.py --weak --jitter $j --engines $ne --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$W"
done; done; done 2>&1 | tee $out/queues.log
unset GPU_MAX_HW_QUEUES
python - <<'PY' | tee $out/box.log
import sys, types; sys.path.insert(0, ".")
import bench, torch
cx = types.SimpleNamespace(torch=torch, dev_index=0, dev=torch.device("cuda", 0))
print(bench.box_block(cx))
PY
