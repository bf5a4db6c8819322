#!/bin/bash
# twelfth GPU call of round 4: the N > 1 bench path at FULL size on the one-GPU box (two and four ranks on device 0, gloo)
out=gpurun_out/r04l; mkdir -p $out
for n in 2 4; do
  ( time YACRD_BENCH_DEVICE=0 YACRD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 10 --warmup 3 > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err ) 2> $out/bench_gpus$n.time
  tail -3 $out/bench_gpus$n.time; tail -c 300 $out/bench_gpus$n.err
  python - <<PY
import json
d=json.loads(open("$out/bench_gpus$n.json").read().strip().splitlines()[-1])
h=d["headline"]
print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["torch_distributed_backend"], d["parity"], h["generate_s"], [ (p["reads"], round(p["ms_per_step"],3)) for p in h["per_rank"]], h["interval_imbalance_max_over_min"])
PY
done
ls /dev/shm | head
