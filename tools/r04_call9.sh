#!/bin/bash
# ninth GPU call of round 4: slab sizes of the deferred sweep, CLI with --gpus N, tests
out=gpurun_out/r04i; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %s %.4f ms frac %.3f follow-on %s whole-path %.3f deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel"][-14:], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for v in s1024 s512 s256 s2048t512; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[2]: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done; done > $out/ab_slab.log 2>&1; cat $out/ab_slab.log
for v in s1024 s512 s256 s2048t512; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[4]: "; python bench.py --config 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/ab_slab_configs4.log 2>&1; cat $out/ab_slab_configs4.log
cp variants/libs1024.so yacrd_amd/lib/libyacrd_hip.so
timeout 900 python tools/edit_bench.py > $out/edit_bench.log 2>&1; cat $out/edit_bench.log
