#!/bin/bash
# tenth GPU call of round 4: larger slabs of the deferred sweep, full-size parity on the spread generator, fuzz soak
out=gpurun_out/r04j; mkdir -p $out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %s %.4f ms frac %.3f follow-on %s whole-path %.3f deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel"][-14:], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("deferred_reads_rank0"), d["parity"][:9]))'
for rep in 1 2; do for v in s2048t512 s2048t256 s4096t512 s4096t1024 s8192t1024; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[2]: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done; done > $out/ab_slab.log 2>&1; cat $out/ab_slab.log
for v in s2048t512 s4096t512 s4096t1024 s8192t1024; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== $v configs[4]: "; python bench.py --config 4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "$P"
done > $out/ab_slab_configs4.log 2>&1; cat $out/ab_slab_configs4.log
cp variants/libs2048t512.so yacrd_amd/lib/libyacrd_hip.so
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
YACRD_SYNTH_FLAGS=25602 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter100_configs_1_2.jsonl 2> $out/scale100.err; cut -c1-200 $out/scale_jitter100_configs_1_2.jsonl
YACRD_SYNTH_FLAGS=19206 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter300_configs_1_2.jsonl 2> $out/scale300.err; cut -c1-200 $out/scale_jitter300_configs_1_2.jsonl
timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
YACRD_SPLIT_MIN_READS=0 timeout 400 python tools/gpu_fuzz.py 300 > $out/fuzz_split0.log 2>&1; tail -1 $out/fuzz_split0.log
YACRD_FUZZ_ITEMS2=1 timeout 300 python tools/gpu_fuzz.py 200 > $out/fuzz_items2.log 2>&1; tail -1 $out/fuzz_items2.log
