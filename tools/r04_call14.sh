#!/bin/bash
# fourteenth GPU call of round 4: the second looks of the screen inlined / behind a call / absent, configs[1] batches
out=gpurun_out/r04n; mkdir -p $out
Q='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["headline"]; r=d["roofline"]; print("ms/batch %.5f single %.5f screen %.4f healthy %s deferred %s phases %s %s" % (d["ms_per_step"], h["unpredicted_single_batch"]["ms_per_batch"], r["kernel_ms"], h["healthy_reads"], h["deferred_reads"], {k: round(v,4) for k,v in h["phases_full_timing_ms"].items() if k in ("sweep_small_ms","compact_ms","fused_ms")}, d["parity"][:9]))'
for rep in 1 2; do for v in slideinline slidecall noslides; do cp variants/lib$v.so yacrd_amd/lib/libyacrd_hip.so
  for j in 0 30 100 300; do echo -n "== $v configs[1] jitter $j: "; python bench.py --weak --no-extras --no-cpu-baseline --jitter $j 2>/dev/null | python -c "$Q"; done
done; done > $out/ab_slides_call.log 2>&1; cat $out/ab_slides_call.log
cp variants/libslidecall.so yacrd_amd/lib/libyacrd_hip.so
python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest_call.log 2>&1; tail -2 $out/pytest_call.log
timeout 200 python tools/gpu_fuzz.py 120 > $out/fuzz_call.log 2>&1; tail -1 $out/fuzz_call.log
