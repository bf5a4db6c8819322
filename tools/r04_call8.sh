#!/bin/bash
# eighth GPU call of round 4: the final build — tests, editors through the shared mapping, the one-item build on wide spreads, the driver's line
out=gpurun_out/r04h; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 900 python tools/edit_bench.py > $out/edit_bench.log 2>&1; cat $out/edit_bench.log
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; h=d["headline"]; print("reads %d ms/step %.4f screen %s %.4f ms frac %.3f follow-on %s whole-path %.3f healthy %s deferred %s %s" % (h["reads"], d["ms_per_step"], r["kernel"][-14:], r["kernel_ms"], r["frac"], r.get("finish_compact_kernel_ms"), h["whole_path_frac_of_peak"], h.get("healthy_reads_rank0"), h.get("deferred_reads_rank0"), d["parity"][:9]))'
for fl in 0 524288 262144; do for j in 100 300; do echo -n "== flags $fl configs[2] jitter $j: "; python bench.py --config 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --jitter $j --flags $fl 2>/dev/null | python -c "$P"; done; done > $out/ab_items_adaptive.log 2>&1; cat $out/ab_items_adaptive.log
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; tail -3 $out/bench_default.time
timeout 1500 python tools/e2e_scrubb_full.py > $out/e2e_scrubb_full.log 2>&1; cat $out/e2e_scrubb_full.log
