#!/bin/bash
# twenty-ninth GPU call of round 5: the final build's default bench line, and the judged profile of configs[4] once more
out=gpurun_out/r05zb; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1200 python bench.py > $out/bench_default.log 2>$out/bench_default.err; tail -c 2600 $out/bench_default.log; cp bench_extras.json $out/ 2>/dev/null
PROFILE_WORKLOADS="configs4" bash tools/profile_r05.sh $out/prof > $out/profile.log 2>&1; head -9 $out/prof/kernel_stats_configs4.csv | cut -c1-150
