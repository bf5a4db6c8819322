#!/bin/bash
# ninth GPU call of round 6: + the inert zero-length interval at pmin, dead groups skipped in the workgroup screen; the new crafted test
# for what the screen leaves: parity, fuzz, how many reads still reach sweep_lds_read, configs[3]
out=gpurun_out/r06i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "workgroup or fused or skewed or screen or filtered" 2>&1 | tail -4 | tee $out/parity.log
YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 120 2>&1 | tail -2 | tee $out/fuzz_med.log
python - <<'PY' 2>&1 | tee $out/cfg3_counters.log
import numpy as np
import yacrd_amd, oracle
from yacrd_amd import host
off, iv, ln = host.synth_csr(host.SYNTH_SKEWED, 10000, 30000000, 20241108 + 4)
want = oracle.run(off, iv, ln.astype(np.uint64), 4, 0.4, n_threads=16)
with yacrd_amd.Engine(flags=yacrd_amd.F_TIMING_FULL) as e:
    for _ in range(3):
        got = e.run(off, iv, ln, 4, 0.4)
    t = e.timing(); c = e.debug_counters()
    print("configs[3] every read:", all(np.array_equal(a, b) for a, b in zip(got, want)))
    print({k: round(v, 4) for k, v in t.items() if k.endswith("_ms") and not k.startswith("class") and v}, "class_ms", [round(x, 4) for x in t["class_ms"]])
    print("   counters: n", c["n"][:12], "reads left to sweep_lds_read (fb_med)", c["fb_med"], "over_med", c["over_med"], "fb_big", c["fb_big"], "regions", c["total_regions"])
PY
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "phases", {k: round(v,4) for k,v in (h.get("phases_full_timing_ms") or {}).items()}, d["parity"][:9])'
for sh in 0 1 2 4 0; do
  echo -n "== cfg3 share $sh: "; YACRD_FUSED_SHARE=$sh timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3_share.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o s -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 20 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs3.csv \;
rm -rf $out/prof
head -8 $out/kernel_stats_configs3.csv | cut -c1-150
