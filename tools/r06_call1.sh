#!/bin/bash
# first GPU call of round 6: the new tests (bench --gpus 2 launching itself, steady paths at sigma 300, the multi-device branch
# forced on one device), configs[3] taken apart (counters, the three-launch chain beside the fused launch, timing-only builds
# without the fallback / without the count), non-temporal interval loads A/B on configs[4] / [2]
out=gpurun_out/r06a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_bench.py::test_plain_python_with_gpus_2_launches_two_ranks tests/test_gpu_paths.py \
  tests/test_gpu_ingest_group.py::test_the_multi_device_branch_forced_on_one_device -x -q 2>&1 | tail -15 | tee $out/new_tests.log
python - <<'PY' 2>&1 | tee $out/cfg3_counters.log
import numpy as np, time
import yacrd_amd
from yacrd_amd import host
off, iv, ln = host.synth_csr(host.SYNTH_SKEWED, 10000, 30000000, 20241108 + 4)
n = np.diff(off.astype(np.int64))
print("reads", len(ln), "intervals", int(off[-1]), "n<=4096", int((n <= 4096).sum()), "n<=8192", int((n <= 8192).sum()), "n<=16384", int((n <= 16384).sum()), "max", int(n.max()))
for flags, name in ((0, "default"), (yacrd_amd.F_NO_FUSED_SCREEN, "three-launch chain")):
    with yacrd_amd.Engine(flags=flags | yacrd_amd.F_TIMING_FULL) as e:
        for _ in range(3):
            e.run(off, iv, ln, 4, 0.4)
        t = e.timing(); c = e.debug_counters()
        print(name, {k: round(v, 4) for k, v in t.items() if k.endswith("_ms") and not k.startswith("class") and v}, "class_ms", [round(x, 4) for x in t["class_ms"]])
        print("   counters: n", c["n"][:12], "fb_med", c["fb_med"], "over_med", c["over_med"], "fb_big", c["fb_big"], "rej", c["rej_small"], c["rej_med"], c["rej_big"], "regions", c["total_regions"])
PY
Q='import sys,json; d=json.loads(sys.stdin.readline()); h=d["headline"]; r=d["roofline"]; print(d["config"]["workload"][:12], "ms", round(d["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "follow_on", round(r.get("finish_compact_kernel_ms") or 0,4), "phases", {k: round(v,4) for k,v in (h.get("phases_full_timing_ms") or {}).items()}, d["parity"][:9])'
cp yacrd_amd/lib/libyacrd_hip.so /tmp/keep.so
for v in keep nofb nofb_nocount keep nofb nofb_nocount; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== cfg3 $v: "; timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done 2>&1 | tee $out/cfg3_experiments.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
echo -n "== cfg3 chain (NO_FUSED_SCREEN): " | tee -a $out/cfg3_experiments.log
timeout 600 python bench.py --config 3 --no-extras --no-cpu-baseline --print-extras --flags 1048576 2>/dev/null | head -1 | python -c "$Q" | tee -a $out/cfg3_experiments.log
rocprofv3 --kernel-trace --stats -d $out/prof_cfg3_chain -o chain -- python bench.py --config 3 --no-extras --no-cpu-baseline --steps 10 --flags 1048576 > $out/prof_cfg3_chain.log 2>&1
for c in 4 2; do for v in keep nt keep nt; do cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so; [ $v = keep ] || cp variants/lib_$v.so yacrd_amd/lib/libyacrd_hip.so
  echo -n "== cfg$c $v: "; timeout 900 python bench.py --config $c --no-extras --no-cpu-baseline --print-extras 2>/dev/null | head -1 | python -c "$Q"
done; done 2>&1 | tee $out/ab_nt.log
cp /tmp/keep.so yacrd_amd/lib/libyacrd_hip.so
find $out/prof_cfg3_chain -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/cfg3_chain_kernel_stats.csv
rm -rf $out/prof_cfg3_chain
