#!/bin/bash
# forty-third GPU call of round 6, the library with the ramp's mirror in the second looks and the one-writer editors: suite, smoke, fuzz, bench line
out=gpurun_out/r06M; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -6 | tee $out/gpu_suite.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
{ timeout 200 python tools/gpu_fuzz.py 90; YACRD_FUZZ_MED=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_WIDE=1 timeout 200 python tools/gpu_fuzz.py 60; YACRD_FUZZ_ONE_LAUNCH=1 timeout 200 python tools/gpu_fuzz.py 45; } 2>&1 | grep gpu_fuzz | tee $out/fuzz_soak.log
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1000 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
YACRD_SYNTH_FLAGS=$(python -c "import sys; sys.path.insert(0,'.'); from yacrd_amd import host; print(host.SYNTH_F_JITTER | host.synth_f_sigma(300))") timeout 900 python tools/scale_check.py 2 3 > $out/scale_jitter300_configs_1_2.jsonl 2> $out/scale300.err
YACRD_SYNTH_FLAGS=$(python -c "import sys; sys.path.insert(0,'.'); from yacrd_amd import host; print(host.SYNTH_F_JITTER | host.synth_f_sigma(100))") timeout 900 python tools/scale_check.py 2 3 > $out/scale_jitter100_configs_1_2.jsonl 2> $out/scale100.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06M/scale*.jsonl")):
    for l in open(f):
        d = json.loads(l); print(f.split("/")[-1], d["config"], d["synth_flags"], d["reads"], d["bit_exact_all_reads"], d["partition8_invariant"], d["order_invariant"])
PY
