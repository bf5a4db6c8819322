#!/bin/bash
# thirty-fourth GPU call of round 6, the library as built from a clean tree: smoke, a longer fuzz soak over six flag sets, every read of
# configs[1] / [2] from the jittered generator (sigma 100 / 300) at full size, configs[3] / [4] clamped once more
out=gpurun_out/r06H; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
{ timeout 300 python tools/gpu_fuzz.py 150; YACRD_FUZZ_MED=1 timeout 300 python tools/gpu_fuzz.py 120; YACRD_FUZZ_WIDE=1 timeout 300 python tools/gpu_fuzz.py 100; YACRD_FUZZ_ITEMS2=1 timeout 300 python tools/gpu_fuzz.py 100; YACRD_FUZZ_ONE_LAUNCH=1 timeout 300 python tools/gpu_fuzz.py 100; YACRD_SPLIT_MIN_READS=0 timeout 300 python tools/gpu_fuzz.py 100; } 2>&1 | grep gpu_fuzz | tee $out/fuzz_soak.log
F100=$(python -c "import sys; sys.path.insert(0,'.'); from yacrd_amd import host; print(host.SYNTH_F_JITTER | host.synth_f_sigma(100))")
F300=$(python -c "import sys; sys.path.insert(0,'.'); from yacrd_amd import host; print(host.SYNTH_F_JITTER | host.synth_f_sigma(300))")
YACRD_SYNTH_FLAGS=$F100 timeout 900 python tools/scale_check.py 2 3 > $out/scale_jitter100_configs_1_2.jsonl 2> $out/scale100.err
YACRD_SYNTH_FLAGS=$F300 timeout 900 python tools/scale_check.py 2 3 > $out/scale_jitter300_configs_1_2.jsonl 2> $out/scale300.err
timeout 900 python tools/scale_check.py 4 5 > $out/scale_configs_3_4.jsonl 2> $out/scale34.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06H/scale*.jsonl")):
    for l in open(f):
        d = json.loads(l); print(f.split("/")[-1], d["config"], d["synth_flags"], d["reads"], d["bit_exact_all_reads"], d["partition8_invariant"], d["order_invariant"])
PY
