#!/bin/bash
# twenty-eighth GPU call of round 4: the final build at full size — every read of BASELINE configs[1..4] against the oracle
# (clamped generator; dovetail ends spread by sigma = 100 / 300 on configs[1..2]), and smoke()
out=gpurun_out/r04zb; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1200 python tools/scale_check.py 2 3 4 5 > $out/scale_configs_1_2_3_4.jsonl 2> $out/scale.err; cut -c1-220 $out/scale_configs_1_2_3_4.jsonl
YACRD_SYNTH_FLAGS=25602 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter100_configs_1_2.jsonl 2> $out/scale100.err; cut -c1-220 $out/scale_jitter100_configs_1_2.jsonl
YACRD_SYNTH_FLAGS=19206 timeout 600 python tools/scale_check.py 2 3 > $out/scale_jitter300_configs_1_2.jsonl 2> $out/scale300.err; cut -c1-220 $out/scale_jitter300_configs_1_2.jsonl
