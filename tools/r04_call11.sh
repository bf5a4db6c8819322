#!/bin/bash
# eleventh GPU call of round 4: the final build — tests, the driver's line, the judged profiles
out=gpurun_out/r04k; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; tail -3 $out/bench_default.time; tail -c 400 $out/bench_default.err
bash tools/profile_r04.sh $out/prof > $out/prof.log 2>&1; tail -4 $out/prof.log
python bench.py --weak --no-extras > $out/bench_weak_configs1.json 2>/dev/null
python bench.py --config 3 --no-extras > $out/bench_configs3.json 2>/dev/null
python bench.py --config 2 --no-extras > $out/bench_configs2.json 2>/dev/null
timeout 900 python tools/scale_check.py 2 3 4 5 > $out/scale_configs_1_2_3_4.jsonl 2> $out/scale.err; cut -c1-160 $out/scale_configs_1_2_3_4.jsonl
