#!/bin/bash
# sixth GPU call of round 4: the default build — tests, editors by threads, the driver's line, the judged profiles, full-size parity
out=gpurun_out/r04f; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 900 python tools/edit_bench.py > $out/edit_bench.log 2>&1; cat $out/edit_bench.log
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; tail -3 $out/bench_default.time; tail -c 600 $out/bench_default.err
bash tools/profile_r04.sh $out/prof > $out/prof.log 2>&1; tail -12 $out/prof.log
timeout 1500 python tools/scale_check.py 2 3 4 5 > $out/scale_configs_1_2_3_4.jsonl 2> $out/scale.err; cat $out/scale_configs_1_2_3_4.jsonl | cut -c1-400
python bench.py --weak --no-extras > $out/bench_weak_configs1.json 2>/dev/null
python bench.py --config 3 --no-extras > $out/bench_configs3.json 2>/dev/null
python bench.py --config 2 --no-extras > $out/bench_configs2.json 2>/dev/null
ls -la $out $out/prof
