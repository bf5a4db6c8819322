#!/bin/bash
# twenty-sixth GPU call of round 6: the default bench line of the final bench.py (small extra blocks: fastest of three regions), the bench tests
out=gpurun_out/r06z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3; tail -c 1300 $out/bench_default.json; cp bench_extras.json $out/bench_extras.json
python - <<'PY' | tee $out/regions.log
import json
d = json.load(open("bench_extras.json"))
print("small_batches", [round(x * 1e3, 1) for x in d["small_batches"]["timed_regions_ms_per_step"]])
for k, b in d["jitter"].items():
    if isinstance(b, dict) and "timed_regions_ms_per_step" in b:
        print(k, [round(x * 1e3, 1) for x in b["timed_regions_ms_per_step"]], b["paths"])
print("box", d["box"])
PY
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3 | tee $out/bench_tests.log
