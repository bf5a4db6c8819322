"""bench.py's LAST stdout line is what the driver parses: compact (< 4 KB), strict JSON, the contract's keys with
`roofline` and `cpu_baseline` in it — everything else goes to bench_extras.json (a 30 KB line came
back `parsed: null`).  CPU check on the full object of a recorded run (profiles/r04_bench_default.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bad_const(name):
    raise ValueError(name)


def test_compact_line_from_a_recorded_full_object():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    full["headline"]["whole_path_frac_of_peak"] = float("nan")  # a NaN anywhere must not reach the line
    full["jitter"]["configs[1]"]["ms_per_step"] = float("inf")
    line = bench.compact_line(bench.no_nan(full), "/somewhere/bench_extras.json")
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(s.encode()) < bench.COMPACT_LIMIT
    back = json.loads(s, parse_constant=_bad_const)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "parity", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert "configs[4]" in back["config"]["workload"] and len(back["config"]["workload"]) <= 300
    r = back["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] == full["roofline"]["traffic"]
    assert back["cpu_baseline"]["cores"] == 16 and back["cpu_baseline"]["kind"] == "port"
    assert back["whole_path_frac_of_peak"] is None and back["jitter"]["configs[1]"][0] is None
    assert back["configs2_ms"] > 0 and back["skewed_ms"] > 0 and back["one_launch_us"] > 0
    assert back["extras"] == "bench_extras.json"


def test_compact_line_survives_failed_blocks():
    import bench
    full = {"metric": "reads_per_sec_classified", "value": 1.0, "unit": "reads/s", "n_gpus": 1, "steps": 1, "warmup": 1,
            "ms_per_step": 1.0, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic", "config": {"workload": "w" * 5000}, "parity": "bit-exact", "roofline": {"bound": "hbm"},
            "cpu_baseline": {"error": "x" * 100}, "configs2": {"error": "boom"}, "jitter": {"configs[1]": {"error": "e"}}}
    line = bench.compact_line(bench.no_nan(full), "not written: OSError()")
    s = json.dumps(line, allow_nan=False)
    assert len(s) < bench.COMPACT_LIMIT and line["configs2_ms"] is None and line["extras"].startswith("not written")
