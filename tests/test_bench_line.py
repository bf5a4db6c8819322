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


def test_gpus_flag_launches_the_ranks_itself(monkeypatch):
    """`--gpus N` was parsed and ignored.  Without a launcher and N > 1, bench.py becomes
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`; N = 1 stays one process."""
    import argparse
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake_execv(exe, cmd):
        seen["exe"], seen["cmd"] = exe, list(cmd)
        raise SystemExit(0)
    monkeypatch.setattr(bench.os, "execv", fake_execv)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    try:
        bench.launch_ranks(argparse.Namespace(gpus=4))
    except SystemExit:
        pass
    cmd = seen["cmd"]
    assert seen["exe"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7"]
    seen.clear()
    for g in (None, 1):  # one rank: no launcher
        a = argparse.Namespace(gpus=g)
        bench.launch_ranks(a)
        assert a.gpus == 1 and not seen


def test_gpus_flag_must_agree_with_the_launcher(monkeypatch):
    import argparse
    import subprocess
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    a = argparse.Namespace(gpus=None)
    bench.launch_ranks(a)
    assert a.gpus == 2  # (no flag: the launcher's count)
    a = argparse.Namespace(gpus=2)
    bench.launch_ranks(a)
    assert a.gpus == 2
    # a disagreement stops the run before anything is loaded or measured
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and not p.stdout.strip()


def test_compact_line_limit_is_enforced_whatever_the_blocks_hold():
    """the line could still pass 4096 bytes once both optional blocks were dropped."""
    import bench
    full = {"metric": "reads_per_sec_classified", "value": 1.0, "unit": "reads/s", "n_gpus": 8, "steps": 1, "warmup": 1,
            "ms_per_step": 1.0, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic", "config": {"workload": "w" * 5000, "parallelism": "y" * 5000}, "parity": "p" * 5000,
            "roofline": {"bound": "hbm", "traffic_source_short": "z" * 3000}, "cpu_baseline": {"error": "x" * 5000},
            "headline": {"per_rank": [{"ms_per_step": 1.0}] * 8}}
    line = bench.compact_line(bench.no_nan(full), "e.json")
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(s.encode()) < bench.COMPACT_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "parity", "roofline", "cpu_baseline"):
        assert k in line, k
