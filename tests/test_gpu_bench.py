"""bench.py's contract on the GPU box: the one-line JSON with roofline / cpu_baseline and the extra
blocks (every config scaled down: plumbing, not a measurement), and the N>1 path (one process per rank,
read partition, barrier + max over ranks) with the ENGINE under the ranks — two ranks on the one device
through YACRD_BENCH_DEVICE / gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reject_constant(name):
    raise ValueError("non-strict JSON constant %s in the driver's line" % name)


def _last_json(text):
    """The driver's view: the LAST stdout line, compact (a 30 KB line came back `parsed: null`),
    strict JSON (no NaN / Infinity)."""
    last = text.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last.encode()) < 4096, len(last)
    return json.loads(last, parse_constant=_reject_constant)


def test_single_gpu_line_scaled(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                        "--scale", "0.02", "--small-steps", "20", "--sub-steps", "3", "--sigma100-steps", "2",
                        "--extras-file", str(tmp_path / "extras.json")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1  # nothing but the compact line on stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "parity", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["roofline"]["bound"] == "hbm" and "traffic" in line["roofline"] and "traffic_source" in line["roofline"]
    assert line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["sample"]
    assert "configs[4]" in line["config"]["workload"] and line["config"]["workload"].startswith("SCALED")
    for k in ("whole_path_frac_of_peak", "configs2_ms", "skewed_ms", "one_launch_us", "e2e_overlaps_per_sec",
              "configs1_pipelined_us", "configs1_one_at_a_time_us"):
        assert isinstance(line[k], float) and line[k] > 0, (k, line[k])
    v100 = line["value_sigma100"]
    assert v100["parity"].startswith("bit-exact") and v100["ms_per_step"] > 0 and v100["steps"] == 2
    assert set(line["jitter"]) == {"configs[1]", "configs[2]", "configs[1]_sigma100", "configs[2]_sigma100",
                                   "configs[1]_sigma300", "configs[2]_sigma300"}
    assert line["extras"] == "extras.json"
    d = json.loads((tmp_path / "extras.json").read_text(), parse_constant=_reject_constant)
    for k in ("metric", "value", "ms_per_step", "steps", "parity"):
        assert d[k] == line[k], k
    assert d["metric"] == "reads_per_sec_classified" and d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert "configs[4]" in d["config"]["workload"] and d["config"]["workload"].startswith("SCALED")
    assert d["headline"]["reads"] == 100000 and d["config"]["torch_distributed_backend"].startswith("none")
    assert d["steps"] == 4 and d["warmup"] == 2
    assert d["parity"].startswith("bit-exact")
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    assert r["timed_launches"] == r["launches"] == 4  # every launch of the timed region carries its events
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    for k in ("pcie_inclusive", "end_to_end", "small_batches", "configs2", "skewed"):
        assert "error" not in d[k], d[k]
    sb = d["small_batches"]
    assert sb["parity"].startswith("bit-exact") and sb["roofline"]["timed_launches"] == 20
    assert d["pcie_inclusive"]["h2d_GBps"] > 5 and d["pcie_inclusive"]["reads_per_sec"] < sb["reads_per_sec"]
    e = d["end_to_end"]
    assert e["overlaps"] == 100000 and e["host_parser"]["stream"]["reads_found"] == 2000 and e["overlaps_per_sec"] > 1e6
    assert e["device_parser"]["same_result_as_host_parser"] and e["device_parser"]["phases"]["reads_found"] == 2000
    sc = e["at_scale"]
    assert sc["same_reads_regions_types"] and sc["device_parser"]["overlaps_per_sec"] > 1e6
    for k in ("10%", "40%"):
        assert d["more_bad_reads"][k]["parity"].startswith("bit-exact")
    c2, sk = d["configs2"], d["skewed"]
    assert c2["parity"].startswith("bit-exact") and c2["reads"] == 40000 and "configs[2]" in c2["workload"]
    assert sk["parity"].startswith("bit-exact") and sk["reads"] == 200 and "SKEWED" in sk["workload"]
    assert sk["roofline"]["size_class"] in ("M1", "M2", "BIG")
    for k in ("configs[1]", "configs[2]", "configs[1]_sigma100", "configs[2]_sigma100", "configs[1]_sigma300", "configs[2]_sigma300"):
        j = d["jitter"][k]
        assert j["parity"].startswith("bit-exact") and "reflected" in j["workload"].lower()
        share = d["jitter"]["healthy_share_of_screened_reads"][k]  # (None: a scaled batch too short for the screening build)
        assert share is None or 0.0 <= share <= 1.0


@pytest.mark.parametrize("weak", [False, True])
def test_two_ranks_on_one_device(weak, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, YACRD_BENCH_DEVICE="0", YACRD_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--scale", "0.03"]
    if weak:
        cmd.append("--weak")
    cmd += ["--extras-file", str(tmp_path / "extras.json")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    d = json.loads((tmp_path / "extras.json").read_text())
    assert line["n_gpus"] == 2 and line["value"] == d["value"] and line["config"]["torch_distributed_backend"] == "gloo"
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["parity"].startswith("bit-exact")
    if weak:
        assert d["scaling"] == "weak" and "configs[1]" in d["config"]["workload"]
        return
    # the engine ran under both ranks, each on its own read range, and matched the oracle there
    h = d["headline"]
    assert d["scaling"] == "strong" and "configs[4]" in d["config"]["workload"] and len(h["per_rank"]) == 2
    assert d["config"]["torch_distributed_backend"] == "gloo"
    assert sum(r["reads"] for r in h["per_rank"]) == 150000
    assert sum(r["intervals"] for r in h["per_rank"]) == 30000000
    assert h["interval_imbalance_max_over_min"] < 1.05


def test_plain_python_with_gpus_2_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` — no torchrun around it — must run TWO ranks (it launches itself
    under torch.distributed.run), say so in the line, and carry one entry per rank; and a rank count that disagrees
    with the launcher's stops the run."""
    env = dict(os.environ, YACRD_BENCH_DEVICE="0", YACRD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--scale", "0.03", "--extras-file", str(tmp_path / "extras.json")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    d = json.loads((tmp_path / "extras.json").read_text())
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "strong"
    assert line["rccl_ranks"] == 0 and line["config"]["torch_distributed_backend"] == "gloo"  # (two ranks on ONE device: gloo)
    assert len(line["per_rank_ms"]) == 2 and all(x > 0 for x in line["per_rank_ms"])
    assert abs(max(line["per_rank_ms"]) - line["ms_per_step"]) / line["ms_per_step"] < 0.5
    assert line["parity"].startswith("bit-exact")
    pr = d["headline"]["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and sum(r["reads"] for r in pr) == 150000
    assert all(r["parity_sample_ok"] for r in pr)
    # under a launcher that runs two ranks, --gpus 4 is refused before anything is measured
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "4", "--scale", "0.03"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr and not bad.stdout.strip()
