"""bench.py's contract on the GPU box: the one-line JSON with roofline / cpu_baseline and the extra
blocks, and the N>1 path (one process per rank, read partition, barrier + max over ranks) with the
ENGINE under the ranks — two ranks on the one device through YACRD_BENCH_DEVICE / gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line_small():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "4",
                        "--reads", "20000", "--overlaps", "1000000", "--large-reads", "50000",
                        "--large-overlaps", "2500000", "--large-steps", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["metric"] == "reads_per_sec_classified" and d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["parity"].startswith("bit-exact")
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    for k in ("pcie_inclusive", "end_to_end", "large"):
        assert "error" not in d[k], d[k]
    assert d["pcie_inclusive"]["h2d_GBps"] > 10 and d["pcie_inclusive"]["reads_per_sec"] < d["value"]
    e = d["end_to_end"]
    assert e["overlaps"] == 1000000 and e["stream"]["reads_found"] == 20000 and e["overlaps_per_sec"] > 1e6
    assert d["large"]["parity"].startswith("bit-exact") and d["large"]["n_gpus"] == 1


@pytest.mark.parametrize("strong", [False, True])
def test_two_ranks_on_one_device(strong):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, YACRD_BENCH_DEVICE="0", YACRD_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "4", "--reads", "20000", "--overlaps", "1000000",
           "--large-reads", "60000", "--large-overlaps", "3000000", "--large-steps", "2"]
    if strong:
        cmd.append("--strong")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2
    lg = d["large"]
    assert "error" not in lg, lg
    # the engine ran under both ranks, each on its own read range, and matched the oracle there
    assert lg["parity"].startswith("bit-exact") and len(lg["per_rank"]) == 2
    assert sum(r["reads"] for r in lg["per_rank"]) == 60000
    assert sum(r["intervals"] for r in lg["per_rank"]) == 6000000
    assert lg["interval_imbalance_max_over_min"] < 1.05
    if strong:
        assert d["scaling"] == "strong" and d["value"] == lg["reads_per_sec"]
        assert "configs[2]" in d["config"]["workload"]
    else:
        assert d["scaling"] == "weak"
