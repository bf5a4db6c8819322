#!/usr/bin/env python3
"""tools/e2e_cli.py [reads] [overlaps] — the drop-in CLI end to end on synthetic files (GPU box):
PAF text -> ingest -> engine -> report -> scrubb of a FASTQ, wall time per stage."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yacrd_amd import host  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
O = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
exe = os.path.join(ROOT, "yacrd_amd", "bin", "yacrd")
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    paf, fq = os.path.join(td, "s.paf"), os.path.join(td, "s.fastq")
    t0 = time.perf_counter()
    host.synth_paf(host.SYNTH_ONT, R, O, 20241113, paf)
    host.synth_fastq(host.SYNTH_ONT, R, O, 20241113, R // 200, fq)
    gen = time.perf_counter() - t0
    out = {"reads": R, "overlaps": O, "paf_MB": os.path.getsize(paf) >> 20, "fastq_MB": os.path.getsize(fq) >> 20,
           "generate_s": round(gen, 2)}
    env = dict(os.environ, YACRD_INGEST_TIMING="1")
    for label, extra in (("report_only", []), ("report_and_scrubb", ["scrubb", "-i", fq, "-o", os.path.join(td, "o.fastq")])):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", paf, "-o", os.path.join(td, label + ".yacrd"), "-c", "4", "-n", "0.4", "-t", "0"] + extra,
                           env=env, capture_output=True, text=True)
        out[label + "_s"] = round(time.perf_counter() - t0, 3)
        assert p.returncode == 0, p.stderr
        if label == "report_only":
            out["ingest_phases_ms"] = {l.split()[1] + ("_" + l.split()[2] if not l.split()[2][0].isdigit() else ""): float(l.split()[-2])
                                       for l in p.stderr.splitlines() if l.startswith("[ingest]")}
    out["report_lines"] = sum(1 for _ in open(os.path.join(td, "report_only.yacrd")))
    out["scrubbed_MB"] = os.path.getsize(os.path.join(td, "o.fastq")) >> 20
    print(out)
