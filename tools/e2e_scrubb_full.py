#!/usr/bin/env python3
"""tools/e2e_scrubb_full.py [reads overlaps] — BASELINE configs[4] as the config says it: a 5 M-read synthetic FASTQ +
a 500 M-overlap PAF through the drop-in CLI with its DEFAULT flags (no -t),
    yacrd -i s.paf -o r.yacrd -c 3 -n 0.4 scrubb -i s.fastq -o o.fastq
on one MI355X, wall clock per stage (YACRD_CLI_TIMING), and the scrubbed output checked against the oracle's editor
(oracle/editors.py over the oracle's bad regions) on the file's first records, byte for byte.  Sizes are cut down when
/dev/shm cannot hold the three files."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
from oracle import editors as oed  # noqa: E402
from yacrd_amd import host  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
O = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000_000
d = "/dev/shm"
st = os.statvfs(d)
free = st.f_bavail * st.f_frsize
need = O * 75 + 2 * R * 21000  # PAF + FASTQ in + FASTQ out
while need > 0.8 * free and R > 1000:
    R //= 2
    O //= 2
    need = O * 75 + 2 * R * 21000
print("free on %s: %.0f GB; running %d reads / %d overlaps" % (d, free / 1e9, R, O), flush=True)
seed = 20241108 + 5
paf, fq, rep, out = (os.path.join(d, "yacrd_full_%d.%s" % (os.getpid(), x)) for x in ("paf", "fastq", "yacrd", "out.fastq"))
exe = os.path.join(ROOT, "yacrd_amd", "bin", "yacrd")
try:
    t0 = time.perf_counter()
    host.synth_paf(host.SYNTH_SEQUEL, R, O, seed, paf)
    t1 = time.perf_counter()
    host.synth_fastq(host.SYNTH_SEQUEL, R, O, seed, R // 200, fq)
    t2 = time.perf_counter()
    print("generated PAF %.1f GB in %.0f s, FASTQ %.1f GB in %.0f s" % (os.path.getsize(paf) / 1e9, t1 - t0, os.path.getsize(fq) / 1e9, t2 - t1), flush=True)
    time.sleep(5)  # (the generators' burst on all CPUs: let the cgroup quota recover)
    for rep_no in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", paf, "-o", rep, "-c", "3", "-n", "0.4", "scrubb", "-i", fq, "-o", out],
                           env=dict(os.environ, YACRD_CLI_TIMING="1"), capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr
        stages = {l.split()[1]: float(l.split()[2]) for l in p.stderr.splitlines() if l.startswith("[timing]")}
        print("run %d: %.2f s wall; stages %s; scrubb %.1f GB/s of FASTQ in; report %d MB, scrubbed %.1f GB" % (
            rep_no, dt, stages, os.path.getsize(fq) / max(stages.get("edit", dt), 1e-9) / 1e9, os.path.getsize(rep) >> 20,
            os.path.getsize(out) / 1e9), flush=True)
        time.sleep(3)
    # ---- the check: the oracle's regions for the first K reads, its editor over the FASTQ's first records
    K = min(R, 20000)
    off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, R, O, seed)
    bo, br, rt = oracle.run(off[: K + 1], iv[: int(off[K])], ln[:K].astype(np.uint64), 3, 0.4, n_threads=8)
    table = {"r%09d" % r: ([tuple(int(x) for x in br[k]) for k in range(int(bo[r]), int(bo[r + 1]))], int(ln[r])) for r in range(K)}
    del off, iv
    with open(fq, "rb") as f:
        data = f.read(int(ln[: K // 2].astype(np.int64).sum()) * 2)  # about half of those reads' records
    # whole records of reads below K only (extras x... are unknown reads: copied through)
    cut, pos, n_rec = 0, 0, 0
    while True:
        e = pos
        for _ in range(4):
            e = data.find(b"\n", e) + 1
            if e == 0:
                break
        if e == 0:
            break
        name = data[pos + 1: data.find(b" ", pos)]
        if name.startswith(b"r") and int(name[1:]) >= K:
            break
        pos, cut, n_rec = e, e, n_rec + 1
    want = oed.edit_fastq("scrubb", data[:cut], table, 0.4)
    with open(out, "rb") as f:
        got = f.read(len(want))
    print("oracle check: first %d records (%.1f MB of FASTQ) -> %d bytes scrubbed: %s" % (
        n_rec, cut / 1e6, len(want), "byte-identical" if got == want else "MISMATCH"), flush=True)
    # the report's lines of those K reads against the oracle (the report is in first-appearance order of the PAF)
    names = ["r%09d" % r for r in range(K)]
    lines = set(oracle.report_from_csr(names, ln[:K], bo, br, rt))
    mine = set()
    with open(rep) as f:
        for l in f:
            nm = l.split("\t", 2)[1]
            if nm.startswith("r") and int(nm[1:]) < K:
                mine.add(l.rstrip("\n"))
    print("report check: the %d lines of reads r0 .. r%d %s" % (K, K - 1, "identical" if mine == lines else "MISMATCH"), flush=True)
    assert got == want and mine == lines
finally:
    for x in (paf, fq, rep, out):
        if os.path.exists(x):
            os.remove(x)
