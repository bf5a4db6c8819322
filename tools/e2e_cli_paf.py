#!/usr/bin/env python3
"""tools/e2e_cli_paf.py [reads overlaps] — the drop-in CLI as a user runs it, cold process, PAF text in /dev/shm ->
.yacrd report: wall time of `yacrd -i s.paf -o r.yacrd -c 3 -n 0.4` — the DEFAULT flags: no -t (round 4: the device parser's copy
threads no longer follow -t) — and of the same with YACRD_NO_DEVICE_PARSER=1 -t 0 (the host parser on every CPU; GPU box).  Reports compared line by line."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yacrd_amd import host
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
O = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
exe = os.path.join(ROOT, "yacrd_amd", "bin", "yacrd")
d = "/dev/shm"
paf, gz = os.path.join(d, "yacrd_cli_%d.paf" % os.getpid()), None
try:
    t0 = time.perf_counter()
    host.synth_paf(host.SYNTH_SEQUEL, R, O, 20250307, paf)
    print("generated %.2f GB in %.1f s" % (os.path.getsize(paf) / 1e9, time.perf_counter() - t0), flush=True)
    os.system("cat %s > /dev/null" % paf)  # (a freshly written /dev/shm file is slow to read the first time: 1.2 s for 15 GB)
    time.sleep(3)  # (the generator's burst on all CPUs: let the cgroup quota recover)
    outs = []
    for label, env, extra in (("device parser (default flags)", {}, []), ("host parser (YACRD_NO_DEVICE_PARSER=1 -t 0)", {"YACRD_NO_DEVICE_PARSER": "1"}, ["-t", "0"])):
        out = os.path.join(d, "yacrd_cli_%d_%d.yacrd" % (os.getpid(), len(outs)))
        outs.append(out)
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-i", paf, "-o", out, "-c", "3", "-n", "0.4"] + extra, env=dict(os.environ, **env),
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr
            print("%s: %.3f s wall = %.1f M overlaps/s (report %d MB)" % (label, dt, O / dt / 1e6, os.path.getsize(out) >> 20), flush=True)
            time.sleep(2)
    a, b = open(outs[0]).read(), open(outs[1]).read()
    print("same report from both routes:", a == b, "lines", a.count("\n"))
    assert a == b
finally:
    for f in [paf] + [os.path.join(d, x) for x in os.listdir(d) if x.startswith("yacrd_cli_%d_" % os.getpid())]:
        if os.path.exists(f):
            os.remove(f)
