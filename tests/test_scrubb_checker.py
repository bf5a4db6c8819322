"""tools/e2e_scrubb_full.py's checker (round 5: the whole scrubbed output — report, totals, byte-exact windows — not its
first 0.2 %) on CPU: an output and a report made by the oracle's editor must pass, a flipped base in the middle of the
file, a dropped record, a wrong region in the report must not."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
from oracle import editors as oed  # noqa: E402
from yacrd_amd import host  # noqa: E402


def _case(tmp_path, R=600, O=40000, extras=5):
    import e2e_scrubb_full as chk
    seed = 99
    fq = str(tmp_path / "s.fastq")
    host.synth_fastq(host.SYNTH_SEQUEL, R, O, seed, extras, fq)
    off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, R, O, seed)
    bo, br, rt = oracle.run(off, iv, ln.astype(np.uint64), 3, 0.4, n_threads=2)
    names = ["r%09d" % r for r in range(R)]
    table = {names[r]: ([tuple(int(x) for x in br[k]) for k in range(int(bo[r]), int(bo[r + 1]))], int(ln[r])) for r in range(R)}
    out, rep = str(tmp_path / "o.fastq"), str(tmp_path / "r.yacrd")
    with open(fq, "rb") as f:
        data = oed.edit_fastq("scrubb", f.read(), table, 0.4)
    open(out, "wb").write(data)
    order = np.random.default_rng(1).permutation(R)  # (the report's order is unspecified)
    lines = oracle.report_from_csr(names, ln, bo, br, rt)
    open(rep, "w").write("".join(lines[i] + "\n" for i in order))
    return chk, fq, out, rep, off, iv, ln, extras, data


def test_checker_accepts_the_oracles_output_and_rejects_damage(tmp_path):
    chk, fq, out, rep, off, iv, ln, extras, data = _case(tmp_path)
    quiet = lambda s: None  # noqa: E731
    res = chk.verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, extras, n_windows=40, window_bytes=200_000, log=quiet)
    assert res["ok"], res
    assert res["totals"]["records"] == res["totals"]["records_expected"] and res["totals"]["bytes"] == res["totals"]["bytes_expected"]
    assert res["windows"]["chimeric"] > 0 and res["windows"]["not_covered"] > 0 and res["windows"]["unmentioned"] > 0
    # a flipped base in the middle of the file: same totals, a window must notice (160 windows of 200 KB overlap on this small file)
    mid = data.find(b"\n", len(data) // 2) + 1
    mid = data.find(b"\n", mid) + 5  # inside a sequence line
    dmg = bytearray(data)
    dmg[mid] = ord("N")
    open(out, "wb").write(bytes(dmg))
    res = chk.verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, extras, n_windows=160, window_bytes=200_000, log=quiet)  # (overlapping windows)
    assert not res["ok"] and res["windows"]["mismatches"] > 0
    # a record dropped from the end: the totals notice
    cut = data.rfind(b"\n@", 0, len(data) - 2) + 1
    open(out, "wb").write(data[:cut])
    res = chk.verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, extras, n_windows=8, window_bytes=100_000, log=quiet)
    assert not res["ok"] and res["totals"]["records"] == res["totals"]["records_expected"] - 1
    # a wrong region in the report
    open(out, "wb").write(data)
    txt = open(rep).read().split("\n")
    k = next(i for i, l in enumerate(txt) if l.startswith("Chimeric"))
    f = txt[k].split("\t")
    a, b, c = f[3].split(";")[0].split(",")
    f[3] = ";".join(["%s,%s,%d" % (a, b, int(c) + 1)] + f[3].split(";")[1:])
    txt[k] = "\t".join(f)
    open(rep, "w").write("\n".join(txt))
    res = chk.verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, extras, n_windows=8, window_bytes=100_000, log=quiet)
    assert not res["ok"] and not res["report"]["regions"]


def test_expected_totals_digit_edges():
    import e2e_scrubb_full as chk
    assert chk.digits(np.array([0, 9, 10, 99, 100, 999999999, 1000000000])).tolist() == [1, 1, 2, 2, 3, 9, 10]
