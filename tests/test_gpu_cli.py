"""The `yacrd` drop-in CLI end to end on the GPU box: the reference's own integration tests
(tests/run.rs:95-300) replayed against yacrd_amd/bin/yacrd."""
import gzip
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "yacrd_amd", "bin", "yacrd")


def run(*args):
    p = subprocess.run([BIN] + list(args), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p


@pytest.fixture(scope="module")
def work(golden_dir, tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    shutil.copy(os.path.join(golden_dir, "reads.paf"), d / "reads.paf")
    with gzip.open(os.path.join(golden_dir, "reads.fastq.gz"), "rb") as i, open(d / "reads.fastq", "wb") as o:
        shutil.copyfileobj(i, o)
    return d


def lines(path):
    with open(path) as f:
        return [l.rstrip("\n") for l in f]


def test_detection(work, golden_dir):  # run.rs:95-117, compared as a set (diff_unorder)
    run("-i", str(work / "reads.paf"), "-o", str(work / "result.yacrd"))
    assert set(lines(work / "result.yacrd")) == set(lines(os.path.join(golden_dir, "truth.yacrd")))
    assert len(lines(work / "result.yacrd")) == 230


def test_detection_ondisk_flag_and_gpus(work, golden_dir):  # run.rs:120-160
    run("-i", str(work / "reads.paf"), "-o", str(work / "result.ondisk.yacrd"), "-d",
        str(work / "ondisk"), "--gpus", "1", "-t", "2")
    assert set(lines(work / "result.ondisk.yacrd")) == set(lines(os.path.join(golden_dir, "truth.yacrd")))


@pytest.mark.parametrize("op", ["filter", "extract", "split", "scrubb"])
def test_editors(work, golden_dir, op):  # run.rs:162-300, ordered comparison (diff)
    out = work / ("result.%s.fastq" % op)
    run("-i", str(work / "reads.paf"), "-o", str(work / ("result.%s.yacrd" % op)), op, "-i",
        str(work / "reads.fastq"), "-o", str(out))
    with gzip.open(os.path.join(golden_dir, "truth.%s.fastq.gz" % op), "rt") as f:
        truth = [l.rstrip("\n") for l in f]
    assert lines(out) == truth


def test_report_as_input(work, golden_dir):  # src/main.rs:43-45: FromReport bypass + editor
    rep = os.path.join(golden_dir, "truth.yacrd")
    out = work / "from_report.scrubb.fastq"
    run("-i", rep, "-o", str(work / "again.yacrd"), "scrubb", "-i", str(work / "reads.fastq"), "-o", str(out))
    assert set(lines(work / "again.yacrd")) == set(lines(rep))
    with gzip.open(os.path.join(golden_dir, "truth.scrubb.fastq.gz"), "rt") as f:
        assert lines(out) == [l.rstrip("\n") for l in f]


def test_thresholds_match_secondary_vectors(work):
    import hashlib
    run("-i", str(work / "reads.paf"), "-o", str(work / "c3.yacrd"), "-c", "3", "-n", "0.4")
    body = b"".join(sorted(l.encode() + b"\n" for l in lines(work / "c3.yacrd")))
    assert hashlib.sha256(body).hexdigest() == \
        "0b207320f179e75a93862fded411651225dfc94e982b1afa2f3e8ea3c113dce2"


def test_m4_mhap_and_compressed_inputs(work, golden_dir, tmp_path):
    """N3: .m4 / .mhap overlaps (src/reads2ovl/mod.rs:115-145) and bzip2 / xz / gzip inputs sniffed
    from magic bytes (src/util.rs:57-70), for the overlaps and for the sequences; editor output in
    the input's compression (src/util.rs:72-87)."""
    import bz2
    import lzma
    truth = set(lines(os.path.join(golden_dir, "truth.yacrd")))
    # PAF -> M4 columns (src/io.rs:36-50): a b err shared strand_a ba ea la strand_b bb eb lb
    m4 = []
    for l in lines(work / "reads.paf"):
        f = l.split("\t")
        m4.append(" ".join([f[0], f[5], "0.1", "2", "0", f[2], f[3], f[1], "0", f[7], f[8], f[6]]))
    for name in ("reads.m4", "reads.mhap"):
        with open(tmp_path / name, "w") as o:
            o.write("\n".join(m4) + "\n")
        run("-i", str(tmp_path / name), "-o", str(tmp_path / (name + ".yacrd")), "-t", "2")
        assert set(lines(tmp_path / (name + ".yacrd"))) == truth
    paf = open(work / "reads.paf", "rb").read()
    fq = open(work / "reads.fastq", "rb").read()
    with gzip.open(os.path.join(golden_dir, "truth.scrubb.fastq.gz"), "rb") as f:
        want = f.read()
    for ext, mod in (("bz2", bz2), ("xz", lzma), ("gz", gzip)):
        with mod.open(tmp_path / ("x.paf." + ext), "wb") as o:
            o.write(paf)
        with mod.open(tmp_path / ("r.fastq." + ext), "wb") as o:
            o.write(fq)
        out = tmp_path / ("scrubbed.fastq." + ext)
        run("-i", str(tmp_path / ("x.paf." + ext)), "-o", str(tmp_path / (ext + ".yacrd")), "-t", "0",
            "scrubb", "-i", str(tmp_path / ("r.fastq." + ext)), "-o", str(out))
        assert set(lines(tmp_path / (ext + ".yacrd"))) == truth, ext
        assert mod.open(out, "rb").read() == want, ext
    # (round 4: a compressed overlap file is inflated on the host and parsed on the device)
    p = subprocess.run([BIN, "-i", str(tmp_path / "x.paf.gz"), "-o", str(tmp_path / "t.yacrd")], capture_output=True, text=True,
                       env=dict(os.environ, YACRD_CLI_TIMING="1"))
    assert p.returncode == 0 and "[timing] inflate" in p.stderr and set(lines(tmp_path / "t.yacrd")) == truth
    blob = open(tmp_path / "x.paf.xz", "rb").read()
    open(tmp_path / "cut.paf.xz", "wb").write(blob[:len(blob) // 2])
    p = subprocess.run([BIN, "-i", str(tmp_path / "cut.paf.xz"), "-o", str(tmp_path / "cut.yacrd")],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "xz" in p.stderr


def test_several_engines_stream_group_path(work, golden_dir, tmp_path):
    """--gpus N streams the records to N engines by handle mod N (yacrd_stream_group); a 1-GPU box has no
    second device, so the CLI refuses --gpus 64 loudly, and YACRD_GPUS_ON_DEVICE=0 puts all N engines on
    device 0: same report as --gpus 1, also for gzip input (host parser) and a bigger synthetic file."""
    p = subprocess.run([BIN, "-i", str(work / "reads.paf"), "-o", str(work / "g2.yacrd"), "--gpus", "64"],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "device" in p.stderr
    # (round 5: --gpus N parses on all N engines, yacrd_engines_ingest_overlaps; YACRD_NO_DEVICE_PARSER=1 sends the input down
    # the path of the inputs the device parser does not take — the host parser and the stream group)
    env = dict(os.environ, YACRD_GPUS_ON_DEVICE="0", YACRD_NO_DEVICE_PARSER="1")
    from yacrd_amd import host
    big = str(tmp_path / "big.paf")
    host.synth_paf(host.SYNTH_ONT, 3000, 60000, 20250303, big)
    bigger = str(tmp_path / "bigger.paf")  # more than three 4 MiB chunks: every engine of --gpus 2 / 3 gets a range of its own
    host.synth_paf(host.SYNTH_ONT, 9000, 200000, 20250304, bigger)
    assert os.path.getsize(bigger) > (13 << 20)
    for src, cov in ((str(work / "reads.paf"), "0"), (big, "4"), (bigger, "4")):
        one = str(tmp_path / "one.yacrd")
        p = subprocess.run([BIN, "-i", src, "-o", one, "-c", cov, "-t", "4"], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        for n in ("2", "3"):
            out = str(tmp_path / ("n%s.yacrd" % n))
            p = subprocess.run([BIN, "-i", src, "-o", out, "-c", cov, "-t", "4", "--gpus", n], env=env,
                               capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            assert open(out).read() == open(one).read()
            # the default route for N > 1: the device parser on every engine
            env2 = dict(os.environ, YACRD_GPUS_ON_DEVICE="0", YACRD_CLI_TIMING="1")
            p = subprocess.run([BIN, "-i", src, "-o", out, "-c", cov, "--gpus", n], env=env2, capture_output=True, text=True)
            assert p.returncode == 0 and "device parser: %s engine(s)" % n in p.stderr, p.stderr
            assert open(out).read() == open(one).read()
    # ... and a compressed input through it (inflated on the host, the text handed to all engines)
    import gzip
    with open(bigger, "rb") as f, gzip.open(bigger + ".gz", "wb", compresslevel=1) as g:
        g.write(f.read())
    p = subprocess.run([BIN, "-i", bigger + ".gz", "-o", out, "-c", "4", "--gpus", "3"], env=env2, capture_output=True, text=True)
    assert p.returncode == 0 and "device parser: 3 engine(s)" in p.stderr, p.stderr
    assert open(out).read() == open(one).read()


def test_bad_usage_is_loud(work):
    p = subprocess.run([BIN, "-i", str(work / "reads.paf")], capture_output=True, text=True)
    assert p.returncode != 0
    p = subprocess.run([BIN, "-i", str(work / "nope.paf"), "-o", str(work / "x")], capture_output=True, text=True)
    assert p.returncode != 0 and "nope.paf" in p.stderr


def test_synthetic_pipeline_config5_shape(tmp_path):
    """configs[4] at reduced scale through the whole drop-in: PAF text -> ingest -> engine ->
    report -> scrubb of a FASTQ, against the CPU restatements (oracle ingest + sweep + editor)."""
    import numpy as np
    import oracle
    from oracle import editors
    from yacrd_amd import host
    R, O, seed = 4000, 200000, 20241113
    paf, fq = str(tmp_path / "s.paf"), str(tmp_path / "s.fastq")
    host.synth_paf(host.SYNTH_SEQUEL, R, O, seed, paf)
    host.synth_fastq(host.SYNTH_SEQUEL, R, O, seed, 20, fq)   # 20 reads that no overlap mentions
    for op in ("scrubb", "split"):
        rep, out = str(tmp_path / ("%s.yacrd" % op)), str(tmp_path / ("%s.fastq" % op))
        run("-i", paf, "-o", rep, "-c", "3", "-n", "0.4", "-t", "4", op, "-i", fq, "-o", out)
        with open(paf) as f:
            reads = oracle.parse_paf(f)
        table = {k: (oracle.compute_bad_part(v[0], v[1], 3), v[1]) for k, v in reads.items()}
        want_report = set(oracle.report_line(k, ln, reg, oracle.type_of_read(ln, reg, 0.4))
                          for k, (reg, ln) in table.items())
        assert set(lines(rep)) == want_report and len(lines(rep)) == len(table)
        with open(fq, "rb") as f:
            want = editors.edit_fastq(op, f.read(), table, 0.4)
        with open(out, "rb") as f:
            assert f.read() == want
    types = [l.split("\t")[0] for l in lines(rep)]
    assert types.count("Chimeric") > 0 and types.count("NotCovered") > 0


def test_scrubb_whole_output_at_a_hundredth_of_configs4(tmp_path):
    """configs[4] ("bit-exact vs CPU scrubb output") at 1/100 of its size — 50 000 reads / 5 M overlaps, ~1 GB of FASTQ —
    through the CLI with default flags, checked the way tools/e2e_scrubb_full.py checks the full size (round 5): every
    line of the report, the scrubbed file's record and byte totals against what the oracle's regions imply, byte-exact
    windows spread over the whole file (reference: src/editor/scrubbing.rs:156-236, tests/run.rs:254-300)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_scrubb_full as chk
    from yacrd_amd import host
    d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else str(tmp_path)
    R, O, seed = 50_000, 5_000_000, 20241108 + 5
    tag = os.path.join(d, "yacrd_test_%d_" % os.getpid())
    paf, fq, rep, out = tag + "s.paf", tag + "s.fastq", tag + "r.yacrd", tag + "o.fastq"
    try:
        host.synth_paf(host.SYNTH_SEQUEL, R, O, seed, paf)
        host.synth_fastq(host.SYNTH_SEQUEL, R, O, seed, R // 200, fq)
        run("-i", paf, "-o", rep, "-c", "3", "-n", "0.4", "scrubb", "-i", fq, "-o", out)
        off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, R, O, seed)
        res = chk.verify_scrubb(fq, out, rep, off, iv, ln, 3, 0.4, R // 200, n_windows=200, window_bytes=300_000, log=lambda s: None)
        assert res["ok"], res
        assert res["windows"]["records"] > 2000 and res["totals"]["types"][1] > 500 and res["totals"]["types"][2] > 500
    finally:
        for x in (paf, fq, rep, out):
            if os.path.exists(x):
                os.remove(x)


# filter / extract on OVERLAP files through the CLI: the reference's unit vectors (src/editor/filter.rs:289-359,
# extract.rs:293-362: read 1 bad — regions (10, 490), (510, 1000) of 1000 — here from a .yacrd report), N4
_PAF_UNIT = ("1\t12000\t20\t4500\t-\t2\t10000\t5500\t10000\t4500\t4500\t255\n"
             "1\t12000\t5500\t10000\t-\t3\t10000\t0\t4500\t4500\t4500\t255\n")
_M4_UNIT = "1 2 0.1 2 0 100 450 1000 0 550 900 1000\n1 3 0.1 2 0 550 900 1000 0 100 450 1000\n"
_REPORT_UNIT = "NotCovered\t1\t1000\t480,10,490;490,510,1000\nNotBad\t2\t1000\t\nNotBad\t3\t1000\t\n"


@pytest.mark.parametrize("op", ["filter", "extract"])
@pytest.mark.parametrize("ext,text", [(".paf", _PAF_UNIT), (".m4", _M4_UNIT), (".mhap", _M4_UNIT)])
def test_overlap_file_editors_through_the_cli(tmp_path, op, ext, text):
    rep = tmp_path / "in.yacrd"
    rep.write_text(_REPORT_UNIT)
    src, out = tmp_path / ("ovl" + ext), tmp_path / ("out" + ext)
    src.write_text(text)
    run("-i", str(rep), "-o", str(tmp_path / "again.yacrd"), op, "-i", str(src), "-o", str(out))
    # every line names read 1: filter drops them all, extract keeps them all
    assert out.read_text() == ("" if op == "filter" else text)
    # ... and a line between two good reads survives filter / is dropped by extract
    more = text + (("2\t1000\t0\t500\t+\t3\t1000\t500\t1000\t500\t500\t255\n") if ext == ".paf" else "2 3 0.1 2 0 0 500 1000 0 500 1000 1000\n")
    src.write_text(more)
    run("-i", str(rep), "-o", str(tmp_path / "again.yacrd"), op, "-i", str(src), "-o", str(out))
    assert out.read_text() == (more[len(text):] if op == "filter" else text)


def test_overlap_file_editors_from_a_detection_run(work, tmp_path):
    """The same with the bad parts coming from the GPU (detection on reads.paf, -c 0 -n 0.8: 4 chimeric reads), the
    overlap file being the PAF itself: filter keeps exactly the lines whose two reads are NotBad, extract the others
    (filter.rs:140-183, extract.rs:144-187), against the oracle's classification."""
    import oracle
    paf = work / "reads.paf"
    with open(paf) as f:
        reads = oracle.parse_paf(f)
    bad = {k for k, (iv, ln) in reads.items() if oracle.type_of_read(ln, oracle.compute_bad_part(iv, ln, 0), 0.8) != oracle.NOT_BAD}
    assert len(bad) == 4
    src = lines(paf)
    for op in ("filter", "extract"):
        out = tmp_path / (op + ".paf")
        run("-i", str(paf), "-o", str(tmp_path / (op + ".yacrd")), op, "-i", str(paf), "-o", str(out))
        hit = [l for l in src if (l.split("\t")[0] in bad or l.split("\t")[5] in bad)]
        want = [l for l in src if l not in hit] if op == "filter" else hit
        assert lines(out) == want, op
    assert 0 < len(hit) < len(src)
