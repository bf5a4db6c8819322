"""yacrd_engine_ingest_paf — PAF text to read types with the PARSE on the GPU — against the host parser, the
oracle's ingest (oracle.parse_paf, pinned on the reference's vectors) and the reference fixture; whatever only the
host parser handles must come back as NeedsHostParser, never as a wrong answer."""
import os

import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import host
from cases import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    with yacrd_amd.Engine() as e:
        yield e


def _check(engine, path, cov, nc):
    got, names, lengths, stats = engine.ingest_paf(path, cov, nc)
    with open(path, newline="") as f:
        reads = oracle.parse_paf(f.read())
    w_names, off, iv, ln = oracle.to_csr(reads)
    assert names == list(w_names), "first-appearance order of the reads"
    assert np.array_equal(lengths.astype(np.uint64), ln), "first length seen"
    assert stats["n_reads"] == len(names) and stats["n_records"] * 2 == int(off[-1])
    want = oracle.run(off, iv, ln, cov, nc, n_threads=4)
    assert_same(got, want, path)
    return got, names, lengths, stats


def test_reference_fixture(engine, golden_dir):
    got, names, lengths, _ = _check(engine, os.path.join(golden_dir, "reads.paf"), 0, 0.8)
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        truth = set(line.rstrip("\n") for line in f)
    lines = oracle.report_from_csr(names, lengths.astype(np.uint64), got.bad_offsets, got.bad_regions, got.read_type)
    assert set(lines) == truth


@pytest.mark.parametrize("prof,R,O,cov", [(host.SYNTH_ONT, 3000, 60000, 4), (host.SYNTH_SEQUEL, 800, 90000, 3),
                                         (host.SYNTH_ONT, 40000, 45000, 0)])
def test_synthetic_paf_matches_host_parser_and_oracle(engine, tmp_path, prof, R, O, cov):
    paf = str(tmp_path / "s.paf")
    host.synth_paf(prof, R, O, 11 + cov, paf)
    got, names, lengths, stats = _check(engine, paf, cov, 0.4)
    c = host.csr_from_file(paf, n_threads=3)
    assert c.names == names and np.array_equal(c.lengths, lengths)
    assert stats["text_bytes"] == os.path.getsize(paf)


def test_text_variants(engine, tmp_path):
    base = "a\t100\t1\t50\t+\tb\t200\t2\t60\t49\t58\t255\tcm:i:5\n" \
           "b\t999\t3\t70\t-\tc\t300\t4\t80\n" \
           "\n" \
           "c\t+300\t+5\t+90\t\u00e9\ta\t100\t0\t100\textra\n"
    for i, text in enumerate([base, base.replace("\n", "\r\n"), base.rstrip("\n"), "", "\n\n",
                              "x\t10\t0\t5\t+\tx\t10\t5\t10"]):
        p = str(tmp_path / ("v%d.paf" % i))
        with open(p, "w", newline="") as f:
            f.write(text)
        _check(engine, p, 0, 0.8)


@pytest.mark.parametrize("text", [
    '"a"\t100\t1\t50\t+\tb\t200\t2\t60\n',                       # a quoted field
    "a\t100\t1\t50\t+\tb\t200\t2\t60\t1\rb\t9\t0\t1\t+\ta\t100\t5\t6\n",  # a lone CR ends a record
    "a\t0x64\t1\t50\t+\tb\t200\t2\t60\n",                            # the csv crate reads 0x integers
    "a\t100\t1\t50\t+\tb\t200\t2\n",                                  # eight columns: the host words the error
    "a\t100\t1\t50\t++\tb\t200\t2\t60\n",                            # strand is not one character
    "a\t100\t1\t4294967296\t+\tb\t200\t2\t60\n",                     # u32 overflow
    "a\t4294967296\t1\t5\t+\tb\t200\t2\t60\n",                       # a length beyond the engine
])
def test_inputs_for_the_host_parser(engine, tmp_path, text):
    p = str(tmp_path / "h.paf")
    with open(p, "w", newline="") as f:
        f.write(text)
    with pytest.raises(yacrd_amd.NeedsHostParser):
        engine.ingest_paf(p, 0, 0.8)
    # and the engine is still usable
    off = np.array([0, 1], dtype=np.uint64)
    assert engine.run(off, np.array([[0, 5]], dtype=np.uint32), np.array([10], dtype=np.uint32), 0, 0.8).bad_regions.tolist() == [[5, 10]]


def test_long_lines_and_ids_across_tiles(engine, tmp_path):
    """Lines of a few bytes up to several KB (trailing tag columns, as minimap2 -c writes them), ids of up to 3000
    bytes, empty lines, CRLF — over a few hundred KB, so that line starts, ids and fields fall on and across the
    32 KiB tiles the parse kernel stages in LDS and the 1 KiB it stages beyond them."""
    rng = np.random.default_rng(5)
    ids = ["r%d" % i for i in range(300)] + ["x" * int(n) + str(i) for i, n in enumerate(rng.integers(40, 3000, size=12))]
    lens = {k: int(rng.integers(1000, 100000)) for k in ids}
    for crlf in (False, True):
        out = []
        for i in range(6000):
            a, b = ids[int(rng.integers(len(ids)))], ids[int(rng.integers(len(ids)))]
            sa, sb = int(rng.integers(0, lens[a] - 1)), int(rng.integers(0, lens[b] - 1))
            ea, eb = int(rng.integers(sa + 1, lens[a] + 1)), int(rng.integers(sb + 1, lens[b] + 1))
            line = "%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d" % (a, lens[a], sa, ea, "+-"[i & 1], b, lens[b], sb, eb)
            k = int(rng.integers(0, 6))
            if k >= 2:
                line += "\t%d\t%d\t255" % (ea - sa, eb - sb)
            if k >= 4:
                line += "\tcg:Z:" + "12M3I" * int(rng.integers(1, 1200))
            out.append(line)
            if rng.random() < 0.02:
                out.append("")
        eol = "\r\n" if crlf else "\n"
        p = str(tmp_path / ("long%d.paf" % crlf))
        with open(p, "w", newline="") as f:
            f.write(eol.join(out) + eol)
        assert os.path.getsize(p) > 300000
        _check(engine, p, 2, 0.4)


def test_record_room_estimate_falls_short_and_segments(engine, tmp_path):
    """The room for the records comes from the line density of the file's first MiB (+25 %): a file that opens
    with long lines and goes on with short ones outgrows it, and the parse is repeated with the exact number.
    The file is also longer than one 128 MiB scan + parse segment: lines straddle the segment boundary."""
    rng = np.random.default_rng(77)
    path = str(tmp_path / "skew.paf")
    ids = ["r%05d" % i for i in range(3000)]
    with open(path, "w") as f:
        # ~1.3 MiB of lines carrying a 2 KB tag behind the nine columns: ~600 lines per MiB
        pad = "x" * 2000
        for k in range(700):
            a, b = ids[k % 50], ids[(k * 7 + 1) % 50]
            f.write("%s\t9000\t%d\t%d\t+\t%s\t9000\t%d\t%d\ttg:Z:%s\n" % (a, k, k + 500, b, 2 * k, 2 * k + 700, pad))
        # then short lines: ~30 per KiB; enough of them to pass 128 MiB
        block = []
        for k in range(20000):
            a, b = int(rng.integers(0, 3000)), int(rng.integers(0, 3000))
            s1, s2 = int(rng.integers(0, 8000)), int(rng.integers(0, 8000))
            block.append("%s\t9000\t%d\t%d\t-\t%s\t9000\t%d\t%d\n" % (ids[a], s1, s1 + int(rng.integers(1, 900)), ids[b], s2,
                                                                  s2 + int(rng.integers(1, 900))))
        text = "".join(block)
        reps = (130 << 20) // len(text) + 1
        for _ in range(reps):
            f.write(text)
    assert os.path.getsize(path) > (129 << 20)
    got, names, lengths, stats = engine.ingest_paf(path, 3, 0.4)
    c = host.csr_from_file(path, n_threads=8)
    assert names == c.names and np.array_equal(lengths, c.lengths)
    assert stats["n_records"] == 700 + 20000 * reps
    want = oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), 3, 0.4, n_threads=8)
    assert_same(got, want, "skewed line lengths")
    # the parser keeps its buffers between calls; trim gives them back, and the next call allocates again
    engine.trim()
    again, names2, _, _ = engine.ingest_paf(path, 3, 0.4)
    assert names2 == names
    assert_same(again, want, "after yacrd_engine_trim")
    engine.trim()


def _random_paf(rng, n_lines, anomaly):
    """Random PAF text: plain records in the shapes the fast path takes (optional '+' signs, trailing columns, CRLF,
    empty lines, ids of 1..40 bytes incl. non-ASCII, a multi-byte strand); `anomaly` adds ONE thing that is either the
    host parser's business or an error in the reference."""
    ids = ["r%d" % i for i in range(int(rng.integers(1, 12)))] + ["ü%d" % i for i in range(2)] + ["x" * int(rng.integers(1, 40))]
    lens = {k: int(rng.integers(1, 5000)) for k in ids}
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    lines = []
    for _ in range(n_lines):
        a, b = ids[int(rng.integers(len(ids)))], ids[int(rng.integers(len(ids)))]
        f = [a, str(lens[a]), str(int(rng.integers(0, 6000))), str(int(rng.integers(0, 6000))), str(rng.choice(["+", "-", "*", "é"])),
             b, str(lens[b] if rng.random() < 0.8 else int(rng.integers(1, 9999))), str(int(rng.integers(0, 6000))), str(int(rng.integers(0, 6000)))]
        for k in (1, 2, 3, 6, 7, 8):
            if rng.random() < 0.1:
                f[k] = "+" + f[k]
        if rng.random() < 0.4:
            f += ["%d" % int(rng.integers(0, 300)), "tp:A:P", "cm:i:%d" % int(rng.integers(0, 99))][: int(rng.integers(1, 4))]
        lines.append("\t".join(f))
        if rng.random() < 0.05:
            lines.append("")
    if anomaly and lines:
        k = int(rng.integers(len(lines)))
        parts = lines[k].split("\t") if lines[k] else ["a", "1", "0", "1", "+", "b", "1", "0", "1"]
        kind = anomaly
        if kind == 1:
            parts[0] = '"' + parts[0] + '"'               # quoted id: same record for the csv crate
        elif kind == 2:
            parts[2] = "0x1f"                               # hex integer
        elif kind == 3:
            parts = parts[:8]                               # too few columns: an error
        elif kind == 4:
            parts[4] = "+-"                                 # strand of two characters: an error
        elif kind == 5:
            parts[3] = "4294967296"                         # u32 overflow: an error
        elif kind == 6:
            parts[8] = parts[8] + "\rtail"                  # a lone CR splits the record
        elif kind == 7:
            parts[7] = "-3"                                 # negative: an error
        elif kind == 8:
            parts[1] = ""                                   # empty length: an error
        lines[k] = "\t".join(parts)
    text = eol.join(lines)
    if lines and rng.random() < 0.7:
        text += eol
    return text


_SEEDS = [20250305, 7, 99991] + [int(x) for x in os.environ.get("YACRD_PAF_FUZZ_SEEDS", "").split(",") if x]  # (soak runs: more seeds)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_text_against_the_oracle_ingest(engine, tmp_path, seed):
    """Differential fuzz of the device parser: on every random text it either hands the file to the host parser
    (NeedsHostParser) or returns exactly what the oracle's ingest + sweep return; plain texts must NOT fall back."""
    rng = np.random.default_rng(seed)
    taken = fell_back = 0
    for case in range(400):
        anomaly = 0 if case % 2 == 0 else int(rng.integers(1, 9))
        text = _random_paf(rng, int(rng.integers(0, 120)), anomaly)
        p = str(tmp_path / "f.paf")
        with open(p, "w", newline="", encoding="utf-8") as f:
            f.write(text)
        try:
            reads = oracle.parse_paf(text)
            want = oracle.to_csr(reads)
        except (ValueError, IndexError):
            want = None
        cov = int(rng.integers(0, 4))
        try:
            got, names, lengths, stats = engine.ingest_paf(p, cov, 0.4)
        except yacrd_amd.NeedsHostParser:
            fell_back += 1
            assert anomaly != 0, "a plain text went to the host parser:\n%r" % text
            continue
        taken += 1
        assert want is not None, "the device parser accepted what the reference rejects:\n%r" % text
        w_names, off, iv, ln = want
        assert names == list(w_names) and np.array_equal(lengths.astype(np.uint64), ln), text
        assert_same(got, oracle.run(off, iv, ln, cov, 0.4, n_threads=2), "case %d" % case)
    assert taken >= 200 and fell_back >= 100, (taken, fell_back)


# ---- M4 / MHAP (Reads2Ovl::init_m4, src/reads2ovl/mod.rs:115-145; M4Record, src/io.rs:36-50) on the device --------
def _paf_to_m4(text, err="0.1", shared="2"):
    out = []
    for l in text.split("\n"):
        if not l:
            out.append(l)
            continue
        f = l.rstrip("\r").split("\t")
        out.append(" ".join([f[0], f[5], err, shared, "0", f[2], f[3], f[1], "1", f[7], f[8], f[6]]) + ("\r" if l.endswith("\r") else ""))
    return "\n".join(out)


def _check_m4(engine, path, cov, nc):
    got, names, lengths, stats = engine.ingest_paf(path, cov, nc, fmt=2)
    with open(path, newline="") as f:
        reads = oracle.parse_m4(f.read())
    w_names, off, iv, ln = oracle.to_csr(reads)
    assert names == list(w_names) and np.array_equal(lengths.astype(np.uint64), ln)
    assert stats["n_reads"] == len(names) and stats["n_records"] * 2 == int(off[-1])
    assert_same(got, oracle.run(off, iv, ln, cov, nc, n_threads=4), path)
    return got, names, lengths


def test_m4_fixture_and_synthetic(engine, golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "reads.paf")) as f:
        m4 = _paf_to_m4(f.read())
    p = str(tmp_path / "reads.m4")
    with open(p, "w", newline="") as f:
        f.write(m4)
    got, names, lengths = _check_m4(engine, p, 0, 0.8)
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        truth = set(line.rstrip("\n") for line in f)
    assert set(oracle.report_from_csr(names, lengths.astype(np.uint64), got.bad_offsets, got.bad_regions, got.read_type)) == truth
    # by file name (format 0), and the same reads as the host parser finds
    got0, names0, _, _ = engine.ingest_paf(p, 0, 0.8, fmt=0)
    assert names0 == names and np.array_equal(got0.read_type, got.read_type)
    paf = str(tmp_path / "s.paf")
    host.synth_paf(host.SYNTH_ONT, 3000, 60000, 12, paf)
    with open(paf) as f:
        text = _paf_to_m4(f.read(), err="1.5e-2", shared="+17")
    p2 = str(tmp_path / "s.mhap")
    with open(p2, "w", newline="") as f:
        f.write(text)
    _, names2, lengths2 = _check_m4(engine, p2, 4, 0.4)
    c = host.csr_from_file(p2, n_threads=3)
    assert c.names == names2 and np.array_equal(c.lengths, lengths2)


@pytest.mark.parametrize("err,taken", [("0.1", True), ("12", True), ("-0.5", True), ("+3.25E+2", True), ("1e5", True),
                                       ("inf", False), ("NaN", False), (".5", False), ("1.", False), ("0x1p3", False), ("", False)])
def test_m4_error_rate_shapes(engine, tmp_path, err, taken):
    """The device takes the plain decimal forms of M4's f64 column; whatever else Rust's f64::from_str may accept (or
    reject) is the host parser's: never a wrong answer."""
    text = "a b %s 42 0 1 50 100 1 2 60 200\nb c %s 7 + 3 70 200 - 4 80 300 extra cols\n" % (err, err)
    p = str(tmp_path / "e.m4")
    with open(p, "w", newline="") as f:
        f.write(text)
    if taken:
        _check_m4(engine, p, 0, 0.8)
    else:
        with pytest.raises(yacrd_amd.NeedsHostParser):
            engine.ingest_paf(p, 0, 0.8, fmt=2)


@pytest.mark.parametrize("text", [
    "a b 0.1 2 0 1 50 100 1 2 60\n",                # eleven columns
    "a b 0.1 2 00 1 50 100 1 2 60 200\n",           # strand of two characters
    "a b 0.1 2 0 1 50 100 1 2 60 4294967296\n",     # a length beyond the engine
    "a b 0.1 -2 0 1 50 100 1 2 60 200\n",           # _shared_min is a u64
    "a  b 0.1 2 0 1 50 100 1 2 60 200\n",           # two spaces: an empty field
    '"a" b 0.1 2 0 1 50 100 1 2 60 200\n',          # a quoted field
])
def test_m4_inputs_for_the_host_parser(engine, tmp_path, text):
    p = str(tmp_path / "h.m4")
    with open(p, "w", newline="") as f:
        f.write(text)
    with pytest.raises(yacrd_amd.NeedsHostParser):
        engine.ingest_paf(p, 0, 0.8, fmt=2)


def test_m4_random_text_against_the_oracle_ingest(engine, tmp_path):
    rng = np.random.default_rng(4242)
    taken = fell_back = 0
    for case in range(300):
        anomaly = 0 if case % 2 == 0 else int(rng.integers(1, 9))
        text = _paf_to_m4(_random_paf(rng, int(rng.integers(0, 100)), 0).replace("\r\n", "\n"),
                          err=str(rng.choice(["0.25", "3", "1e-3", "+7.5"])), shared=str(int(rng.integers(0, 1000))))
        lines = text.split("\n")
        if anomaly and any(lines):
            k = int(rng.choice([i for i, l in enumerate(lines) if l]))
            f = lines[k].split(" ")
            if anomaly == 1:
                f[0] = '"' + f[0] + '"'
            elif anomaly == 2:
                f[5] = "0x1f"
            elif anomaly == 3:
                f = f[:11]
            elif anomaly == 4:
                f[8] = "+-"
            elif anomaly == 5:
                f[6] = "4294967296"
            elif anomaly == 6:
                f[2] = "inf"
            elif anomaly == 7:
                f[3] = "-3"
            else:
                f[9] = ""
            lines[k] = " ".join(f)
            text = "\n".join(lines)
        p = str(tmp_path / "f.m4")
        with open(p, "w", newline="", encoding="utf-8") as fh:
            fh.write(text)
        try:
            want = oracle.to_csr(oracle.parse_m4(text))
        except (ValueError, IndexError):
            want = None
        cov = int(rng.integers(0, 4))
        try:
            got, names, lengths, stats = engine.ingest_paf(p, cov, 0.4, fmt=2)
        except yacrd_amd.NeedsHostParser:
            fell_back += 1
            assert anomaly != 0, "a plain text went to the host parser:\n%r" % text
            continue
        taken += 1
        assert want is not None, "the device parser accepted what the reference rejects:\n%r" % text
        w_names, off, iv, ln = want
        assert names == list(w_names) and np.array_equal(lengths.astype(np.uint64), ln), text
        assert_same(got, oracle.run(off, iv, ln, cov, 0.4, n_threads=2), "case %d" % case)
    assert taken >= 150 and fell_back >= 60, (taken, fell_back)


def test_compressed_files_take_the_device_parser(engine, tmp_path):
    """Round 4: a gzip / bzip2 / xz / BGZF overlap file is inflated by libyacrd_host
    (yacrd_text_from_file; the reference sniffs every input like this, src/util.rs:57-70) and parsed ON THE DEVICE from
    memory (yacrd_engine_ingest_overlaps_mem): reads, lengths, regions and types of the plain file; the file variant
    refuses a compressed file (YACRD_EFALLBACK) instead of scanning its bytes as text."""
    import bz2
    import gzip
    import lzma
    from test_host import _bgzf
    paf = str(tmp_path / "s.paf")
    host.synth_paf(host.SYNTH_ONT, 5000, 150000, 91, paf)
    want, names, lengths, _ = _check(engine, paf, 4, 0.4)
    text = open(paf, "rb").read()
    for name, blob in (("a.paf.gz", gzip.compress(text, 1)), ("b.paf.bz2", bz2.compress(text)), ("c.paf.xz", lzma.compress(text, preset=1)),
                       ("d.paf.gz", _bgzf(text, 60000))):
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        with pytest.raises(yacrd_amd.NeedsHostParser):
            engine.ingest_paf(p, 4, 0.4)
        with host.text_from_file(p) as t:
            assert t.n_bytes == len(text)
            got, n2, l2, st = engine.ingest_text((t.address, t.n_bytes), 4, 0.4)
        assert n2 == names and np.array_equal(l2, lengths) and st["text_bytes"] == len(text)
        assert_same(got, (want.bad_offsets, want.bad_regions, want.read_type), name)
    # bytes in, M4 from memory, the empty text
    got, n2, l2, _ = engine.ingest_text(text, 4, 0.4)
    assert n2 == names
    got, n2, l2, _ = engine.ingest_text(b"", 0, 0.8)
    assert n2 == [] and len(got.read_type) == 0
    got, n2, l2, _ = engine.ingest_text(b"1 2 0.1 2 0 100 450 1000 0 550 900 1000\n", 0, 0.8, fmt=2)
    assert n2 == ["1", "2"] and l2.tolist() == [1000, 1000]
