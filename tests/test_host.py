"""Host side (libyacrd_host.so): ingest -> CSR, report writer, synthetic generator, and that
both C-ABI libraries export every symbol their headers declare.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yacrd_[a-z0-9_]+)\s*\(", text)))


def test_engine_library_exports_every_declared_symbol():
    names = _declared("yacrd_engine.h")
    assert sorted(yacrd_amd.EXPORTED_SYMBOLS) == names
    lib = ctypes.CDLL(yacrd_amd.lib_path())
    for n in names:
        assert hasattr(lib, n), n
    assert yacrd_amd.load_library().yacrd_abi_version() == 7


def test_host_library_exports_every_declared_symbol():
    names = _declared("yacrd_host.h")
    assert sorted(host.EXPORTED_SYMBOLS) == names
    lib = host.load_library()
    for n in names:
        assert hasattr(lib, n), n


def test_engine_fails_loudly_without_gpu():
    import subprocess, sys
    code = ("import yacrd_amd\n"
            "try:\n    yacrd_amd.Engine()\n    print('CREATED')\n"
            "except yacrd_amd.EngineError as e:\n    print('ERR', e)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert "ERR" in out.stdout and "no HIP device" in out.stdout, out.stdout + out.stderr


def _same_csr(c, names, offsets, intervals, lengths):
    assert c.names == names
    assert np.array_equal(c.offsets, offsets)
    assert np.array_equal(c.lengths.astype(np.uint64), lengths)
    for r in range(len(names)):  # order inside a read is free when parsed multi-threaded
        a, b = int(offsets[r]), int(offsets[r + 1])
        assert sorted(map(tuple, c.intervals[a:b].tolist())) == sorted(map(tuple, intervals[a:b].tolist()))


@pytest.mark.parametrize("threads", [1, 3])
def test_ingest_fixture_matches_reference_semantics(golden_dir, threads):
    path = os.path.join(golden_dir, "reads.paf")
    with open(path) as f:
        want = oracle.to_csr(oracle.parse_paf(f))
    c = host.csr_from_file(path, n_threads=threads)
    assert (c.n_reads, c.n_intervals, c.n_records) == (230, 2572, 1286)
    _same_csr(c, *want)
    if threads == 1:
        assert np.array_equal(c.intervals, want[2])
    assert c.find(c.names[17]) == 17 and c.find("nope") == -1


def test_ingest_reference_unit_vectors():
    from test_oracle import M4_FILE, PAF_FILE
    for text, fmt in ((PAF_FILE, host.FMT_PAF), (M4_FILE, host.FMT_M4)):
        c = host.csr_from_memory(text, fmt, 1)
        assert c.names == ["1", "2", "3"]
        assert c.intervals.tolist() == [[20, 4500], [5500, 10000], [5500, 10000], [0, 4500]]
        assert c.lengths.tolist() == [12000, 10000, 10000]


def test_ingest_rules():
    # first length wins; extra columns ignored; empty lines and CRLF tolerated; no final newline
    text = ("a\t100\t1\t50\t+\tb\t200\t2\t60\textra\tcols\r\n\n"
            "b\t999\t3\t70\t-\ta\t888\t4\t80")
    c = host.csr_from_memory(text, host.FMT_PAF, 1)
    assert c.names == ["a", "b"] and c.lengths.tolist() == [100, 200]
    assert c.intervals.tolist() == [[1, 50], [4, 80], [2, 60], [3, 70]]
    for bad in ("a\t100\t1\t50\t+\tb\t200\t2\n",          # short record
                "a\t100\tx\t50\t+\tb\t200\t2\t60\n",      # non numeric
                "a\t100\t-1\t50\t+\tb\t200\t2\t60\n",     # negative
                "a\t100\t1\t50\t++\tb\t200\t2\t60\n",     # strand is a single char
                "a\t100\t1\t4294967296\t+\tb\t200\t2\t60\n"):  # u32 overflow
        with pytest.raises(host.HostError):
            host.csr_from_memory(bad, host.FMT_PAF, 1)
    assert host.csr_from_memory("", host.FMT_PAF, 1).n_reads == 0


def test_unused_typed_fields_agree_with_the_oracle():
    """The fields the reference deserialises and drops (PafRecord._strand: char; M4Record._error: f64,
    _shared_min: u64, _strand_a/_strand_b: char — src/io.rs:23-50) still fail the record when they do not parse
    (mod.rs:93-97 / :125-129): host parser and oracle accept and reject the same lines."""
    def both(text, fmt, parse):
        try:
            want = oracle.to_csr(parse(text))
        except (ValueError, IndexError):
            want = None
        try:
            got = host.csr_from_memory(text, fmt, 1)
        except host.HostError:
            got = None
        assert (want is None) == (got is None), text
        if want is not None:
            assert got.names == want[0] and got.intervals.tolist() == want[2].tolist()
        return want is not None

    paf = "a\t100\t1\t50\t%s\tb\t200\t2\t60\n"
    for strand, ok in (("+", True), ("-", True), ("*", True), ("\u00e9", True), ("", False), ("++", False), ("+-", False)):
        assert both(paf % strand, host.FMT_PAF, oracle.parse_paf) == ok, strand
    m4 = "a b %s %s %s 1 50 100 %s 2 60 200\n"
    for err, shared, sa, sb, ok in (("0.1", "42", "0", "1", True), ("1e-3", "0x2a", "+", "-", True),
                                    (".5", "+7", "0", "0", True), ("1.", "7", "0", "0", True),
                                    ("inf", "7", "0", "0", True), ("NaN", "7", "0", "0", True),
                                    ("-infinity", "7", "0", "0", True), ("0x1p3", "7", "0", "0", False),
                                    ("e5", "7", "0", "0", False), (".", "7", "0", "0", False),
                                    ("1.5", "-7", "0", "0", False), ("1.5", "7.0", "0", "0", False),
                                    ("1.5", "18446744073709551616", "0", "0", False),
                                    ("1.5", "7", "00", "0", False), ("1.5", "7", "0", "01", False),
                                    ("abc", "7", "0", "0", False)):
        assert both(m4 % (err, shared, sa, sb), host.FMT_M4, oracle.parse_m4) == ok, (err, shared, sa, sb)


def test_ingest_gzip_and_format_sniffing(golden_dir, tmp_path):
    import gzip, shutil
    src = os.path.join(golden_dir, "reads.paf")
    gz = str(tmp_path / "x.paf.gz")
    with open(src, "rb") as i, gzip.open(gz, "wb") as o:
        shutil.copyfileobj(i, o)
    assert host.csr_from_file(gz).n_intervals == 2572
    odd = str(tmp_path / "reads.txt")
    shutil.copy(src, odd)
    with pytest.raises(host.HostError):
        host.csr_from_file(odd)


def test_report_writer_reproduces_truth(golden_dir, tmp_path):
    """Report bytes (src/editor/mod.rs:61-107) from oracle results over the ingested CSR."""
    c = host.csr_from_file(os.path.join(golden_dir, "reads.paf"), n_threads=2)
    bo, br, rt = oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), 0, 0.8)
    out = str(tmp_path / "r.yacrd")
    c.write_report(out, bo, br, rt)
    with open(out) as f:
        got = [l.rstrip("\n") for l in f]
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        truth = set(l.rstrip("\n") for l in f)
    assert len(got) == 230 and set(got) == truth
    assert got == oracle.report_from_csr(c.names, c.lengths, bo, br, rt)


def test_synth_is_deterministic_and_paf_agrees(tmp_path):
    a = host.synth_csr(host.SYNTH_ONT, 300, 6000, 20241110)
    b = host.synth_csr(host.SYNTH_ONT, 300, 6000, 20241110)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    offsets, intervals, lengths = a
    assert int(offsets[-1]) == 12000 and lengths.min() >= 500 and lengths.max() <= 150000
    reg = intervals[:, 0] < intervals[:, 1]
    assert reg.mean() > 0.99
    assert (intervals[reg][:, 1] <= np.repeat(lengths, np.diff(offsets).astype(np.int64))[reg]).all()
    p = str(tmp_path / "s.paf")
    host.synth_paf(host.SYNTH_ONT, 300, 6000, 20241110, p)
    c = host.csr_from_file(p, n_threads=1)
    assert c.n_records == 6000
    # same per-read content (read order differs: first appearance vs numeric id)
    for i, name in enumerate(c.names):
        r = int(name[1:])
        a0, a1 = int(offsets[r]), int(offsets[r + 1])
        b0, b1 = int(c.offsets[i]), int(c.offsets[i + 1])
        assert c.lengths[i] == lengths[r]
        assert np.array_equal(c.intervals[b0:b1], intervals[a0:a1])


def test_synth_profiles():
    o, iv, ln = host.synth_csr(host.SYNTH_SKEWED, 100, 60000, 20241112)
    n = np.diff(o).astype(np.int64)
    assert n.min() >= 1000 and n.max() > 3 * n.min() and ln.min() >= 200000
    o, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, 500, 20000, 20241111)
    assert ln.min() >= 1000 and ln.max() <= 60000


def test_partition_reads_balances_intervals():
    o, _, _ = host.synth_csr(host.SYNTH_ONT, 5000, 100000, 7)
    for parts in (1, 2, 4, 8):
        cuts = yacrd_amd.partition_reads(o, parts)
        assert cuts[0] == 0 and cuts[-1] == 5000 and (np.diff(cuts.astype(np.int64)) >= 0).all()
        work = [int(o[int(cuts[i + 1])] - o[int(cuts[i])]) for i in range(parts)]
        assert max(work) - min(work) <= 0.05 * (200000 / parts) + 200


def test_ingest_is_identical_for_every_thread_count(tmp_path):
    """Shared id table + per-chunk fill cursors: names, lengths, offsets AND the order of the
    intervals inside every read equal the single-threaded (= line order, = oracle) result."""
    import oracle
    paf = str(tmp_path / "t.paf")
    host.synth_paf(host.SYNTH_ONT, 20000, 300000, 11, paf)
    with open(paf) as f:
        names, off, iv, ln = oracle.to_csr(oracle.parse_paf(f))
    iv = np.asarray(iv).reshape(-1, 2)
    for th in (1, 3, 8, 32):
        c = host.csr_from_file(paf, n_threads=th)
        assert c.names == list(names)
        assert np.array_equal(c.lengths, np.asarray(ln).astype(np.uint32))
        assert np.array_equal(c.offsets, off)
        assert np.array_equal(c.intervals.reshape(-1, 2), iv), th


# ---- compressed overlap files as text in memory (yacrd_text_from_file; round 4) -------------------------------
def _bgzf(data, block=40000):
    """BGZF (bgzip): gzip members with a "BC" extra subfield holding the member's size - 1, at most 64 KiB each,
    and the empty end-of-file member."""
    import struct
    import zlib
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def test_text_from_file_codecs(tmp_path):
    import bz2
    import gzip
    import lzma
    rng = np.random.default_rng(9)
    text = b"".join(b"r%07d\t%d\t%d\t%d\t+\tr%07d\t9000\t10\t900\t800\t800\t255\n" % (i, 5000 + i % 7, i % 100, 1000 + i % 100, (i * 7) % 5000)
                    for i in range(60000)) + bytes(rng.integers(65, 90, 1000, dtype=np.uint8))
    plain = tmp_path / "a.paf"
    plain.write_bytes(text)
    assert host.text_from_file(str(plain)) is None  # not compressed: read where it lies
    cases = {"one.paf.gz": gzip.compress(text, 1), "two.paf.gz": gzip.compress(text[:100000]) + gzip.compress(text[100000:]),
             "b.paf.bz2": bz2.compress(text), "x.paf.xz": lzma.compress(text), "bg.paf.gz": _bgzf(text)}
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        for th in (1, 4):
            with host.text_from_file(str(p), n_threads=th) as t:
                assert t.bytes() == text, name
                if name == "bg.paf.gz":
                    assert t.members == len(text) // 40000 + 2 and (th == 1 or t.threads >= 1)
                else:
                    assert t.members == 1
    # truncated / corrupt streams are errors, like the reference's readers
    for name in ("one.paf.gz", "bg.paf.gz", "b.paf.bz2", "x.paf.xz"):
        blob = cases[name]
        bad = tmp_path / ("bad_" + name)
        bad.write_bytes(blob[: len(blob) * 2 // 3])
        with pytest.raises(host.HostError):
            host.text_from_file(str(bad))
    flipped = bytearray(cases["bg.paf.gz"])
    flipped[len(flipped) // 2] ^= 0x55
    (tmp_path / "flip.paf.gz").write_bytes(bytes(flipped))
    with pytest.raises(host.HostError):
        host.text_from_file(str(tmp_path / "flip.paf.gz"))
    empty = tmp_path / "e.paf.gz"
    empty.write_bytes(gzip.compress(b""))
    with host.text_from_file(str(empty)) as t:
        assert t.n_bytes == 0
