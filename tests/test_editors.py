"""Host editors (scrubb / split / filter / extract) and the .yacrd re-reader against the
reference's golden files and unit byte-strings.  CPU only: the BadPart table fed to the editors is
computed by the oracle here; on the GPU box tests/test_gpu_cli.py drives the same code from the
engine's results."""
import gzip
import os
import shutil

import numpy as np
import pytest

import oracle
from yacrd_amd import host

OPS = {"scrubb": host.OP_SCRUBB, "filter": host.OP_FILTER, "extract": host.OP_EXTRACT,
       "split": host.OP_SPLIT}


def table_from_reads(reads, cov, nc):
    names, offsets, intervals, lengths = oracle.to_csr(reads)
    bo, br, rt = oracle.run(offsets, intervals, lengths, cov, nc)
    return names, lengths.astype(np.uint32), bo, br, rt


@pytest.fixture(scope="module")
def fixture_table(golden_dir):
    with open(os.path.join(golden_dir, "reads.paf")) as f:
        return table_from_reads(oracle.parse_paf(f), 0, 0.8)


@pytest.fixture(scope="module")
def reads_fastq(golden_dir, tmp_path_factory):
    p = str(tmp_path_factory.mktemp("fq") / "reads.fastq")
    with gzip.open(os.path.join(golden_dir, "reads.fastq.gz"), "rb") as i, open(p, "wb") as o:
        shutil.copyfileobj(i, o)
    return p


# ---- tests/run.rs:162-300: filter / extract / split / scrubb, ordered line-exact comparison
@pytest.mark.parametrize("op", ["scrubb", "filter", "extract", "split"])
def test_golden_fastq(golden_dir, fixture_table, reads_fastq, tmp_path, op):
    out = str(tmp_path / ("result.%s.fastq" % op))
    host.edit_file(OPS[op], reads_fastq, out, *fixture_table)
    with gzip.open(os.path.join(golden_dir, "truth.%s.fastq.gz" % op), "rb") as f:
        truth = f.read()
    assert open(out, "rb").read() == truth


def test_gzip_in_gzip_out(golden_dir, fixture_table, tmp_path):
    src = str(tmp_path / "reads.fastq.gz")
    shutil.copy(os.path.join(golden_dir, "reads.fastq.gz"), src)
    out = str(tmp_path / "scrubbed.fastq.gz")
    host.edit_file(host.OP_SCRUBB, src, out, *fixture_table)
    assert open(out, "rb").read(2) == b"\x1f\x8b"
    with gzip.open(out, "rb") as f, gzip.open(os.path.join(golden_dir, "truth.scrubb.fastq.gz")) as t:
        assert f.read() == t.read()


# ---- unit byte-strings of the reference's editor tests ----------------------------------------
FASTA_22 = b">1\nACTGGGGGGACTGGGGGGACTG\n>2\nACTG\n>3\nACTG\n"
FASTQ_22 = b"@1\nACTGGGGGGACTGGGGGGACTG\n+\n??????????????????????\n@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"
FASTA_4 = b">1\nACTG\n>2\nACTG\n>3\nACTG\n"
FASTQ_4 = b"@1\nACTG\n+\n????\n@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"
PAF = (b"1\t12000\t20\t4500\t-\t2\t10000\t5500\t10000\t4500\t4500\t255\n"
       b"1\t12000\t5500\t10000\t-\t3\t10000\t0\t4500\t4500\t4500\t255\n")
M4 = b"1 2 0.1 2 0 100 450 1000 0 550 900 1000\n1 3 0.1 2 0 550 900 1000 0 100 450 1000\n"
CHIM = {"1": [[(10, 490), (510, 1000)], 1000]}  # read 1 Chimeric at -c 0 -n 0.8

UNIT = [
    # scrubbing.rs:247-395
    ("scrubb", ".fasta", FASTA_22, {"1": [[(0, 4), (9, 13), (18, 22)], 22]},
     b">1_0_4\nACTG\n>1_9_13\nACTG\n>1_18_22\nACTG\n>2\nACTG\n>3\nACTG\n"),
    ("scrubb", ".fasta", FASTA_22, {"1": [[(4, 18)], 22]},
     b">1_4_18\nGGGGGACTGGGGGG\n>2\nACTG\n>3\nACTG\n"),
    ("scrubb", ".fastq", FASTQ_22, {"1": [[(0, 4), (9, 13), (18, 22)], 22]},
     b"@1_0_4\nACTG\n+\n????\n@1_9_13\nACTG\n+\n????\n@1_18_22\nACTG\n+\n????\n@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"),
    ("scrubb", ".fastq", FASTQ_22, {"1": [[(4, 18)], 22]},
     b"@1_4_18\nGGGGGACTGGGGGG\n+\n??????????????\n@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"),
    # split.rs:237-321
    ("split", ".fasta", FASTA_22, {"1": [[(9, 13), (18, 22)], 22]},
     b">1_0_13\nACTGGGGGGACTG\n>1_18_22\nACTG\n>2\nACTG\n>3\nACTG\n"),
    ("split", ".fastq", FASTQ_22, {"1": [[(9, 13), (18, 22)], 22]},
     b"@1_0_13\nACTGGGGGGACTG\n+\n?????????????\n@1_18_22\nACTG\n+\n????\n@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"),
    # filter.rs:239-359
    ("filter", ".fasta", FASTA_4, CHIM, b">2\nACTG\n>3\nACTG\n"),
    ("filter", ".fastq", FASTQ_4, CHIM, b"@2\nACTG\n+\n????\n@3\nACTG\n+\n????\n"),
    ("filter", ".paf", PAF, CHIM, b""),
    ("filter", ".m4", M4, CHIM, b""),
    # extract.rs:243-362
    ("extract", ".fasta", FASTA_4, CHIM, b">1\nACTG\n"),
    ("extract", ".fastq", FASTQ_4, CHIM, b"@1\nACTG\n+\n????\n"),
    ("extract", ".paf", PAF, CHIM, PAF),
    ("extract", ".m4", M4, CHIM, M4),
]


@pytest.mark.parametrize("op,ext,data,reads,expect", UNIT)
def test_reference_editor_unit_vectors(tmp_path, op, ext, data, reads, expect):
    src = str(tmp_path / ("in" + ext))
    out = str(tmp_path / ("out" + ext))
    open(src, "wb").write(data)
    host.edit_file(OPS[op], src, out, *table_from_reads(reads, 0, 0.8))
    assert open(out, "rb").read() == expect


def test_editor_rules(tmp_path):
    # NotCovered reads are dropped by scrubb and split; description is kept on every piece;
    # unknown reads pass through untouched; out-of-range cut positions stop the read.
    reads = {"nc": [[(0, 10)], 100], "ch": [[(0, 10), (12, 20)], 20], "short": [[(5, 50)], 60]}
    table = table_from_reads(reads, 0, 0.8)
    fq = (b"@nc some desc\n" + b"A" * 100 + b"\n+\n" + b"?" * 100 + b"\n"
          b"@ch d1 d2\nACGTACGTACGTACGTACGT\n+\nABCDEFGHIJKLMNOPQRST\n"
          b"@short\nACGTACGTAC\n+\n??????????\n"
          b"@other x\nAC\n+\n??\n")
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(fq)
    out = str(tmp_path / "out.fq")
    host.edit_file(host.OP_SCRUBB, src, out, *table)
    got = open(out, "rb").read()
    assert got == (b"@ch_0_10 d1 d2\nACGTACGTAC\n+\nABCDEFGHIJ\n@ch_12_20 d1 d2\nACGTACGT\n+\nMNOPQRST\n"
                   b"@other x\nAC\n+\n??\n")  # short: first piece 5..50 is out of range -> nothing
    host.edit_file(host.OP_SPLIT, src, out, *table)
    assert open(out, "rb").read().startswith(b"@ch_0_10 d1 d2\n")
    with pytest.raises(host.HostError):
        host.edit_file(host.OP_SCRUBB, str(tmp_path / "x.paf"), out, *table)
    with pytest.raises(host.HostError):
        host.edit_file(host.OP_SCRUBB, str(tmp_path / "noext"), out, *table)


# ---- FromReport: src/stack.rs:270-309, :392-430
def test_report_reader(tmp_path, golden_dir):
    p = str(tmp_path / "r.yacrd")
    open(p, "w").write("NotBad\tSRR8494940.65223\t2706\t1131,0,1131;16,2690,2706\n"
                       "NotCovered\tSRR8494940.141626\t30116\t326,0,326;27159,2957,30116\n"
                       "Chimeric\tSRR8494940.91655\t15691\t151,0,151;4056,7213,11269;58,15633,15691\n"
                       "NotBad\tperfect\t2706\t\n")
    names, lengths, bo, br = host.report_read(p)
    assert names == ["SRR8494940.65223", "SRR8494940.141626", "SRR8494940.91655", "perfect"]
    assert lengths.tolist() == [2706, 30116, 15691, 2706]
    assert br.tolist() == [[0, 1131], [2690, 2706], [0, 326], [2957, 30116], [0, 151],
                           [7213, 11269], [15633, 15691]]
    assert bo.tolist() == [0, 2, 4, 7, 7]
    open(p, "w").write("Chimeric\tx\t15691\t151,0,151;58,156\n")  # corrupt (stack.rs:392-409)
    with pytest.raises(host.HostError):
        host.report_read(p)
    # round trip of the golden report
    names, lengths, bo, br = host.report_read(os.path.join(golden_dir, "truth.yacrd"))
    assert len(names) == 230 and int(bo[-1]) == 462


# ---- the chunk-parallel editors (round 4): byte-identical to the one-thread loop -----------------------------
def _synthetic_case(tmp_path, n_reads=300, n_ovl=9000):
    paf = str(tmp_path / "s.paf")
    fq = str(tmp_path / "s.fastq")
    host.synth_paf(host.SYNTH_ONT, n_reads, n_ovl, 4242, paf)
    host.synth_fastq(host.SYNTH_ONT, n_reads, n_ovl, 4242, 7, fq)
    with open(paf) as f:
        table = table_from_reads(oracle.parse_paf(f), 4, 0.4)
    return fq, table


@pytest.mark.parametrize("op", ["scrubb", "filter", "extract", "split"])
def test_parallel_editors_equal_one_thread(tmp_path, monkeypatch, op):
    fq, table = _synthetic_case(tmp_path)
    one = str(tmp_path / "one.fastq")
    host.edit_file(OPS[op], fq, one, *table, n_threads=1)
    want = open(one, "rb").read()
    assert len(want) > 100000
    for chunk in ("100", "4096", "65536"):
        monkeypatch.setenv("YACRD_EDIT_CHUNK", chunk)
        for th in (2, 5):
            out = str(tmp_path / ("par_%s_%d.fastq" % (chunk, th)))
            host.edit_file(OPS[op], fq, out, *table, n_threads=th)
            assert open(out, "rb").read() == want, (op, chunk, th)
    # the output's other ways into the file (default: the threads take turns; all at once; a shared mapping)
    monkeypatch.setenv("YACRD_EDIT_CHUNK", "3000")
    for way in ("pwrite", "map", "turns"):
        monkeypatch.setenv("YACRD_EDIT_OUT", way)
        out = str(tmp_path / ("way_%s.fastq" % way))
        host.edit_file(OPS[op], fq, out, *table, n_threads=4)
        assert open(out, "rb").read() == want, (op, way)
    monkeypatch.delenv("YACRD_EDIT_OUT")
    # FASTA (multi-line sequences): the same records, wrapped at 60 columns
    fa = str(tmp_path / "s.fasta")
    with open(fq) as f, open(fa, "w") as o:
        lines = f.read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            o.write(">" + lines[i][1:] + "\n")
            for k in range(0, len(lines[i + 1]), 60):
                o.write(lines[i + 1][k:k + 60] + "\n")
    monkeypatch.delenv("YACRD_EDIT_CHUNK")
    host.edit_file(OPS[op], fa, one + ".fa", *table, n_threads=1)
    monkeypatch.setenv("YACRD_EDIT_CHUNK", "1000")
    host.edit_file(OPS[op], fa, one + ".par.fa", *table, n_threads=4)
    assert open(one + ".fa", "rb").read() == open(one + ".par.fa", "rb").read()


def test_parallel_editors_tricky_fastq(tmp_path, monkeypatch):
    """Quality lines that begin with '@' or '+', CRLF terminators, blank lines between records, a last record
    without a newline: the chunk boundaries never split a record, the output is the one-thread loop's."""
    rng = np.random.default_rng(5)
    recs = []
    for i in range(400):
        n = int(rng.integers(1, 90))
        seq = "".join(rng.choice(list("ACGT"), n))
        qual = "".join(rng.choice(list("@+?I#"), n))
        if i % 3 == 0:
            qual = "@" + qual[1:]
        if i % 7 == 0:
            qual = "+" + qual[1:]
        nl = "\r\n" if i % 5 == 0 else "\n"
        recs.append("@r%d d%d%s%s%s+%s%s%s" % (i, i, nl, seq, nl, nl, qual, nl) + ("\n" if i % 11 == 0 else ""))
    data = "".join(recs).rstrip("\n")
    src = str(tmp_path / "t.fastq")
    open(src, "w", newline="").write(data)
    reads = {"r%d" % i: [[(0, 3), (5, 9)], 40] for i in range(0, 400, 2)}
    table = table_from_reads(reads, 0, 0.8)
    one = str(tmp_path / "one.fastq")
    host.edit_file(host.OP_SCRUBB, src, one, *table, n_threads=1)
    for chunk in ("64", "333", "5000"):
        monkeypatch.setenv("YACRD_EDIT_CHUNK", chunk)
        out = str(tmp_path / ("p%s.fastq" % chunk))
        host.edit_file(host.OP_SCRUBB, src, out, *table, n_threads=3)
        assert open(out, "rb").read() == open(one, "rb").read(), chunk


def test_parallel_editors_malformed_record_is_the_one_thread_error(tmp_path, monkeypatch):
    good = "".join("@r%d\nACGTACGT\n+\n????????\n" % i for i in range(500))
    bad = good + "@broken\nACGT\n+\n??\n" + good  # sequence / quality lengths differ
    src = str(tmp_path / "b.fastq")
    open(src, "w").write(bad)
    table = table_from_reads({"r1": [[(0, 3)], 8]}, 0, 0.8)
    outs = []
    for th, chunk in ((1, None), (4, "700")):
        if chunk:
            monkeypatch.setenv("YACRD_EDIT_CHUNK", chunk)
        out = str(tmp_path / ("o%d.fastq" % th))
        with pytest.raises(host.HostError) as ei:
            host.edit_file(host.OP_FILTER, src, out, *table, n_threads=th)
        outs.append(str(ei.value))
    assert outs[0] == outs[1] and "fastq format failed" in outs[0]


def test_parallel_editors_stream_to_a_pipe(tmp_path, monkeypatch):
    """A FIFO as output: the chunk-parallel path needs a seekable file, so such an output takes the
    one-thread loop, which streams like the reference's BufWriter — same bytes as to a regular file."""
    import threading
    fq, table = _synthetic_case(tmp_path, n_reads=120, n_ovl=3000)
    want_p = str(tmp_path / "want.fastq")
    host.edit_file(host.OP_SCRUBB, fq, want_p, *table, n_threads=1)
    want = open(want_p, "rb").read()
    monkeypatch.setenv("YACRD_EDIT_CHUNK", "2000")  # (many chunks: the parallel path would be taken for a regular file)
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    host.edit_file(host.OP_SCRUBB, fq, fifo, *table, n_threads=4)
    t.join(timeout=60)
    assert not t.is_alive() and got and got[0] == want
