"""Host side of the streaming ingest (yacrd_ingest_stream), the csv-crate record syntax, the
compressed inputs and the id-table capacity: CPU only.  The sink here is a Python stand-in for the
engine's yacrd_stream (same yacrd_rec_sink ABI); the GPU half is tests/test_gpu_stream.py."""
import bz2
import ctypes
import gzip
import lzma
import os
import shutil
import threading

import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import host


class PySink:
    """yacrd_rec_sink over numpy buffers: collects every committed record."""

    def __init__(self, capacity=1000, n_buffers=64):
        self.capacity = capacity
        self.lock = threading.Lock()
        self.bufs = [np.zeros(capacity, dtype=yacrd_amd.OVL_REC_DTYPE) for _ in range(n_buffers)]
        self.free = list(range(n_buffers))
        self.by_addr = {b.ctypes.data: i for i, b in enumerate(self.bufs)}
        self.chunks = []
        self.commits = 0

        def acquire(ctx, buf, cap):
            with self.lock:
                if not self.free:  # grow: a parser thread may hold one buffer each
                    self.bufs.append(np.zeros(capacity, dtype=yacrd_amd.OVL_REC_DTYPE))
                    self.by_addr[self.bufs[-1].ctypes.data] = len(self.bufs) - 1
                    self.free.append(len(self.bufs) - 1)
                i = self.free.pop()
            buf[0] = ctypes.cast(self.bufs[i].ctypes.data, ctypes.POINTER(yacrd_amd.OvlRec))
            cap[0] = capacity
            return 0

        def commit(ctx, buf, n):
            addr = ctypes.cast(buf, ctypes.c_void_p).value
            with self.lock:
                i = self.by_addr[addr]
                self.chunks.append(self.bufs[i][:n].copy())
                self.free.append(i)
                self.commits += 1
            return 0

        self._acq, self._com = yacrd_amd.engine._ACQUIRE(acquire), yacrd_amd.engine._COMMIT(commit)
        self.struct = yacrd_amd.RecSink(None, self._acq, self._com)

    def records(self):
        return np.concatenate(self.chunks) if self.chunks else np.zeros(0, yacrd_amd.OVL_REC_DTYPE)


def per_read_intervals(recs, handle_map, n_reads):
    """What the GPU CSR build computes, in numpy: read id -> sorted list of (start, end)."""
    out = [[] for _ in range(n_reads)]
    for r in recs:
        out[int(handle_map[r["a"]])].append((int(r["sa"]), int(r["ea"])))
        out[int(handle_map[r["b"]])].append((int(r["sb"]), int(r["eb"])))
    return [sorted(x) for x in out]


def csr_intervals(c):
    return [sorted(map(tuple, c.intervals[int(c.offsets[r]):int(c.offsets[r + 1])].tolist()))
            for r in range(c.n_reads)]


@pytest.mark.parametrize("threads", [1, 4])
def test_streamed_records_equal_the_host_csr(golden_dir, threads):
    path = os.path.join(golden_dir, "reads.paf")
    ref = host.csr_from_file(path, n_threads=1)
    sink = PySink(capacity=100)
    c = host.ingest_stream(path, sink.struct, n_threads=threads)
    assert c.streamed and c.offsets is None and c.intervals is None
    assert (c.n_reads, c.n_intervals, c.n_records) == (230, 2572, 1286)
    assert c.names == ref.names and np.array_equal(c.lengths, ref.lengths)
    recs = sink.records()
    assert len(recs) == 1286 and sink.commits >= 13
    used = c.handle_map[c.handle_map != 0xFFFFFFFF]
    assert sorted(used.tolist()) == list(range(230))  # a bijection handles -> read ids
    assert per_read_intervals(recs, c.handle_map, 230) == csr_intervals(ref)


def test_streaming_is_independent_of_threads_and_buffers(tmp_path):
    paf = str(tmp_path / "t.paf")
    host.synth_paf(host.SYNTH_ONT, 5000, 60000, 12, paf)
    ref = host.csr_from_file(paf, n_threads=1)
    want = csr_intervals(ref)
    for th, cap in ((1, 7), (3, 1000), (8, 50000)):
        sink = PySink(capacity=cap)
        c = host.ingest_stream(paf, sink.struct, n_threads=th)
        assert c.names == ref.names and np.array_equal(c.lengths, ref.lengths), (th, cap)
        assert per_read_intervals(sink.records(), c.handle_map, c.n_reads) == want, (th, cap)


def test_streaming_m4_and_memory_entry_point():
    from test_oracle import M4_FILE
    sink = PySink()
    c = host.ingest_stream_memory(M4_FILE, sink.struct, host.FMT_M4, 2)
    assert c.names == ["1", "2", "3"] and c.lengths.tolist() == [12000, 10000, 10000]
    assert per_read_intervals(sink.records(), c.handle_map, 3) == [[(20, 4500), (5500, 10000)], [(5500, 10000)],
                                                                   [(0, 4500)]]
    sink = PySink()
    assert host.ingest_stream_memory("", sink.struct, host.FMT_PAF, 2).n_reads == 0 and not sink.chunks


def test_a_failing_sink_fails_the_ingest(golden_dir):
    sink = PySink()
    bad = yacrd_amd.RecSink(None, yacrd_amd.engine._ACQUIRE(lambda ctx, buf, cap: 1), sink._com)
    with pytest.raises(host.HostError, match="sink"):
        host.ingest_stream(os.path.join(golden_dir, "reads.paf"), bad, n_threads=2)


# ---- csv-crate record syntax (src/reads2ovl/mod.rs:84-88 builder defaults) -----------------------
CSV_CASES = [
    # quoted ids, a doubled quote, a tab inside quotes, text after the closing quote
    '"a b"\t100\t1\t50\t+\t"q""x"\t200\t2\t60\n"t\tab"\t300\t3\t70\t-\t"ab"cd\t400\t4\t80\n',
    # a quote inside an unquoted field is data
    'r"1\t100\t1\t50\t+\tr2"\t200\t2\t60\n',
    # CRLF, lone CR as a terminator, blank records
    "a\t100\t1\t50\t+\tb\t200\t2\t60\r\nb\t999\t3\t70\t-\ta\t888\t4\t80\rc\t5\t0\t1\t+\ta\t1\t5\t6\r\r\n\n",
    # 0x integers and explicit plus signs
    "a\t0x64\t+1\t0x32\t+\tb\t+200\t0x2\t60\n",
    # quoted numbers, a \r inside quotes is data
    '"a\rz"\t"100"\t"1"\t50\t"+"\tb\t200\t2\t60\n',
    #  more than nine columns with CR-only line endings: every \r ends a record
    "a\t100\t1\t50\t+\tb\t200\t2\t60\t49\t58\t255\rc\t300\t3\t70\t-\td\t400\t4\t80\t1\t2\t255\r",
    "a\t100\t1\t50\t+\tb\t200\t2\t60\t49\t58\t255\tcm:i:5\rc\t300\t3\t70\t-\td\t400\t4\t80\n"
    "e\t10\t1\t5\t+\tf\t20\t2\t6\ttp:A:S\r\re\t10\t2\t6\t-\tf\t20\t3\t7\t1\r\n",
    # a quoted trailing column swallows a tab and a \r; a quote inside an unquoted trailing column is data
    'a\t100\t1\t50\t+\tb\t200\t2\t60\t"x\ty\rz"\t7\nc\t5\t0\t1\t+\ta\t100\t5\t6\tk"q\n',
]


@pytest.mark.parametrize("text", CSV_CASES)
@pytest.mark.parametrize("threads", [1, 2])
def test_csv_record_syntax(text, threads):
    names, off, iv, ln = oracle.to_csr(oracle.parse_paf(text))
    c = host.csr_from_memory(text, host.FMT_PAF, threads)
    assert c.names == list(names)
    assert np.array_equal(c.lengths.astype(np.uint64), ln) and np.array_equal(c.offsets, off)
    assert csr_intervals(c) == [sorted(map(tuple, np.asarray(iv)[int(off[r]):int(off[r + 1])].tolist()))
                                for r in range(len(names))]


def test_csv_known_answers():
    c = host.csr_from_memory(CSV_CASES[0], host.FMT_PAF, 1)
    assert c.names == ["a b", 'q"x', "t\tab", "abcd"]
    c = host.csr_from_memory(CSV_CASES[3], host.FMT_PAF, 1)
    assert c.lengths.tolist() == [100, 200] and c.intervals.tolist() == [[1, 50], [2, 60]]
    c = host.csr_from_memory("1 2 0.1 2 0 20 4500 0x2ee0 0 5500 10000 10000\r", host.FMT_M4, 1)
    assert c.lengths.tolist() == [12000, 10000]


@pytest.mark.parametrize("bad", [
    '"a\t100\t1\t50\t+\tb\t200\t2\t60\nx\t1\t0\t1\t+\ty\t1\t0\t1\n',  # the quote spans lines: loud
    "a\t0x\t1\t50\t+\tb\t200\t2\t60\n",                                   # empty hex
    "a\t100\t0x1g\t50\t+\tb\t200\t2\t60\n",
    "a\t100\t1\t0x100000000\t+\tb\t200\t2\t60\n",                         # u32 overflow in hex
    "1 2 0x1p3 2 0 20 4500 12000 0 5500 10000 10000\n",                   # hex float: not Rust's f64
    "r\r2\t100\t1\t50\t+\tb\t200\t2\t60\t1\t2\t255\n",                 # a \r inside an id: a short record
    "a\t100\t1\t50\t+\tb\t200\t2\t60\ttp:A:S\rjunk\n",                   # ... inside a tag column: ditto
    'a\t100\t1\t50\t+\tb\t200\t2\t60\t"open\n',                          # a trailing quote that never closes
])
def test_csv_rejections(bad):
    fmt = host.FMT_M4 if bad.startswith("1 2") else host.FMT_PAF
    with pytest.raises(host.HostError, match="format failed"):
        host.csr_from_memory(bad, fmt, 1)


def test_error_reports_the_line_for_every_thread_count(tmp_path):
    paf = str(tmp_path / "e.paf")
    host.synth_paf(host.SYNTH_ONT, 2000, 40000, 5, paf)
    lines = open(paf).read().split("\n")
    lines[31234] = "broken line"
    open(paf, "w").write("\n".join(lines))
    for th in (1, 4):
        with pytest.raises(host.HostError, match=r"\(line 31235\)"):
            host.csr_from_file(paf, n_threads=th)
        with pytest.raises(host.HostError, match=r"\(line 31235\)"):
            host.ingest_stream(paf, PySink().struct, n_threads=th)


# ---- compressed inputs: niffler sniffs gzip / bzip2 / xz (src/util.rs:57-70) -------------------
def _compressed_copies(src, tmp_path, stem):
    data = open(src, "rb").read()
    out = {}
    for ext, opener in (("gz", gzip.open), ("bz2", bz2.open), ("xz", lzma.open)):
        p = str(tmp_path / ("%s.%s" % (stem, ext)))
        with opener(p, "wb") as f:
            f.write(data)
        out[ext] = p
    return data, out


def test_ingest_bzip2_xz_gzip(golden_dir, tmp_path):
    src = os.path.join(golden_dir, "reads.paf")
    ref = host.csr_from_file(src, n_threads=1)
    _, files = _compressed_copies(src, tmp_path, "reads.paf")
    for ext, p in files.items():
        for th in (1, 3):
            c = host.csr_from_file(p, n_threads=th)
            assert c.names == ref.names and np.array_equal(c.offsets, ref.offsets), ext
            assert np.array_equal(c.intervals, ref.intervals), ext  # line order, like the plain file
        sink = PySink()
        c = host.ingest_stream(p, sink.struct, n_threads=2)
        assert per_read_intervals(sink.records(), c.handle_map, 230) == csr_intervals(ref), ext


def test_multi_member_and_large_compressed_streams(tmp_path):
    paf = str(tmp_path / "big.paf")
    host.synth_paf(host.SYNTH_ONT, 3000, 150000, 3, paf)  # ~11 MB: several decoder blocks
    ref = host.csr_from_file(paf, n_threads=2)
    data = open(paf, "rb").read()
    cut = data.index(b"\n", len(data) // 2) + 1
    for ext, comp in (("gz", gzip.compress), ("bz2", bz2.compress), ("xz", lzma.compress)):
        p = str(tmp_path / ("two.paf." + ext))
        with open(p, "wb") as f:  # two concatenated members / streams
            f.write(comp(data[:cut]))
            f.write(comp(data[cut:]))
        c = host.csr_from_file(p, n_threads=4)
        assert c.names == ref.names and np.array_equal(c.offsets, ref.offsets), ext
        assert np.array_equal(c.intervals, ref.intervals), ext


def test_truncated_compressed_input_is_an_error(golden_dir, tmp_path):
    """A stream cut short must not read as a shorter file (the reference's decoders fail too)."""
    src = os.path.join(golden_dir, "reads.paf")
    _, files = _compressed_copies(src, tmp_path, "t.paf")
    for ext, p in files.items():
        blob = open(p, "rb").read()
        for keep in (len(blob) - 9, len(blob) // 2):
            q = str(tmp_path / ("cut%d.paf.%s" % (keep, ext)))
            open(q, "wb").write(blob[:keep])
            with pytest.raises(host.HostError, match="unexpected end|corrupt|format failed"):
                host.csr_from_file(q, n_threads=2)


def test_editors_read_and_write_bzip2_xz(golden_dir, tmp_path):
    """scrubb of reads.fastq given as .bz2 / .xz: same records out, in the input's compression
    (src/util.rs:72-87), and a truncated input is an error, not a short output."""
    c = host.csr_from_file(os.path.join(golden_dir, "reads.paf"), n_threads=1)
    bo, br, rt = oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), 0, 0.8)
    fq = gzip.open(os.path.join(golden_dir, "reads.fastq.gz"), "rb").read()
    want = gzip.open(os.path.join(golden_dir, "truth.scrubb.fastq.gz"), "rb").read()
    for ext, mod in (("bz2", bz2), ("xz", lzma), ("gz", gzip)):
        src = str(tmp_path / ("reads.fastq." + ext))
        with mod.open(src, "wb") as f:
            f.write(fq)
        dst = str(tmp_path / ("out.fastq." + ext))
        host.edit_file(host.OP_SCRUBB, src, dst, c.names, c.lengths, bo, br, rt)
        magic = open(dst, "rb").read(6)
        assert magic.startswith({"bz2": b"BZh", "xz": b"\xfd7zXZ\x00", "gz": b"\x1f\x8b"}[ext])
        assert mod.open(dst, "rb").read() == want, ext
        blob = open(src, "rb").read()
        cut = str(tmp_path / ("cut.fastq." + ext))
        open(cut, "wb").write(blob[:len(blob) * 2 // 3])
        with pytest.raises(host.HostError):
            host.edit_file(host.OP_SCRUBB, cut, dst, c.names, c.lengths, bo, br, rt)


def test_truncated_gzip_on_a_record_boundary(golden_dir, tmp_path):
    """a .gz whose data ends exactly at a record boundary but whose stream is not
    finished (no trailer) used to read as a clean, shorter file."""
    import zlib
    fq = gzip.open(os.path.join(golden_dir, "reads.fastq.gz"), "rb").read()
    lines = fq.split(b"\n")
    part = b"\n".join(lines[:400]) + b"\n"  # 100 whole records
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    blob = co.compress(part) + co.flush(zlib.Z_FULL_FLUSH)  # all data present, no end of stream
    src = str(tmp_path / "r.fastq.gz")
    open(src, "wb").write(blob)
    c = host.csr_from_file(os.path.join(golden_dir, "reads.paf"), n_threads=1)
    bo, br, rt = oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), 0, 0.8)
    with pytest.raises(host.HostError, match="unexpected end"):
        host.edit_file(host.OP_FILTER, src, str(tmp_path / "o.fastq.gz"), c.names, c.lengths, bo, br, rt)
    paf = open(os.path.join(golden_dir, "reads.paf"), "rb").read()
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    p = str(tmp_path / "r.paf.gz")
    open(p, "wb").write(co.compress(paf) + co.flush(zlib.Z_FULL_FLUSH))
    with pytest.raises(host.HostError, match="unexpected end"):
        host.csr_from_file(p, n_threads=1)


def test_more_than_4m_reads_on_one_thread(tmp_path):
    """`-t 1` used a single-shard id table capped at 2^22 ids."""
    n = 4_400_000
    ids = np.arange(n, dtype=np.int64)
    rows = np.char.add(np.char.add("r", ids[0::2].astype(str)), "\t9\t0\t5\t+\tr")
    rows = np.char.add(np.char.add(rows, ids[1::2].astype(str)), "\t9\t1\t6")
    p = str(tmp_path / "many.paf")
    with open(p, "w") as f:
        f.write("\n".join(rows.tolist()))
        f.write("\n")
    c = host.load_library()
    h = ctypes.c_void_p()
    assert c.yacrd_csr_from_file(p.encode(), 0, 1, ctypes.byref(h)) == 0, c.yacrd_host_last_error()
    v = host._View()
    c.yacrd_csr_get(h, ctypes.byref(v))
    assert (v.n_reads, v.n_intervals) == (n, n)
    lengths = np.ctypeslib.as_array(v.lengths, shape=(n,))
    assert (lengths == 9).all()
    off = np.ctypeslib.as_array(v.name_off, shape=(n + 1,))
    names = ctypes.string_at(v.names, int(off[-1]))
    assert names[:int(off[3])] == b"r0r1r2" and names[int(off[n - 1]):] == b"r%d" % (n - 1)
    c.yacrd_csr_free(h)
