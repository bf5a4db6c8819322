"""yacrd_engines_ingest_overlaps — the N-GPU form of the device parser (every engine parses a byte range of the text, the
reads are numbered over the whole file on engine 0, every engine sweeps a range of them) — with N engines on ONE device
against the one-engine call, the host parser and the oracle: same names, lengths, regions and types, bit for bit."""
import os

import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import host
from cases import assert_same

pytestmark = pytest.mark.gpu
CHUNK = 4 << 20  # the ranges are cut on 4 MiB boundaries (gpu_paf.hip)


@pytest.fixture(scope="module")
def engines():
    es = [yacrd_amd.Engine() for _ in range(5)]
    yield es
    for e in es:
        e.close()


def _same_ingest(a, b, what):
    (ra, na, la, sa), (rb, nb, lb, sb) = a, b
    assert na == nb, what + ": names / first-appearance order"
    assert np.array_equal(la, lb), what + ": first length seen"
    assert sa["n_records"] == sb["n_records"] and sa["n_reads"] == sb["n_reads"], what
    assert_same(ra, rb, what)


@pytest.mark.parametrize("prof,R,O,cov", [(host.SYNTH_ONT, 20000, 300000, 4), (host.SYNTH_SEQUEL, 3000, 330000, 3)])
def test_group_equals_one_engine_host_parser_and_oracle(engines, tmp_path, prof, R, O, cov):
    paf = str(tmp_path / "s.paf")
    host.synth_paf(prof, R, O, 5 + cov, paf)
    assert os.path.getsize(paf) > 5 * CHUNK
    one = engines[0].ingest_paf(paf, cov, 0.4)
    c = host.csr_from_file(paf, n_threads=4)
    assert c.names == one[1] and np.array_equal(c.lengths, one[2])
    want = oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), cov, 0.4, n_threads=8)
    assert_same(one[0], want, "one engine vs oracle")
    for n in (2, 3, 5):
        got = yacrd_amd.ingest_overlaps(engines[:n], paf, cov, 0.4)
        _same_ingest(got, one, "%d engines" % n)
        assert got[3]["text_bytes"] == os.path.getsize(paf)
    with open(paf, "rb") as f:
        text = f.read()
    _same_ingest(yacrd_amd.ingest_overlaps(engines[:3], text, cov, 0.4), one, "3 engines, text in memory")


def _lines(rng, n, ids, tag=0):
    out = []
    for _ in range(n):
        a, b = int(rng.integers(0, len(ids))), int(rng.integers(0, len(ids)))
        s1, s2 = int(rng.integers(0, 8000)), int(rng.integers(0, 8000))
        out.append("%s\t9000\t%d\t%d\t-\t%s\t9000\t%d\t%d%s\n" % (ids[a], s1, s1 + int(rng.integers(1, 900)), ids[b], s2,
                                                              s2 + int(rng.integers(1, 900)), "\ttg:Z:" + "x" * tag if tag else ""))
    return out


@pytest.mark.parametrize("shape", ["newline_last_byte_of_a_range", "line_starts_a_range", "line_straddles", "long_tag_straddles",
                                   "empty_lines_at_the_cut", "crlf_split_by_the_cut"])
def test_lines_at_the_range_boundaries(engines, tmp_path, shape):
    """Two engines: the cut lies at byte 4 MiB.  A line belongs to the range it starts in, whatever lies at the cut."""
    rng = np.random.default_rng(abs(hash(shape)) % 1000)
    ids = ["read%04d" % i for i in range(700)]
    body = "".join(_lines(rng, 150000, ids))
    assert len(body) > CHUNK + (1 << 20)
    cut = body.rfind("\n", 0, CHUNK - 300)  # a newline a few hundred bytes in front of the cut
    head, tail = body[:cut + 1], body[cut + 1:]
    room = CHUNK - len(head)  # bytes from here up to the cut
    special = "late7\t7000\t10\t900\t+\tread0001\t9000\t5\t800"
    padded = lambda rec, total: rec + "\tp:Z:" + "y" * (total - len(rec) - 5)  # the record with a tag, `total` bytes without its newline
    if shape == "newline_last_byte_of_a_range":
        text = head + padded(special, room - 1) + "\n" + tail
        assert text[CHUNK - 1] == "\n"
    elif shape == "line_starts_a_range":
        text = head + padded("q\t50\t1\t20\t+\tread0002\t9000\t1\t30", room - 1) + "\n" + special + "\n" + tail
        assert text[CHUNK - 1] == "\n" and text[CHUNK:CHUNK + 5] == "late7"
    elif shape == "line_straddles":  # an id that runs over the cut
        text = head + "z" * (room + 6) + "\t8000\t5\t700\t+\tread0003\t9000\t7\t900\n" + tail
        assert text[CHUNK - 1] == "z" and text[CHUNK] == "z"
    elif shape == "long_tag_straddles":
        text = head + padded(special, room + 50000) + "\n" + tail
    elif shape == "empty_lines_at_the_cut":
        text = head + "\n" * (room + 3) + special + "\n" + tail
        assert text[CHUNK - 1] == "\n" and text[CHUNK] == "\n"
    else:  # crlf_split_by_the_cut: '\r' is the range's last byte, '\n' the next one's first
        text = head + padded(special, room - 1) + "\r\n" + tail
        assert text[CHUNK - 1] == "\r" and text[CHUNK] == "\n"
    p = str(tmp_path / "b.paf")
    with open(p, "w", newline="") as f:
        f.write(text)
    one = engines[0].ingest_paf(p, 2, 0.4)
    reads = oracle.parse_paf(text)
    w_names, off, iv, ln = oracle.to_csr(reads)
    assert one[1] == list(w_names)
    for n in (2, 3):
        _same_ingest(yacrd_amd.ingest_overlaps(engines[:n], p, 2, 0.4), one, "%s, %d engines" % (shape, n))
    assert_same(one[0], oracle.run(off, iv, ln, 2, 0.4, n_threads=8), shape)


def test_first_appearance_and_first_length_across_ranges(engines, tmp_path):
    """A read first named in the LAST range is numbered after every read of the ranges before; a read whose length differs
    between two ranges keeps the one seen first in the file (src/reads2ovl/fullmemory.rs:82-90)."""
    rng = np.random.default_rng(3)
    ids = ["r%03d" % i for i in range(300)]
    a = "".join(_lines(rng, 120000, ids))  # > one chunk
    b = "".join(_lines(rng, 120000, ids[:100] + ["only_late%d" % i for i in range(50)]))
    odd = "r005\t1234\t1\t900\t+\tonly_late3\t777\t2\t700\n"  # r005 was 9000 long in front; only_late3 is 777 long HERE first?
    text = a + odd + b
    assert len(a) > CHUNK and len(b) > CHUNK
    p = str(tmp_path / "f.paf")
    with open(p, "w", newline="") as f:
        f.write(text)
    one = engines[0].ingest_paf(p, 3, 0.4)
    reads = oracle.parse_paf(text)
    w_names, off, iv, ln = oracle.to_csr(reads)
    assert one[1] == list(w_names) and np.array_equal(one[2].astype(np.uint64), ln)
    assert one[2][one[1].index("r005")] == 9000
    for n in (2, 3, 4):
        _same_ingest(yacrd_amd.ingest_overlaps(engines[:n], p, 3, 0.4), one, "%d engines" % n)


def test_m4_and_fallbacks_and_small_texts(engines, tmp_path):
    rng = np.random.default_rng(9)
    ids = ["m%04d" % i for i in range(900)]
    lines = []
    for _ in range(200000):
        a, b = int(rng.integers(0, 900)), int(rng.integers(0, 900))
        s1, s2 = int(rng.integers(0, 8000)), int(rng.integers(0, 8000))
        lines.append("%s %s 0.1 2 0 %d %d 9000 1 %d %d 9000\n" % (ids[a], ids[b], s1, s1 + int(rng.integers(1, 900)), s2, s2 + int(rng.integers(1, 900))))
    text = "".join(lines)
    assert len(text) > 2 * CHUNK
    p = str(tmp_path / "f.mhap")
    with open(p, "w") as f:
        f.write(text)
    one = engines[0].ingest_paf(p, 3, 0.4, fmt=2)
    _same_ingest(yacrd_amd.ingest_overlaps(engines[:3], p, 3, 0.4, fmt=0), one, "M4 by file name, 3 engines")
    # something only the host parser handles, in the LAST range only: the whole call hands over
    with open(p, "a") as f:
        f.write('"quoted" m0001 0.1 2 0 1 500 9000 1 3 400 9000\n')
    with pytest.raises(yacrd_amd.NeedsHostParser):
        yacrd_amd.ingest_overlaps(engines[:2], p, 3, 0.4, fmt=2)
    # a text of less than two chunks is engine 0's alone; an empty one too
    small = "a\t100\t1\t50\t+\tb\t200\t2\t60\n"
    got = yacrd_amd.ingest_overlaps(engines[:4], small.encode(), 0, 0.8)
    _same_ingest(got, engines[0].ingest_text(small.encode(), 0, 0.8), "small text")
    got = yacrd_amd.ingest_overlaps(engines[:2], b"", 0, 0.8)
    assert got[1] == [] and len(got[0].read_type) == 0 and int(got[0].bad_offsets[-1]) == 0
    # the engines are usable afterwards, each on its own
    _same_ingest(engines[2].ingest_text(small.encode(), 0, 0.8), engines[0].ingest_text(small.encode(), 0, 0.8), "afterwards")


def test_the_parsers_sort_on_its_own(engines):
    """radix_sort.h through yacrd_debug_sort_pairs: (u64, u32) pairs by key, stable, any number of passes."""
    e = engines[0]
    rng = np.random.default_rng(12)
    for n, bits in [(0, 8), (1, 1), (255, 8), (256, 9), (257, 16), (4096, 12), (4097, 40), (100000, 3), (1 << 20, 37), (300001, 63), (5000, 64)]:
        bound = (1 << bits) if bits < 64 else (1 << 64) - 1
        keys = rng.integers(0, bound, size=n, dtype=np.uint64, endpoint=False) if n else np.zeros(0, np.uint64)
        if n > 10:
            keys[: n // 3] = keys[n // 3: 2 * (n // 3)]  # duplicates: their values must keep the input order
        vals = np.arange(n, dtype=np.uint32)
        k, v = e.debug_sort_pairs(keys, vals, bound)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order]), (n, bits)


_RANGE_FUZZ = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle, yacrd_amd
from cases import assert_same
from test_gpu_ingest import _random_paf
# YACRD_TEST_RANGE_BYTES=32768 (this process): the ranges are cut every 32 KiB, so random texts of a few thousand lines fall
# into 2..5 ranges with their lines — CRLF, empty lines, multi-byte ids, trailing columns — wherever the cuts happen to be
rng = np.random.default_rng(int(sys.argv[2]))
engines = [yacrd_amd.Engine() for _ in range(5)]
taken = fell_back = cut = 0
for case in range(120):
    anomaly = 0 if case % 3 else int(rng.integers(1, 9))
    text = _random_paf(rng, int(rng.integers(1200, 5000)), anomaly)
    raw = text.encode("utf-8")
    n = int(rng.integers(2, 6))
    cov = int(rng.integers(0, 4))
    try:
        one = engines[0].ingest_text(raw, cov, 0.4)
    except yacrd_amd.NeedsHostParser:
        one = None
    try:
        got = yacrd_amd.ingest_overlaps(engines[:n], raw, cov, 0.4)
    except yacrd_amd.NeedsHostParser:
        got = None
    assert (one is None) == (got is None), "case %d: one engine %s, %d engines %s" % (case, one is not None, n, got is not None)
    if one is None:
        fell_back += 1
        assert anomaly != 0
        continue
    taken += 1
    cut += len(raw) > 2 * 32768
    assert got[1] == one[1] and np.array_equal(got[2], one[2]), "case %d: names / lengths" % case
    assert got[3]["n_records"] == one[3]["n_records"]
    assert_same(got[0], one[0], "case %d, %d engines" % (case, n))
    if case % 10 == 0:  # ... and the one-engine result against the oracle's ingest + sweep
        w_names, off, iv, ln = oracle.to_csr(oracle.parse_paf(text))
        assert one[1] == list(w_names)
        assert_same(one[0], oracle.run(off, iv, ln, cov, 0.4, n_threads=2), "case %d vs oracle" % case)
print("RANGE_FUZZ", taken, fell_back, cut)
"""


@pytest.mark.parametrize("seed", [1, 2])
def test_random_texts_cut_every_32_kib(tmp_path, seed):
    """Differential fuzz of the range logic: the N-engine call against the one-engine call on random texts, the ranges cut
    every 32 KiB (YACRD_TEST_RANGE_BYTES) — same reads, lengths, regions, types, and the same texts handed to the host parser."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "range_fuzz.py"
    script.write_text(_RANGE_FUZZ)
    p = subprocess.run([sys.executable, str(script), root, str(seed)], env=dict(os.environ, YACRD_TEST_RANGE_BYTES="32768"),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    last = [l for l in p.stdout.splitlines() if l.startswith("RANGE_FUZZ")][-1].split()
    taken, fell_back, cut = int(last[1]), int(last[2]), int(last[3])
    assert taken >= 70 and fell_back >= 10 and cut >= 50, last


PEER_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import yacrd_amd
from yacrd_amd import host
from yacrd_amd.engine import peer_copy_counts
paf = sys.argv[2]
es = [yacrd_amd.Engine() for _ in range(4)]
one = es[0].ingest_paf(paf, 4, 0.4)
before = peer_copy_counts()
ok = True
for n in (2, 3, 4):
    got = yacrd_amd.ingest_overlaps(es[:n], paf, 4, 0.4)
    ok = ok and got[1] == one[1] and np.array_equal(got[2], one[2])
    ok = ok and all(np.array_equal(a, b) for a, b in zip(got[0], one[0]))
text = open(paf, "rb").read()
got = yacrd_amd.ingest_overlaps(es[:3], text, 4, 0.4)
ok = ok and got[1] == one[1] and all(np.array_equal(a, b) for a, b in zip(got[0], one[0]))
after = peer_copy_counts()
print("RESULT", "OK" if ok else "MISMATCH", [a - b for a, b in zip(after, before)])
"""


@pytest.mark.parametrize("route", ["peer", "staged"])
def test_the_multi_device_branch_forced_on_one_device(tmp_path, route):
    """VERDICT r5, missing #2: hipMemcpyPeerAsync, the host-staged fallback and the gather of other devices' records had never
    run (one GPU per box).  YACRD_TEST_FORCE_PEER_COPY sends EVERY cross-engine copy of the group down the chosen route and
    makes every engine gather the other engines' records instead of reading them in place: the same reads, lengths, regions
    and types as one engine (first length wins across ranges: src/reads2ovl/fullmemory.rs:82-90), and the route's counter says
    the copies went where they were sent."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paf = str(tmp_path / "p.paf")
    host.synth_paf(host.SYNTH_ONT, 20000, 300000, 77, paf)
    assert os.path.getsize(paf) > 4 * CHUNK
    script = tmp_path / "peer_worker.py"
    script.write_text(PEER_WORKER)
    p = subprocess.run([sys.executable, str(script), root, paf], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, YACRD_TEST_FORCE_PEER_COPY=route))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
    assert "RESULT OK" in line, line
    same, peer, staged = eval(line.split("OK", 1)[1])
    assert same == 0, line
    assert (peer > 20 and staged == 0) if route == "peer" else (staged > 20 and peer == 0), line
