"""Streaming ingest over several engines (yacrd_stream_group, include/yacrd_engine.h): records are routed by
handle mod N while the parser runs, every device builds the CSR of its own reads in HBM and sweeps it, the
results come back in first-appearance order.  N engines on the box's one GPU (as
test_gpu_parity.py::test_read_partitioned_multi_engine does) against the oracle, bit-exact."""
import os

import numpy as np
import pytest

import oracle
import yacrd_amd
from cases import assert_same
from yacrd_amd import host

pytestmark = pytest.mark.gpu


def oracle_for(c, cov, nc):
    return oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), cov, nc, n_threads=4)


@pytest.fixture(scope="module")
def engines():
    es = [yacrd_amd.Engine() for _ in range(5)]
    yield es
    for e in es:
        e.close()


@pytest.mark.parametrize("n,threads,chunk,nbuf", [(1, 4, 0, 0), (2, 1, 1000, 3), (3, 8, 37, 2), (5, 16, 5000, 0)])
def test_group_equals_oracle(engines, tmp_path, n, threads, chunk, nbuf):
    paf = str(tmp_path / "s.paf")
    host.synth_paf(host.SYNTH_ONT, 4000, 90000, 20250301, paf)
    ref = host.csr_from_file(paf, n_threads=2)
    want = oracle_for(ref, 4, 0.4)
    with yacrd_amd.StreamGroup(engines[:n], chunk, nbuf) as grp:
        sink = grp.sink()
        c = host.ingest_stream(paf, sink, n_threads=threads)
        assert c.names == ref.names and np.array_equal(c.lengths, ref.lengths)
        got = grp.finish(c.handle_map, c.lengths, 4, 0.4)
        stats = [grp.stats(d) for d in range(n)]
    assert_same(got, want, "group of %d" % n)
    assert sum(s["reads_owned"] for s in stats) == len(ref.lengths)
    # a record goes to the device of each of its two reads: between one and two copies of it cross PCIe
    moved = sum(s["n_records"] for s in stats)
    assert 90000 <= moved <= 2 * 90000 and (n > 1 or moved == 90000)
    if n > 1:
        assert min(s["reads_owned"] for s in stats) > 0.5 * len(ref.lengths) / n  # handle mod N balances


def test_fixture_through_the_group(engines, golden_dir):
    path = os.path.join(golden_dir, "reads.paf")
    with yacrd_amd.StreamGroup(engines[:3]) as grp:
        c = host.ingest_stream(path, grp.sink(), n_threads=3)
        got = grp.finish(c.handle_map, c.lengths, 0, 0.8)
    lines = oracle.report_from_csr(c.names, c.lengths, got.bad_offsets, got.bad_regions, got.read_type)
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        assert set(lines) == set(l.rstrip("\n") for l in f)


def test_group_is_reusable_takes_other_profiles_and_more_devices_than_reads(engines, tmp_path):
    with yacrd_amd.StreamGroup(engines[:4], 2048, 6) as grp:
        for prof, R, O, cov in ((host.SYNTH_SEQUEL, 3000, 120000, 3), (host.SYNTH_SKEWED, 40, 90000, 4),
                                (host.SYNTH_ONT, 3, 40, 0)):
            paf = str(tmp_path / ("p%d_%d.paf" % (prof, R)))
            host.synth_paf(prof, R, O, 177 + prof, paf)
            ref = host.csr_from_file(paf, n_threads=4)
            c = host.ingest_stream(paf, grp.sink(), n_threads=6)
            got = grp.finish(c.handle_map, c.lengths, cov, 0.4)
            assert_same(got, oracle_for(ref, cov, 0.4), "profile %d" % prof)


def test_raw_records_identity_handles_errors_and_reset(engines):
    rng = np.random.default_rng(15)
    R, N = 301, 20000
    lengths = rng.integers(1000, 50000, R).astype(np.uint32)
    recs = np.zeros(N, dtype=yacrd_amd.OVL_REC_DTYPE)
    recs["a"] = np.sort(rng.integers(0, R, N))
    recs["b"] = rng.integers(0, R, N)
    for side in "ab":
        rid = recs[side]
        s = (rng.random(N) * lengths[rid] * 0.7).astype(np.uint32)
        e = np.minimum(s + 1 + (rng.random(N) * lengths[rid] * 0.5).astype(np.uint32), lengths[rid])
        recs["s" + side], recs["e" + side] = s, e
    per = [[] for _ in range(R)]
    for r in recs:
        per[r["a"]].append((r["sa"], r["ea"]))
        per[r["b"]].append((r["sb"], r["eb"]))
    off = np.zeros(R + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in per])
    iv = np.array([p for x in per for p in x], dtype=np.uint32).reshape(-1, 2)
    want = oracle.run(off, iv, lengths.astype(np.uint64), 2, 0.4, n_threads=2)
    with yacrd_amd.StreamGroup(engines[:3], 999, 4) as grp:
        grp.push(recs)
        assert_same(grp.finish(None, lengths, 2, 0.4), want, "identity handles")
        grp.push(recs[:5000])
        grp.reset()  # forgotten: the next file starts clean
        grp.push(recs)
        assert_same(grp.finish(None, lengths, 2, 0.4), want, "after reset")
        bad = recs[:10].copy()
        bad["b"][3] = R + 5
        grp.push(bad)
        with pytest.raises(yacrd_amd.EngineError, match="outside"):
            grp.finish(None, lengths, 2, 0.4)
        grp.push(recs)  # still usable, and nothing of the failed file is left on any device
        assert_same(grp.finish(None, lengths, 2, 0.4), want, "after an error")
        with pytest.raises(yacrd_amd.EngineError, match="one-to-one"):
            grp.finish(np.zeros(R, np.uint32), lengths, 2, 0.4)
        assert grp.finish(None, lengths[:0], 0, 0.8).bad_offsets.tolist() == [0]
