"""Streaming ingest on the GPU box: records cross PCIe from pinned buffers during the parse, the CSR
is built in HBM (csr_build.h) and swept.  Streamed == one-shot == oracle, bit-exact, through the C
ABI; plus the PCIe-rate input path of yacrd_engine_run and the host-batch pipeline."""
import os

import numpy as np
import pytest

import oracle
import yacrd_amd
from cases import assert_same
from yacrd_amd import host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    with yacrd_amd.Engine() as e:
        yield e


def oracle_for(c, cov, nc):
    return oracle.run(c.offsets, c.intervals, c.lengths.astype(np.uint64), cov, nc, n_threads=4)


@pytest.mark.parametrize("threads,chunk,nbuf", [(1, 0, 0), (4, 1000, 3), (8, 37, 2), (16, 5000, 40)])
def test_streamed_equals_one_shot_equals_oracle(engine, tmp_path, threads, chunk, nbuf):
    paf = str(tmp_path / "s.paf")
    host.synth_paf(host.SYNTH_ONT, 4000, 90000, 20241120, paf)
    ref = host.csr_from_file(paf, n_threads=2)
    want = oracle_for(ref, 4, 0.4)
    one_shot = engine.run(ref.offsets, ref.intervals, ref.lengths, 4, 0.4)
    assert_same(one_shot, want, "one-shot")
    with yacrd_amd.Stream(engine, chunk, nbuf) as st:
        sink = st.sink()
        c = host.ingest_stream(paf, sink, n_threads=threads)
        assert c.names == ref.names and np.array_equal(c.lengths, ref.lengths)
        got = st.finish(c.handle_map, c.lengths, 4, 0.4)
        stats = st.stats()
    assert_same(got, want, "streamed")
    assert stats["n_records"] == 90000 and stats["h2d_bytes"] == 90000 * 24
    assert stats["h2d_busy_ms"] > 0 and stats["build_ms"] > 0


def test_fixture_through_the_stream(engine, golden_dir):
    path = os.path.join(golden_dir, "reads.paf")
    with yacrd_amd.Stream(engine) as st:
        sink = st.sink()
        c = host.ingest_stream(path, sink, n_threads=3)
        got = st.finish(c.handle_map, c.lengths, 0, 0.8)
    lines = oracle.report_from_csr(c.names, c.lengths, got.bad_offsets, got.bad_regions, got.read_type)
    with open(os.path.join(golden_dir, "truth.yacrd")) as f:
        assert set(lines) == set(l.rstrip("\n") for l in f)


def test_stream_is_reusable_and_takes_other_profiles(engine, tmp_path):
    with yacrd_amd.Stream(engine, 2048, 6) as st:
        for prof, R, O, cov in ((host.SYNTH_SEQUEL, 3000, 120000, 3), (host.SYNTH_SKEWED, 40, 90000, 4),
                                (host.SYNTH_ONT, 500, 2000, 0)):
            paf = str(tmp_path / ("p%d.paf" % prof))
            host.synth_paf(prof, R, O, 77 + prof, paf)
            ref = host.csr_from_file(paf, n_threads=4)
            sink = st.sink()
            c = host.ingest_stream(paf, sink, n_threads=6)
            got = st.finish(c.handle_map, c.lengths, cov, 0.4)
            assert_same(got, oracle_for(ref, cov, 0.4), "profile %d" % prof)


def test_raw_records_identity_handles_and_bad_ids(engine):
    """yacrd_stream_* without the parser: records pushed by hand, handles = read ids."""
    rng = np.random.default_rng(5)
    R, N = 300, 20000
    lengths = rng.integers(1000, 50000, R).astype(np.uint32)
    recs = np.zeros(N, dtype=yacrd_amd.OVL_REC_DTYPE)
    recs["a"] = np.sort(rng.integers(0, R, N))  # runs of equal ids, like a PAF grouped by query
    recs["b"] = rng.integers(0, R, N)
    for side, rid in (("a", recs["a"]), ("b", recs["b"])):
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
    with yacrd_amd.Stream(engine, 999, 4) as st:
        st.push(recs)
        assert_same(st.finish(None, lengths, 2, 0.4), want, "identity handles")
        # a handle map that permutes: handle h -> read (R - 1 - h)
        st.push(recs)
        perm = np.arange(R - 1, -1, -1, dtype=np.uint32)
        got = st.finish(perm, lengths[::-1].copy(), 2, 0.4)
        want_p = oracle.run(*_permuted(off, iv, lengths, perm), 2, 0.4, n_threads=2)
        assert_same(got, want_p, "permuted handles")
        bad = recs[:10].copy()
        bad["b"][3] = R + 5
        st.push(bad)
        with pytest.raises(yacrd_amd.EngineError, match="outside"):
            st.finish(None, lengths, 2, 0.4)
        st.push(recs[:100])  # still usable after the error
        st.finish(None, lengths, 2, 0.4)
        assert st.finish(None, lengths[:0], 0, 0.8).bad_offsets.tolist() == [0]  # nothing pushed


def _permuted(off, iv, lengths, perm):
    R = len(lengths)
    order = np.argsort(perm)  # new read id -> old read id
    n = np.diff(off.astype(np.int64))
    new_off = np.zeros(R + 1, np.uint64)
    new_off[1:] = np.cumsum(n[order])
    new_iv = np.concatenate([iv[int(off[o]):int(off[o + 1])] for o in order]) if len(iv) else iv
    return new_off, new_iv, lengths[order].astype(np.uint64)


def test_pinned_and_pageable_inputs_take_the_fast_path(engine):
    """yacrd_engine_run: pinned inputs go by direct DMA, pageable ones through the bounce buffers
    (4 MiB pieces, several copy threads); both bit-exact, and the 6.4 MB of intervals move faster
    than the runtime's single-threaded pageable staging did (9.6 GB/s in round 1)."""
    offsets, intervals, lengths = host.synth_csr(host.SYNTH_ONT, 16000, 1600000, 20241121)
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), 4, 0.4, n_threads=8)
    assert_same(engine.run(offsets, intervals, lengths, 4, 0.4), want, "pageable")
    t_page = engine.timing()["h2d_ms"]
    p_off, p_iv, p_len = (yacrd_amd.PinnedArray.copy_of(x) for x in (offsets, intervals, lengths))
    best = None
    for _ in range(3):
        assert_same(engine.run(p_off.array, p_iv.array, p_len.array, 4, 0.4), want, "pinned")
        t = engine.timing()["h2d_ms"]
        best = t if best is None else min(best, t)
    gbs = intervals.nbytes / (best * 1e-3) / 1e9
    print("pinned H2D %.3f ms = %.1f GB/s; pageable (bounce) %.3f ms" % (best, gbs, t_page))
    assert gbs > 20.0  # PCIe Gen5 x16: ~55 GB/s for large copies; 25 MB is still latency-tinged


def test_submit_collect_pipeline_over_two_engines(engine):
    """Host batches pipelined over two engines: H2D of one overlaps the kernels of the other."""
    batches = [host.synth_csr(host.SYNTH_ONT, 5000, 250000, 100 + k) for k in range(2)]
    wants = [oracle.run(o, iv, ln.astype(np.uint64), 4, 0.4, n_threads=8) for o, iv, ln in batches]
    pinned = [[yacrd_amd.PinnedArray.copy_of(x) for x in b] for b in batches]
    with yacrd_amd.Engine() as e2:
        engs = [engine, e2]
        inflight = [None, None]
        for step in range(8):  # same shapes come back: the third round runs without a plan sync
            j = step % 2
            if inflight[j] is not None:
                assert_same(engs[j].collect(), wants[inflight[j]], "step %d" % step)
            k = (step // 2) % 2
            engs[j].submit(*(p.array for p in pinned[k]), 4, 0.4)
            inflight[j] = k
        for j in range(2):
            assert_same(engs[j].collect(), wants[inflight[j]], "drain")
        with pytest.raises(yacrd_amd.EngineError):
            e2.collect()
