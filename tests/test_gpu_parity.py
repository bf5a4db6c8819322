"""HIP path vs CPU oracle, bit-exact, through the C ABI.  Needs an MI355X."""
import os

import numpy as np
import pytest

import oracle
import yacrd_amd
from cases import assert_same, make_csr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    with yacrd_amd.Engine() as e:
        yield e


@pytest.fixture(scope="module")
def engine_general():
    with yacrd_amd.Engine(flags=yacrd_amd.F_FORCE_GENERAL) as e:
        yield e


@pytest.fixture(scope="module")
def engine_lds():
    with yacrd_amd.Engine(flags=yacrd_amd.F_FORCE_LDS_SORT) as e:
        yield e


def check(e, csr, cov, nc, ctx):
    offsets, intervals, lengths = csr
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, nc, n_threads=4)
    got = e.run(offsets, intervals, lengths, cov, nc)
    assert_same(got, want, ctx)
    return got


# ---- reference golden vectors through the GPU -------------------------------------------------
def test_reference_known_answers(engine):
    from test_oracle import STACK_KATS
    for ovl, length, cov, expect in STACK_KATS:
        off = np.array([0, len(ovl)], dtype=np.uint64)
        got = engine.run(off, np.array(ovl, dtype=np.uint32), np.array([length]), cov, 0.8)
        assert got.bad_regions.tolist() == [list(x) for x in expect]


@pytest.mark.parametrize("cov,nc", [(0, 0.8), (1, 0.8), (2, 0.4), (3, 0.4), (4, 0.4)])
def test_fixture_truth(engine, golden_dir, cov, nc):
    with open(os.path.join(golden_dir, "reads.paf")) as f:
        reads = oracle.parse_paf(f)
    names, offsets, intervals, lengths = oracle.to_csr(reads)
    got = check(engine, (offsets, intervals, lengths.astype(np.uint32)), cov, nc, "fixture")
    if cov == 0:
        with open(os.path.join(golden_dir, "truth.yacrd")) as f:
            truth = set(line.rstrip("\n") for line in f)
        lines = oracle.report_from_csr(names, lengths, got.bad_offsets, got.bad_regions,
                                       got.read_type)
        assert set(lines) == truth


# ---- size classes x coverage ------------------------------------------------------------------
REGULAR_MODES = ("regular", "abutting", "dups", "beyond", "sparse", "zero_len")


@pytest.mark.parametrize("cov", [0, 1, 3, 4, 9])
def test_small_class(engine, cov):
    rng = np.random.default_rng(1)
    sizes = np.concatenate([np.arange(0, 40), rng.integers(1, 513, size=1500),
                            [511, 512, 256, 255, 257, 128, 64, 63, 65, 32, 33, 1, 2]])
    check(engine, make_csr(100 + cov, sizes, REGULAR_MODES), cov, 0.4, "small c=%d" % cov)


# ---- coverage pre-filter (sweep_wave.h `prefilter`) -------------------------------------------
@pytest.mark.parametrize("cov", [0, 1, 4, 9, 40])
def test_prefilter_on_deep_pileups(cov):
    """Reads of 65..512 intervals (classes R16 / H16 / W16) piled deep: most bins are safe at small c.
    Same results with the filter (default), without it, and from the oracle; the filter fires."""
    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.integers(65, 513, size=3000), [65, 128, 129, 256, 257, 512]])
    lengths = np.concatenate([rng.integers(1, 200, size=300), rng.integers(200, 150000, size=2706)])
    # the filter is only used by wavefronts whose intervals are all plain (start < end <= len):
    # modes come in blocks of 256 reads, so some wavefronts qualify and some do not
    csr = make_csr(900 + cov, sizes, REGULAR_MODES, lengths=lengths, mode_block=256)
    with yacrd_amd.Engine(flags=yacrd_amd.F_COUNT_PREFILTERED) as e:
        got = check(e, csr, cov, 0.4, "prefilter c=%d" % cov)
        fired = e.timing()["prefiltered_reads"]
    with yacrd_amd.Engine(flags=yacrd_amd.F_NO_PREFILTER | yacrd_amd.F_COUNT_PREFILTERED) as e:
        ref = e.run(*csr, cov, 0.4)
        assert e.timing()["prefiltered_reads"] == 0
    assert_same(got, (ref.bad_offsets, ref.bad_regions, ref.read_type), "prefilter on/off")
    if cov <= 9:
        assert fired > 500, fired
    if cov == 40:   # depth never exceeds c on most of these reads: nothing is safe, nothing dropped
        assert fired < 3006


def test_prefilter_on_synthetic_profiles():
    from yacrd_amd import host
    for prof, R, O, cov in ((host.SYNTH_ONT, 4000, 200000, 4), (host.SYNTH_SEQUEL, 3000, 300000, 3)):
        off, iv, ln = host.synth_csr(prof, R, O, 99)
        with yacrd_amd.Engine(flags=yacrd_amd.F_COUNT_PREFILTERED) as e:
            check(e, (off, iv.reshape(-1, 2), ln), cov, 0.4, "synthetic profile %d" % prof)
            assert e.timing()["prefiltered_reads"] > R // 2


@pytest.mark.parametrize("cov", [0, 4])
def test_small_class_lds_variant(engine_lds, cov):
    rng = np.random.default_rng(2)
    sizes = np.concatenate([np.arange(0, 20), rng.integers(1, 513, size=600)])
    check(engine_lds, make_csr(150 + cov, sizes, REGULAR_MODES), cov, 0.4, "small-lds")


@pytest.mark.parametrize("cov", [0, 3, 4])
def test_medium_classes(engine, cov):
    rng = np.random.default_rng(3)
    sizes = np.concatenate([rng.integers(513, 4097, size=40), rng.integers(4097, 16385, size=12),
                            [513, 4096, 4097, 16384, 2048, 8192]])
    check(engine, make_csr(200 + cov, sizes, REGULAR_MODES, len_lo=20000, len_hi=400000), cov,
          0.4, "medium c=%d" % cov)


@pytest.mark.parametrize("cov", [0, 4, 60])
def test_medium_classes_prefilter(cov):
    """Workgroup-LDS classes (513..16384 intervals) on deep pile-ups with tiny, short and long
    reads: with the pre-filter, without it, and the oracle agree."""
    rng = np.random.default_rng(31)
    sizes = np.concatenate([rng.integers(513, 4097, size=40), rng.integers(4097, 16385, size=14),
                            [513, 4096, 4097, 16384]])
    lengths = np.concatenate([rng.integers(1, 300, size=6), rng.integers(300, 5000, size=10),
                              rng.integers(5000, 900000, size=42)])
    csr = make_csr(260 + cov, sizes, ("regular", "abutting", "dups", "sparse", "beyond", "zero_len"),
                   lengths=lengths)
    with yacrd_amd.Engine() as e:
        got = check(e, csr, cov, 0.4, "medium prefilter c=%d" % cov)
    with yacrd_amd.Engine(flags=yacrd_amd.F_NO_PREFILTER) as e:
        ref = e.run(*csr, cov, 0.4)
    assert_same(got, (ref.bad_offsets, ref.bad_regions, ref.read_type), "medium prefilter on/off")


@pytest.mark.parametrize("cov", [0, 4])
def test_general_class_large_reads(engine, cov):
    sizes = [16385, 20000, 40000, 70001, 5, 300]
    check(engine, make_csr(300 + cov, sizes, ("regular", "abutting"), len_lo=200000,
                           len_hi=1000000), cov, 0.4, "large c=%d" % cov)


@pytest.mark.parametrize("cov", [0, 3])
def test_big_path_with_degenerate_reads(engine, cov):
    """> 16384 intervals: device-wide segmented sort; degenerate giants fall back to the exact kernel."""
    sizes = [20000, 17000, 33000, 16385, 131072, 100, 65536]
    check(engine, make_csr(350 + cov, sizes, ("degenerate", "regular", "huge_pos", "dups", "sparse"),
                           len_lo=300000, len_hi=900000), cov, 0.4, "big+degenerate c=%d" % cov)


@pytest.mark.parametrize("cov", [0, 1, 2, 4])
def test_degenerate_and_huge_positions(engine, cov):
    rng = np.random.default_rng(4)
    sizes = np.concatenate([np.arange(1, 30), rng.integers(1, 600, size=400),
                            rng.integers(600, 5000, size=10)])
    check(engine, make_csr(400 + cov, sizes, ("degenerate", "regular", "huge_pos", "degenerate")),
          cov, 0.4, "degenerate c=%d" % cov)


@pytest.mark.parametrize("cov", [0, 2, 4])
def test_general_kernel_on_everything(engine_general, cov):
    """The exact general kernel must agree with the oracle on regular reads too."""
    rng = np.random.default_rng(5)
    sizes = np.concatenate([np.arange(0, 30), rng.integers(1, 700, size=300),
                            rng.integers(700, 6000, size=8)])
    modes = REGULAR_MODES + ("degenerate", "huge_pos")
    check(engine_general, make_csr(500 + cov, sizes, modes), cov, 0.4, "general-all c=%d" % cov)


def test_paths_agree_with_each_other(engine, engine_general, engine_lds):
    rng = np.random.default_rng(6)
    sizes = rng.integers(0, 513, size=800)
    csr = make_csr(600, sizes, REGULAR_MODES)
    a = engine.run(*csr, 3, 0.4)
    b = engine_general.run(*csr, 3, 0.4)
    c = engine_lds.run(*csr, 3, 0.4)
    assert_same(a, (b.bad_offsets, b.bad_regions, b.read_type), "wave vs general")
    assert_same(c, (b.bad_offsets, b.bad_regions, b.read_type), "lds vs general")


# ---- edge cases -------------------------------------------------------------------------------
def test_region_buffer_outgrown_and_counters_handed_over():
    """bad_regions starts at 4 R + 1024 entries; reads with hundreds of gaps each outgrow it: the follow-on kernel's
    last slab hands the counters over with the overflow flag derived from the total (finish_compact.h), the host
    grows the buffer and redoes the follow-on kernel only.  A fresh engine per case (buffers only grow)."""
    for n_reads, per_read, cov in ((50, 400, 0), (1500, 120, 0), (3000, 9, 0), (40, 3000, 1)):
        rng = np.random.default_rng(n_reads)
        offs, ivs, lens = [0], [], []
        for _ in range(n_reads):
            starts = np.sort(rng.choice(np.arange(0, 40 * per_read, 20), size=per_read, replace=False))
            iv = np.stack([starts, starts + rng.integers(1, 10, size=per_read)], axis=1)  # disjoint: a gap after each
            if cov:
                iv = np.repeat(iv, 2, axis=0)  # depth 2 > c = 1 inside every interval
            ivs.append(iv)
            offs.append(offs[-1] + len(iv))
            lens.append(40 * per_read + 50)
        off = np.array(offs, np.uint64)
        iv = np.concatenate(ivs).astype(np.uint32)
        ln = np.array(lens, np.uint32)
        want = oracle.run(off, iv, ln.astype(np.uint64), cov, 0.4, n_threads=2)
        assert int(want[0][-1]) > 4 * n_reads + 1024  # the case does outgrow the first allocation
        with yacrd_amd.Engine() as e:
            assert_same(e.run(off, iv, ln, cov, 0.4), want, "first run %d x %d" % (n_reads, per_read))
            assert_same(e.run(off, iv, ln, cov, 0.4), want, "second run")


def test_empty_batch(engine):
    got = engine.run(np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint32),
                     0, 0.8)
    assert got.bad_offsets.tolist() == [0] and got.bad_regions.shape == (0, 2)


def test_reads_without_intervals_and_zero_length(engine):
    offsets = np.array([0, 0, 0, 2, 2], dtype=np.uint64)
    intervals = np.array([[0, 10], [5, 20]], dtype=np.uint32)
    lengths = np.array([1000, 0, 20, 7], dtype=np.uint32)
    check(engine, (offsets, intervals, lengths), 0, 0.8, "empty reads")


def test_coverage_saturation(engine):
    csr = make_csr(700, [50, 200, 3], ("regular",))
    check(engine, csr, 0xFFFFFFFF, 0.4, "c=u32 max")
    check(engine, csr, 1000, 0.0, "c=1000 n=0")


def test_classification_thresholds(engine):
    """-n edge: strict >, NaN for len 0, NotCovered before Chimeric (editor/mod.rs:85-100)."""
    offsets = np.array([0, 1, 2, 4], dtype=np.uint64)
    intervals = np.array([[0, 600], [0, 599], [0, 300], [700, 1000]], dtype=np.uint32)
    lengths = np.array([1000, 1000, 1000], dtype=np.uint32)
    for nc in (0.4, 0.401, 0.399999, 0.0, 1.0):
        check(engine, (offsets, intervals, lengths), 0, nc, "thresholds n=%r" % nc)


def test_standalone_classify_kernel(engine):
    csr = make_csr(800, np.random.default_rng(8).integers(1, 300, size=500), REGULAR_MODES)
    got = check(engine, csr, 2, 0.4, "classify")
    for nc in (0.1, 0.4, 0.8):
        rt = engine.classify(got.bad_offsets, got.bad_regions, csr[2], nc)
        want = np.array([oracle.type_of_read(int(csr[2][r]),
                                             got.bad_regions[int(got.bad_offsets[r]):
                                                             int(got.bad_offsets[r + 1])].tolist(),
                                             nc) for r in range(len(csr[2]))], dtype=np.uint8)
        assert np.array_equal(rt, want)


def test_repeated_runs_are_deterministic(engine):
    csr = make_csr(900, np.random.default_rng(9).integers(0, 513, size=3000), REGULAR_MODES)
    a = engine.run(*csr, 4, 0.4)
    for _ in range(3):
        b = engine.run(*csr, 4, 0.4)
        assert_same(b, (a.bad_offsets, a.bad_regions, a.read_type), "rerun")


def test_device_resident_entry_point(engine):
    torch = pytest.importorskip("torch")
    csr = make_csr(1000, np.random.default_rng(10).integers(0, 400, size=2000), REGULAR_MODES)
    offsets, intervals, lengths = csr
    d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_iv = torch.from_numpy(intervals.astype(np.int64).astype(np.int32) if False else
                            intervals.view(np.int32)).cuda()
    d_len = torch.from_numpy(lengths.view(np.int32)).cuda()
    torch.cuda.synchronize()
    out = engine.run_device(d_off.data_ptr(), d_iv.data_ptr(), d_len.data_ptr(), len(lengths),
                            int(offsets[-1]), 4, 0.4)
    got = engine.fetch()
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), 4, 0.4)
    assert int(out.n_regions) == int(want[0][-1])
    assert_same(got, want, "run_device")
    t = engine.timing()
    assert t["n_small"] == len(lengths) and max(t["class_ms"] + [t["fused_ms"]]) > 0


def test_read_partitioned_multi_engine(engine):
    """contiguous read ranges, one engine each, no collective; identical output."""
    csr = make_csr(1100, np.random.default_rng(11).integers(0, 700, size=3000),
                   REGULAR_MODES + ("degenerate",))
    want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), 3, 0.4, n_threads=4)
    with yacrd_amd.Engine() as e2, yacrd_amd.Engine() as e3:
        for engines in ([engine], [engine, e2], [engine, e2, e3]):
            got = yacrd_amd.run_partitioned(engines, *csr, 3, 0.4)
            assert_same(got, want, "partitioned x%d" % len(engines))


@pytest.mark.parametrize("flags", [0, yacrd_amd.F_ALWAYS_DEFER,
                                   yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2])
def test_class_prediction_is_validated(flags):
    """Runs of identical shape (reads, intervals) reuse the previous run's class set instead of
    waiting for the plan; a batch whose classes differ must still come out bit-exact.  With the
    deferring build as well: its remainder launches (SweepArgs.first > 0) mark and finish their own
    reads."""
    R = 600
    a_sizes = [100] * R                                    # everything in one class
    b_sizes = [10] * 300 + [190] * 299 + [60000 - 3000 - 299 * 190]  # same totals, other classes
    assert sum(a_sizes) == sum(b_sizes) == 60000
    c_sizes = [2] * 599 + [60000 - 2 * 599]                # one read beyond the LDS classes
    # same classes, other proportions: a predicted grid (count + 12.5 % + 64) that is too short
    d_sizes = [100] * 200 + [36] * 200 + [164] * 200
    e_sizes = [100] * 400 + [36] * 100 + [164] * 100
    assert sum(d_sizes) == sum(e_sizes) == 60000
    with yacrd_amd.Engine(flags=flags) as e, yacrd_amd.Engine(flags=flags | yacrd_amd.F_NO_PREDICTION) as ref:
        for rep, sizes in enumerate([a_sizes, a_sizes, b_sizes, b_sizes, c_sizes, a_sizes, c_sizes,
                                     d_sizes, d_sizes, e_sizes, e_sizes, a_sizes, d_sizes]):
            csr = make_csr(1200 + rep, sizes, REGULAR_MODES + ("degenerate",), len_lo=300000,
                           len_hi=900000)
            want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), 3, 0.4, n_threads=4)
            assert_same(e.run(*csr, 3, 0.4), want, "predicted run %d" % rep)
            assert_same(ref.run(*csr, 3, 0.4), want, "unpredicted run %d" % rep)


def _crafted_screen_reads(ns=(65, 100, 128, 129, 200, 256), Ls=(1, 7, 15, 16, 17, 33, 1000, 65537, 10**6),
                          steps=(1, 3, 6, 7, 8, 15, 16, 31, 32, 33, 200)):
    """Reads on the edges of the healthy-read screen: n in the R16 / H16 classes
    (65..256 intervals) by default, piles of k identical or nested intervals around k = c, c + 1, c + 2,
    windows (pmin > 0, pmax < len), one hole, unbalanced piles, lengths below the bin count, ends at len."""
    reads = []

    def pad(iv, n, L):  # fill up to n intervals with deep copies of the first interval
        return iv + [iv[0]] * (n - len(iv))
    for n in ns:
        for L in Ls:
            full = (0, L)
            reads.append(([full] * n, L))                                   # everything spans everything
            if L >= 4:
                w = (L // 4, max(L // 4 + 1, L - L // 4))
                reads.append(([w] * n, L))                                  # a window: pmin > 0, pmax < len
                reads.append((pad([full], n - 3, L) + [(0, L // 2)] * 3, L))   # three ends inside
                reads.append((pad([full], n - 3, L) + [(L // 2, L)] * 3, L))   # three starts inside
                reads.append(([(0, L // 2)] * (n // 2) + [(L // 2, L)] * (n - n // 2), L))      # abutting halves: a hole of depth 0
                reads.append(([(0, L // 2 + 1)] * (n // 2) + [(L // 2, L)] * (n - n // 2), L))  # overlapping by one position
            if L >= 40:
                # dovetail pile-up with k intervals clamped at each end and the rest inside
                for k in (1, 2, 3, 4, 5, 6, 9):
                    inner = [(1 + (j % 7), L - 1 - (j % 5)) for j in range(n - 2 * k)]
                    reads.append(([(0, L - 2)] * k + inner + [(2, L)] * k, L))
                    reads.append(([(0, L - 2)] * k + inner + [(2, L)] * (k + 1), L))  # piles out of balance
            if L >= 1000:
                # spread piles (round 3: the screen works on order statistics): the j-th dovetail starts at
                # j * step and the j-th from the other side ends at L - j * step, for steps that put the
                # (c+1)-th of them inside, on the edge of and beyond the screen's windows (32 positions)
                for step in steps:
                    for k in (3, 5, 6, 12):
                        if (k - 1) * step + 320 >= L - 320 - k:
                            continue
                        left = [(j * step, L - 300 - j) for j in range(k)]
                        right = [(250 + j, L - j * step) for j in range(k)]
                        inner = [(260 + (j % 7), L - 310 - (j % 5)) for j in range(n - 2 * k)]
                        reads.append((left + inner + right, L))
                        reads.append((left + inner + right[:-1] + [(L - 20, L)], L))   # one interval shorter than a window
                        reads.append((left + inner[:-1] + right + [(400, 900)], L))     # an internal start where only the piles cover
    return reads


@pytest.mark.parametrize("cov", [0, 1, 2, 3, 4, 5, 8, 300, 0xFFFFFFFF])
def test_healthy_screen_edges(cov):
    reads = _crafted_screen_reads()
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, 0.4, n_threads=4)
    for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_NO_DEFER, yacrd_amd.F_NO_PREFILTER,
                  yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2):
        with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
            assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "cov %d flags %d" % (cov, flags))
            if flags == yacrd_amd.F_ALWAYS_DEFER:
                t = e.timing()
                assert t["prefiltered_reads"] > 0 and t["deferred_reads"] > 0  # both sides of the screen are exercised


@pytest.mark.parametrize("prof,cov", [(0, 4), (1, 3), (0, 0), (1, 9)])
def test_screen_on_jittered_profiles(prof, cov):
    """dovetail ends spread over a few dozen positions (YACRD_SYNTH_F_JITTER: reflected
    instead of clamped onto 0 / len) — no exact position holds a pile.  Bit-exact against the oracle in
    every build, and the screen still finishes >= 90 % of the reads of the two classes in closed form at
    BASELINE's thresholds (it finished 2 % of them when it needed c + 1 starts on ONE position)."""
    from yacrd_amd import host
    R, O = (6000, 300000) if prof == 0 else (3000, 300000)
    # Round 4: sigma = 100 and 300 — the (c+1)-th smallest start lies beyond the screen's
    # 32-position window for a fifth / most of the ONT-depth reads; the windows slide (sweep_wave.h: kScreenSlides) and
    # the screen still decides >= 90 % of them.
    for sflags in (host.SYNTH_F_JITTER, host.SYNTH_F_JITTER | host.synth_f_sigma(8),
                   host.SYNTH_F_JITTER | host.synth_f_sigma(100), host.SYNTH_F_JITTER | host.synth_f_sigma(300)):
        o, iv, ln = host.synth_csr(prof, R, O, 77 + cov, flags=sflags)
        want = oracle.run(o, iv, ln.astype(np.uint64), cov, 0.4, n_threads=4)
        n = np.diff(o.astype(np.int64))
        in_classes = int(((n > 64) & (n <= 256)).sum())
        wide = yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_WIDE  # the build with the second looks (sliding windows)
        for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_NO_DEFER, yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2, wide):
            with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
                assert_same(e.run(o, iv, ln, cov, 0.4), want, "profile %d synth flags %d flags %d" % (prof, sflags, flags))
                t = e.timing()
                # sigma = 30: every build decides >= 90 %; sigma = 100 / 300: the build with the second looks
                # does (>= 85 % at 300); the default builds take their first batch without them (the next test)
                s8 = sflags == (host.SYNTH_F_JITTER | host.synth_f_sigma(8))
                s300 = sflags == (host.SYNTH_F_JITTER | host.synth_f_sigma(300))
                if cov in (3, 4) and not s8 and flags & yacrd_amd.F_ALWAYS_DEFER and (flags == wide or sflags == host.SYNTH_F_JITTER):
                    assert t["deferred_reads"] <= in_classes * (15 if s300 else 10) // 100, (sflags, flags, t["deferred_reads"], in_classes)
                    assert t["prefiltered_reads"] >= in_classes * (85 if s300 else 90) // 100
                if flags & yacrd_amd.F_ALWAYS_DEFER:
                    assert t["screen_wide"] == (1 if flags == wide else 0)


@pytest.mark.parametrize("always", [True, False])
def test_second_looks_are_switched_on_by_the_deferral_rate(always):
    """Round 4: the default builds of the screen take ONE look (a tenth faster on reads whose dovetail ends lie within the
    32-position window); a batch that leaves more than a tenth of its screened reads to the sort switches the engine to the
    build with the second looks (sliding windows + ramp) for the next 15 batches — spread ends (sigma = 100) and back."""
    from yacrd_amd import host
    R, O = (6000, 300000) if always else (50000, 2500000)  # (default flags: the screen runs from 4 M intervals on)
    spread = host.synth_csr(host.SYNTH_ONT, R, O, 81, flags=host.SYNTH_F_JITTER | host.synth_f_sigma(100))
    tight = host.synth_csr(host.SYNTH_ONT, R, O, 82, flags=host.SYNTH_F_JITTER)
    wants = [oracle.run(b[0], b[1], b[2].astype(np.uint64), 4, 0.4, n_threads=4) for b in (spread, tight)]
    tenth = [int(((np.diff(b[0].astype(np.int64)) > 64) & (np.diff(b[0].astype(np.int64)) <= 256)).sum()) // 10 for b in (spread, tight)]
    with yacrd_amd.Engine(flags=yacrd_amd.F_ALWAYS_DEFER if always else 0) as e:
        seen = []
        for i, which in enumerate([0, 0, 0, 1] + [1] * 16 + [0, 0]):
            b = (spread, tight)[which]
            assert_same(e.run(*b, 4, 0.4), wants[which], "batch %d" % i)
            t = e.timing()
            assert t["screened"] == 1, (i, t)                  # (the sorting build never takes over here)
            seen.append((t["screen_wide"], t["deferred_reads"], which))
        assert seen[0][0] == 0 and seen[0][1] > tenth[0], seen  # the first spread batch: one look, > 10 % left to the sort
        assert seen[1][0] == 1 and seen[1][1] <= tenth[0], seen # the second: the build with the second looks, <= 10 %
        assert all(w == 1 for w, _, _ in seen[1:16]), seen      # ... which stays for 15 batches, tight ones included,
        assert seen[16][0] == 0 and seen[17][0] == 0, seen      # then the default build is tried again (tight batches: it stays)
        assert seen[-2][0] == 0 and seen[-1][0] == 1, seen      # and a spread batch switches the second looks on once more


@pytest.mark.parametrize("cov", [0, 4, 5, 11, 300, 0xFFFFFFFF])
def test_workgroup_screen_edges(cov):
    """The same edges for the workgroup classes' screen (screen_wg.h: one read per workgroup, 128-position
    windows): reads of 513 .. 16 384 intervals, one or two register chunks, windows' edges at 127 / 128 / 129."""
    reads = _crafted_screen_reads(ns=(513, 4096, 4097, 8192, 8193, 16384), Ls=(15, 1000, 65537, 10**6),
                                  steps=(1, 7, 31, 32, 33, 42, 43, 127, 128, 129))
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, 0.4, n_threads=8)
    for flags in (0, yacrd_amd.F_NO_PREFILTER, yacrd_amd.F_NO_FUSED_SCREEN, yacrd_amd.F_STREAM_SCREEN):
        with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
            assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "cov %d flags %d" % (cov, flags))
            if flags != yacrd_amd.F_NO_PREFILTER and cov <= 11:
                assert e.timing()["prefiltered_reads"] > len(reads) // 10  # the screen fires


@pytest.mark.parametrize("cov", [0, 3, 30])
def test_workgroup_fallback_queue(cov):
    """Round 4: the workgroup classes' screen and its fallback in ONE persistent launch (screen_wg.h:
    screen_wg_fused_kernel) — what the screen cannot decide goes through a queue in global memory to whichever
    workgroup has finished screening.  Batches where most reads fail the screen (sparse / abutting / duplicate
    pile-ups, degenerate intervals: the rejection list), more reads than the grid has workgroups, repeated runs on one
    engine (the second is launched on the first's class counts), the three-launch chain beside it."""
    rng = np.random.default_rng(808 + cov)
    sizes = np.concatenate([rng.integers(513, 3000, size=1500), rng.integers(4097, 16385, size=60), [513, 4096, 4097, 8192, 8193, 16384]])
    csr = make_csr(5150 + cov, sizes, ("sparse", "regular", "abutting", "dups", "zero_len", "degenerate", "beyond"),
                   len_lo=20000, len_hi=600000, mode_block=7)
    want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), cov, 0.4, n_threads=8)
    for flags in (0, yacrd_amd.F_NO_FUSED_SCREEN, yacrd_amd.F_STREAM_SCREEN):  # (F_STREAM_SCREEN: one wavefront per read first, round 6's A/B)
        with yacrd_amd.Engine(flags=flags) as e:
            for rep in range(3):
                assert_same(e.run(*csr, cov, 0.4), want, "fallback queue: cov %d flags %d run %d" % (cov, flags, rep))
    # skewed profile (configs[3] shape, reduced): screen + queue + the device-wide screen beside it
    from yacrd_amd import host
    o, iv, ln = host.synth_csr(host.SYNTH_SKEWED, 400, 1200000, 31 + cov)
    w2 = oracle.run(o, iv, ln.astype(np.uint64), cov, 0.4, n_threads=8)
    for flags in (0, yacrd_amd.F_STREAM_SCREEN):
        with yacrd_amd.Engine(flags=flags) as e:
            for rep in range(2):
                assert_same(e.run(o, iv, ln, cov, 0.4), w2, "skewed, flags %d, run %d" % (flags, rep))


def test_fused_workgroup_screen_needs_no_resident_grid(tmp_path):
    """Rounds 4-5: screen_wg_fused_kernel's workgroups WAITED for each other's queue entries, which needed the whole grid
    resident (a bounded wait, a give-up flag, the batch run again).  Round 6: the queue is drained by
    compare-and-swap and nobody waits, so a grid of any size works.  With eight times the workgroups the class needs
    (YACRD_TEST_FUSED_GRID_MULT, read once per process: a subprocess), far more than are ever resident, and a share of one
    read per workgroup: bit-exact, nothing run twice, and quick."""
    import subprocess
    import sys
    code = r"""
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, yacrd_amd
from cases import assert_same, make_csr
rng = np.random.default_rng(77)
sizes = np.concatenate([rng.integers(513, 3000, size=3000), rng.integers(4097, 9000, size=700)])
csr = make_csr(4711, sizes, ("sparse", "regular", "abutting", "dups"), len_lo=20000, len_hi=600000, mode_block=7)
want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), 3, 0.4, n_threads=8)
with yacrd_amd.Engine() as e:
    for rep in range(3):
        t0 = time.time()
        assert_same(e.run(*csr, 3, 0.4), want, "oversized grid, run %%d" %% rep)
        t = e.timing()
        assert t["fused_reruns"] == 0, t["fused_reruns"]
        assert time.time() - t0 < 20
print("oversized grid: ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    for share in ("1", "7"):
        env = dict(os.environ, YACRD_TEST_FUSED_GRID_MULT="8", YACRD_FUSED_SHARE=share)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0 and "oversized grid: ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("cov", [0, 4, 600])
def test_device_wide_screen_edges(cov):
    """screen_big.h (reads of more than 16 384 intervals, 512-position windows): piles spread so that the
    (c+1)-th start / end lies inside, on the edge of and beyond the windows, a read covered only inside a
    window, zero-length intervals in the deep middle / in a pile / doubled, a one-position interval at the
    very end (a start inside the tail window), a shallow block in the middle, reads the screen must refuse."""
    rng = np.random.default_rng(99)
    reads = []
    L = 300000
    for n in (16385, 20000, 70000):
        deep = [(int(s), int(s) + 150000) for s in rng.integers(600, 140000, size=n - 40)]
        for step in (1, 100, 511, 512, 513, 5000):
            left = [(j * step, L - 90000 - j) for j in range(20)]
            right = [(80000 + j, L - j * step) for j in range(20)]
            base = left + deep + right
            reads.append((base, L))
            reads.append((base[:-1] + [(L - 1, L)], L))                         # a start inside the tail window
            reads.append((base[:-2] + [(150000, 150000)] * 2, L))                # zero-length, doubled, where the read is deep
            reads.append((base[:-1] + [(0, 0)], L))                             # ... at position 0
            reads.append((base[:-1] + [(step, step)], L))                       # ... inside the head pile
            reads.append((base[:-1] + [(3, 2)], L))                             # a reversed interval: the exact path's
        win = [(100000 + int(a), 200000 - int(b)) for a, b in rng.integers(0, 300, size=(n, 2))]
        reads.append((win, L))                                                  # covered only inside a window
        holed = [(int(s), int(s) + 60000) for s in rng.integers(0, 80000, size=n // 2)] + \
                [(int(s), int(s) + 60000) for s in rng.integers(160000, 240000, size=n - n // 2)]
        reads.append((holed, L))                                                # nothing covers the middle
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, 0.4, n_threads=8)
    for flags in (0, yacrd_amd.F_NO_PREFILTER):
        with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
            assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "cov %d flags %d" % (cov, flags))
            assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "cov %d flags %d, again" % (cov, flags))
            if flags == 0 and cov <= 4:
                assert e.timing()["prefiltered_reads"] > len(reads) // 6  # the screen fires


def test_deferral_rate_swings_between_batches():
    """The follow-on kernel sorts what the screen marked, slab by slab (finish_compact.h): batches of one
    shape whose share of marked reads swings between none and nearly all (far more than a wavefront's
    worth per 1024-read slab, so every wavefront loops over several items) must come out bit-exact —
    through run() and through submit / wait — and the count of deferred reads is exact."""
    import torch
    R, n, L = 3000, 100, 5000
    healthy = [(0, L)] * n
    holed = [(0, L // 2)] * (n // 2) + [(L // 2, L)] * (n - n // 2)

    def batch(n_holed):
        iv = np.array([p for r in range(R) for p in (holed if r < n_holed else healthy)], dtype=np.uint32)
        off = np.arange(R + 1, dtype=np.uint64) * np.uint64(n)
        return off, iv, np.full(R, L, dtype=np.uint32)
    batches = [batch(0), batch(0), batch(2000), batch(2000), batch(10), batch(2900)]
    wants = [oracle.run(b[0], b[1], b[2].astype(np.uint64), 4, 0.4, n_threads=4) for b in batches]
    flags = yacrd_amd.F_ALWAYS_DEFER
    with yacrd_amd.Engine(flags=flags) as e:
        for i, b in enumerate(batches):
            assert_same(e.run(*b, 4, 0.4), wants[i], "run %d" % i)
            t = e.timing()
            assert t["deferred_reads"] == [0, 0, 2000, 2000, 10, 2900][i]
            assert t["deferred_intervals"] == t["deferred_reads"] * n and t["screened"] == 1
    dev = torch.device("cuda", 0)
    keep, dev_batches = [], []
    for o, iv, ln in batches:
        t = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), iv.view(np.int32).reshape(-1), ln.view(np.int32))]
        keep.append(t)
        dev_batches.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), R, int(o[-1]), 4, 0.4))
    torch.cuda.synchronize()
    engs = [yacrd_amd.Engine(flags=flags) for _ in range(2)]
    yacrd_amd.run_device_batches(engs, dev_batches,
                                 lambda i, e: assert_same(e.fetch(), wants[i], "pipelined batch %d" % i))
    for e in engs:
        e.close()


def test_build_choice_follows_the_share_of_bad_reads():
    """Default flags, batches big enough for the screening build (>= 4 M intervals in R16 + H16): with the
    generator's 2 % chimeras every batch goes through the screen; with 40 % (YACRD_SYNTH_F_CHIMERA_PCT) more than a
    quarter of a screened batch comes back deferred: the second batch tries the build with the second looks (they do not
    help chimeras), then the engine takes the sorting build for the next 15 batches and probes the screen again with the
    18th (engine.hip: wide_left / nodefer_left / kProbeEvery).  Bit-exact either way."""
    from yacrd_amd import host
    for pct, want_screened in ((0, 18), (40, 3)):
        off, iv, ln = host.synth_csr(host.SYNTH_ONT, 60000, 3000000, 31 + pct, host.synth_f_chimera_pct(pct))
        want = oracle.run(off, iv, ln.astype(np.uint64), 4, 0.4, n_threads=8)
        with yacrd_amd.Engine() as e:
            e.timing_total(reset=True)
            for i in range(18):
                got = e.run(off, iv, ln, 4, 0.4)
                if i in (0, 1, 2, 16, 17):
                    assert_same(got, want, "%d %% chimeras, batch %d" % (pct, i))
            t, runs = e.timing_total()
            assert runs == 18 and t["screened"] == want_screened, (pct, t["screened"])


def test_device_batches_pipeline():
    """yacrd_engines_run_device_batches: a list of device-resident batches over 1-3 engines, every
    batch fetched in its callback and compared with the oracle; a callback can stop the loop."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    host_batches = [make_csr(7000 + i, rng.integers(0, 300, size=int(rng.integers(200, 1500))), REGULAR_MODES)
                    for i in range(4)]
    host_batches.append(host_batches[1])  # a repeated shape: the predicted path
    host_batches.append(host_batches[1])
    wants = [oracle.run(b[0], b[1], b[2].astype(np.uint64), 3, 0.4, n_threads=4) for b in host_batches]
    keep, dev_batches = [], []
    for o, iv, ln in host_batches:
        t = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), np.ascontiguousarray(iv).view(np.int32).reshape(-1) if len(iv) else np.zeros(2, np.int32), ln.view(np.int32))]
        keep.append(t)
        dev_batches.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(ln), int(o[-1]), 3, 0.4))
    torch.cuda.synchronize()
    for n_eng in (1, 2, 3):
        engs = [yacrd_amd.Engine() for _ in range(n_eng)]
        seen = []

        def done(i, e):
            assert_same(e.fetch(), wants[i], "engines %d batch %d" % (n_eng, i))
            seen.append(i)
            return False
        for rep in range(2):
            del seen[:]
            last = yacrd_amd.run_device_batches(engs, dev_batches, done)
            assert seen == list(range(len(dev_batches))) and last.n_reads == len(host_batches[-1][2])
        with pytest.raises(yacrd_amd.EngineError, match="callback"):
            yacrd_amd.run_device_batches(engs, dev_batches, lambda i, e: i == 2)
        assert_same(engs[0].run(*host_batches[0], 3, 0.4), wants[0], "engine usable after a stopped loop")
        for e in engs:
            e.close()


def test_pipelined_engines_on_one_device():
    """Several engines on one GPU driven by one host thread each (bench.py's pipeline): the engines
    take turns with the dominant sweep, waits sleep (YACRD_F_BLOCKING_WAIT); every run of every
    engine is bit-exact."""
    import threading
    rng = np.random.default_rng(77)
    batches = [make_csr(4000 + i, rng.integers(0, 300, size=1500), REGULAR_MODES) for i in range(3)]
    wants = [oracle.run(b[0], b[1], b[2].astype(np.uint64), 3, 0.4, n_threads=4) for b in batches]
    errors = []

    def work(i):
        try:
            with yacrd_amd.Engine(flags=yacrd_amd.F_BLOCKING_WAIT) as e:
                for rep in range(40):
                    k = (i + rep) % 3
                    assert_same(e.run(*batches[k], 3, 0.4), wants[k], "engine %d rep %d" % (i, rep))
        except Exception as ex:  # surfaced in the main thread
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def test_submit_wait_pipeline_and_misprediction():
    """yacrd_engine_submit_device / _wait on two engines from one thread: batches of the same
    shape pipeline; a batch whose classes do not fit the previous run's prediction (same reads and
    intervals, other proportions; reads for the device-wide path; degenerate reads) comes out
    bit-exact through wait's fallback."""
    import torch
    R = 600
    a_sizes = [100] * R
    d_sizes = [100] * 200 + [36] * 200 + [164] * 200
    e_sizes = [100] * 400 + [36] * 100 + [164] * 100
    c_sizes = [2] * 599 + [60000 - 2 * 599]
    seq = [a_sizes, a_sizes, a_sizes, d_sizes, d_sizes, e_sizes, e_sizes, c_sizes, a_sizes, a_sizes]
    batches, wants, dev = [], [], []
    for rep, sizes in enumerate(seq):
        csr = make_csr(5200 + rep, sizes, REGULAR_MODES + ("degenerate",), len_lo=300000, len_hi=900000)
        batches.append(csr)
        wants.append(oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), 3, 0.4, n_threads=4))
        dev.append((torch.from_numpy(csr[0].view(np.int64)).cuda(),
                    torch.from_numpy(np.ascontiguousarray(csr[1]).view(np.int32)).cuda(),
                    torch.from_numpy(csr[2].view(np.int32)).cuda()))
    torch.cuda.synchronize()
    with yacrd_amd.Engine() as e0, yacrd_amd.Engine() as e1:
        engs, inflight = (e0, e1), [None, None]
        for k in range(len(seq) + 2):
            j = k % 2
            if inflight[j] is not None:
                out = engs[j].wait()
                want = wants[inflight[j]]
                assert int(out.n_regions) == int(want[0][-1])
                assert_same(engs[j].fetch(), want, "submit/wait batch %d" % inflight[j])
                inflight[j] = None
            if k < len(seq):
                d = dev[k]
                engs[j].submit_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), R,
                                      int(batches[k][0][-1]), 3, 0.4)
                inflight[j] = k
        # between submit and wait the engine takes no other call
        d = dev[0]
        args = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), R, int(batches[0][0][-1]), 3, 0.4)
        e0.run_device(*args)
        e0.submit_device(*args)
        with pytest.raises(yacrd_amd.EngineError):
            e0.run_device(*args)
        with pytest.raises(yacrd_amd.EngineError):
            e0.fetch()
        e0.wait()
        assert_same(e0.fetch(), wants[0], "after wait")


# ---- configs[3]: the skewed profile (ultra-long reads, >= 5 k intervals each, a Zipf tail past the
# workgroup-LDS cap) at reduced R: M2 through the 256-thread kernel, its overflow list, BIG ----------
@pytest.mark.parametrize("cov", [4, 0, 40])
def test_skewed_profile_config4_shape(engine, cov):
    from yacrd_amd import host
    offsets, intervals, lengths = host.synth_csr(host.SYNTH_SKEWED, 300, 900000, 20241112)
    n = np.diff(offsets.astype(np.int64))
    assert n.min() >= 5000 and (n > 16384).sum() >= 2 and lengths.min() >= 200000
    check(engine, (offsets, intervals, lengths), cov, 0.4, "skewed c=%d" % cov)
    t = engine.timing()
    names = yacrd_amd.CLASS_NAMES
    assert t["class_reads"][names.index("M2")] > 250 and t["class_reads"][names.index("BIG")] >= 2


def test_skewed_profile_without_prefilter_and_with_degenerates():
    from yacrd_amd import host
    offsets, intervals, lengths = host.synth_csr(host.SYNTH_SKEWED, 120, 360000, 7)
    intervals = intervals.copy()
    rng = np.random.default_rng(3)
    for r in rng.choice(120, 6, replace=False):  # a zero-length and a reversed interval in some reads
        a = int(offsets[r])
        intervals[a + 5] = (intervals[a + 5][0], intervals[a + 5][0])
        intervals[a + 9] = (intervals[a + 9][1], intervals[a + 9][0])
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), 4, 0.4, n_threads=8)
    for flags in (0, yacrd_amd.F_NO_PREFILTER):
        with yacrd_amd.Engine(flags=flags) as e:
            assert_same(e.run(offsets, intervals, lengths, 4, 0.4), want, "flags %d" % flags)


# ---- the deferring build of the fused register-sort kernel (healthy-read screen + closed form, every
# other read marked and finished by sweep_deferred_kernel) against the oracle and the
# other build ----
@pytest.mark.parametrize("cov", [0, 1, 4, 9, 40])
def test_fused_defer_build(cov):
    from yacrd_amd import host
    rng = np.random.default_rng(11)
    sizes = np.concatenate([rng.integers(65, 257, size=3000), [65, 128, 129, 256], rng.integers(1, 65, size=300)])
    lengths = np.concatenate([rng.integers(1, 300, size=300), rng.integers(300, 150000, size=3004)])
    csr = make_csr(1900 + cov, sizes, REGULAR_MODES, lengths=lengths, mode_block=256)
    want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), cov, 0.4, n_threads=4)
    for prof, R, O in ((host.SYNTH_ONT, 6000, 300000), (host.SYNTH_SEQUEL, 3000, 300000)):
        o, iv, ln = host.synth_csr(prof, R, O, 5 + cov)
        w2 = oracle.run(o, iv, ln.astype(np.uint64), cov, 0.4, n_threads=4)
        two = yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2           # two groups of list entries per wavefront
        for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_NO_DEFER, two):
            with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
                assert_same(e.run(o, iv, ln, cov, 0.4), w2, "profile %d flags %d" % (prof, flags))
                if flags == two:  # and a second, predicted run
                    assert_same(e.run(o, iv, ln, cov, 0.4), w2, "profile %d flags %d, predicted" % (prof, flags))
                    continue
                t = e.timing()
                if cov <= 4:
                    assert t["prefiltered_reads"] > R // 2
                if flags == yacrd_amd.F_ALWAYS_DEFER:  # healthy + deferred = the two classes' reads
                    n = np.diff(o.astype(np.int64))
                    both = t["prefiltered_reads"] + t["deferred_reads"]  # (+ filtered reads of the one-read-per-wavefront class)
                    assert 0 < t["deferred_reads"] and 0 <= both - int(((n > 64) & (n <= 256)).sum()) <= int((n > 256).sum())
                else:
                    assert t["deferred_reads"] == 0
    for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_NO_DEFER, two):
        with yacrd_amd.Engine(flags=flags) as e:
            assert_same(e.run(*csr, cov, 0.4), want, "cases flags %d" % flags)
            assert_same(e.run(*csr, cov, 0.4), want, "cases flags %d, predicted run" % flags)


@pytest.mark.parametrize("prof,cov", [(0, 4), (1, 3), (1, 0), (0, 9)])
def test_hole_closed_form_on_chimeras(prof, cov):
    """Round 4: a read with ONE stretch of low coverage inside — a chimera — can get its three regions in closed form
    from the screen (sweep_wave.h: hole_form, a build option; tests/formulation.py::hole_fast_regions is the emulation).
    30 % chimeras, clamped and spread piles, both builds of the screen: bit-exact with or without it; with it most
    chimeras no longer reach the sort."""
    from yacrd_amd import host
    R, O = (6000, 300000) if prof == 0 else (3000, 300000)
    for sflags in (host.synth_f_chimera_pct(30), host.SYNTH_F_JITTER | host.synth_f_chimera_pct(30),
                   host.SYNTH_F_JITTER | host.synth_f_sigma(100) | host.synth_f_chimera_pct(30)):
        o, iv, ln = host.synth_csr(prof, R, O, 123 + cov, flags=sflags)
        want = oracle.run(o, iv, ln.astype(np.uint64), cov, 0.4, n_threads=4)
        chimeric = int((want[2] == 1).sum())
        for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2, 0):
            with yacrd_amd.Engine(flags=flags) as e:
                for rep in range(2):
                    assert_same(e.run(o, iv, ln, cov, 0.4), want, "profile %d cov %d synth flags %d flags %d run %d" % (prof, cov, sflags, flags, rep))
                t = e.timing()
                # (the library is built without hole_form by default — sweep_wave.h: YK_HOLE_FORM — where it costs the
                # screen more than it saves the sort; YACRD_TEST_HOLE_FORM=1 when testing a build that has it)
                if os.environ.get("YACRD_TEST_HOLE_FORM") == "1" and flags and cov in (3, 4) and not (sflags & (0xFF << 8)):
                    assert chimeric > R // 5 and t["deferred_reads"] < chimeric // 2, (t["deferred_reads"], chimeric)


def test_hole_closed_form_fuzz():
    """The emulation's fuzz (tests/test_formulation.py::test_hole_screen_matches_oracle) on the kernel itself: chimeras with
    gaps of any width, intervals left spanning the junction, piles spread or not, coarse grids, five thresholds."""
    from test_formulation import _chimera_read, _pile_read, _survey_read
    rng = np.random.default_rng(4711)
    reads = []
    for it in range(4000):
        L = int(rng.integers(100, 900)) if it % 7 == 0 else int(rng.integers(900, 60000))
        n = int(rng.integers(65, 257))
        jitter = (0.0, 5.0, 30.0, 100.0)[it % 4]
        base = _survey_read if it % 2 else _pile_read
        iv = _chimera_read(rng, n, L, jitter, base) if it % 3 else base(rng, n, L, jitter)
        if it % 13 == 0:
            g = max(1, L // 16)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        reads.append((iv, L))
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    for cov in (0, 1, 3, 4, 9):
        want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, 0.4, n_threads=8)
        for flags in (yacrd_amd.F_ALWAYS_DEFER, yacrd_amd.F_ALWAYS_DEFER | yacrd_amd.F_SCREEN_ITEMS_2):
            with yacrd_amd.Engine(flags=flags) as e:
                assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "hole fuzz cov %d flags %d" % (cov, flags))


def test_fused_workgroup_screens_of_several_engines_take_turns():
    """Round 4: screen_wg_fused_kernel is a persistent grid sized to be resident as a whole; the engines that share a device
    launch theirs one after the other (engine.hip: g_fused_lane) — three engines, batches of workgroup-class reads large enough
    to fill the device, submitted without waiting: every result bit-exact, nothing hangs."""
    rng = np.random.default_rng(99)
    sizes = rng.integers(513, 2500, size=2500)
    csr = make_csr(6161, sizes, ("regular", "sparse", "abutting", "dups"), len_lo=20000, len_hi=300000, mode_block=5)
    want = oracle.run(csr[0], csr[1], csr[2].astype(np.uint64), 3, 0.4, n_threads=8)
    engs = [yacrd_amd.Engine() for _ in range(3)]
    try:
        for rep in range(4):
            for e in engs:  # three batches in flight: H2D, kernels and D2H of one overlap the others'
                e.submit(*csr, 3, 0.4)
            for j, e in enumerate(engs):
                assert_same(e.collect(), want, "engine %d, round %d" % (j, rep))
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("cov", [0, 1, 4, 9])
def test_workgroup_classes_filtered_sweep_and_inert_intervals(cov):
    """Round 6 (screen_wg.h): what the workgroup classes' screen leaves goes through wg_filtered_read — the screen's table read
    once more, only the unsafe bins' events sorted (tests/formulation.py::unified_filtered_regions) — and only what THAT
    leaves through sweep_lds_read.  Chimeras (one and two junctions, abutting, with spanning intervals), reads covered only
    inside a window, zero-length intervals at the window's first position (inert at c >= 1 when a regular interval starts
    there: drop_inert_at_pmin), in a junction's bin (the sort's), at pmin with no regular start there (the sort's)."""
    rng = np.random.default_rng(6600 + cov)
    reads = []
    for k in range(260):
        n = int(rng.choice([600, 1500, 4096, 4097, 5600, 9000, 16384]))
        L = int(rng.integers(60000, 900000))
        lo, hi = (0, L) if k % 3 else (int(L * 0.3), int(L * 0.7))  # the whole read, or a window
        span = hi - lo
        iv = []
        for j in range(n):
            u = rng.random()
            if u < 0.3:
                s, e = lo, lo + int(rng.integers(500, max(501, int(0.8 * span))))
            elif u < 0.6:
                e = hi
                s = hi - int(rng.integers(500, max(501, int(0.8 * span))))
            else:
                s = lo + int(rng.integers(0, max(1, span - 600)))
                e = min(hi, s + int(rng.integers(500, max(501, span // 2))))
            iv.append((max(lo, s), min(hi, max(e, s + 1))))
        if k % 2:  # a junction (or two): crossing intervals cut back to the side of their midpoint
            for jn in range(1 + (k % 4 == 3)):
                j = lo + int(span * rng.uniform(0.2, 0.8))
                gap = int(rng.integers(0, 120))
                iv = [((s, min(e, j - gap)) if (s + e) // 2 < j else (max(s, j + (gap if k % 8 != 1 else 0)), e)) if s < j < e and rng.random() > 0.002 * (k % 5 == 0) else (s, e)
                      for s, e in iv]
                iv = [(s, max(e, s + 1)) if s < hi else (hi - 1, hi) for s, e in iv]
        if k % 5 == 1:
            iv.append((lo, lo))                       # a degenerate interval clamped onto the window's first position
        if k % 5 == 2:
            iv += [(lo, lo)] * (cov + 1)              # several of them
        if k % 13 == 3:
            iv = [(s + 1, e) if s == lo else (s, e) for s, e in iv] + [(lo, lo)]  # ... and no regular start there
        if k % 11 == 4:
            p = iv[len(iv) // 2][1]
            iv.append((p, p))                         # a zero-length interval somewhere inside
        reads.append((iv[:16384], L))
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    want = oracle.run(offsets, intervals, lengths.astype(np.uint64), cov, 0.4, n_threads=8)
    assert int(want[2].sum()) > 0  # (chimeric / not covered reads are in it)
    for flags in (0, yacrd_amd.F_NO_FUSED_SCREEN, yacrd_amd.F_STREAM_SCREEN, yacrd_amd.F_NO_PREFILTER):
        with yacrd_amd.Engine(flags=flags | yacrd_amd.F_COUNT_PREFILTERED) as e:
            assert_same(e.run(offsets, intervals, lengths, cov, 0.4), want, "cov %d flags %d" % (cov, flags))
            if flags == 0:
                c = e.debug_counters()
                # most of the reads are decided without the whole-read sort: by the screen or by the filtered sweep
                assert e.timing()["prefiltered_reads"] > len(reads) // 2, (e.timing()["prefiltered_reads"], c["fb_med"])
                assert sum(c["fb_med"]) < len(reads) // 2, c["fb_med"]


def test_batches_with_device_wide_reads_are_predicted():
    """Round 6: a batch that holds reads beyond 16 384 intervals is launched on the previous run's class counts too (it used to
    wait for the plan's counts every time); the device-wide screen goes out for the predicted count and intervals of such reads.
    Same batch again: predicted, same result.  A batch of the SAME shape (reads, intervals) whose huge reads differ — one of
    20 000 intervals against two of 10 000 — is a miss: found at the final sync, made good, bit-exact either way."""
    sizes_a = [20000, 600, 600, 600] + [700] * 40
    sizes_b = [10000, 10000, 1200, 600] + [700] * 40
    assert len(sizes_a) == len(sizes_b) and sum(sizes_a) == sum(sizes_b)
    a = make_csr(9101, np.array(sizes_a), ("regular",), len_lo=300000, len_hi=900000, mode_block=1)
    b = make_csr(9102, np.array(sizes_b), ("regular",), len_lo=300000, len_hi=900000, mode_block=1)
    wa = oracle.run(a[0], a[1], a[2].astype(np.uint64), 3, 0.4, n_threads=4)
    wb = oracle.run(b[0], b[1], b[2].astype(np.uint64), 3, 0.4, n_threads=4)
    with yacrd_amd.Engine() as e:
        seen = []
        for i, (csr, want) in enumerate([(a, wa), (a, wa), (a, wa), (b, wb), (b, wb), (a, wa), (a, wa)]):
            assert_same(e.run(*csr, 3, 0.4), want, "run %d" % i)
            t = e.timing()
            seen.append((t["predicted"], t["prediction_misses"]))
        assert seen[0][0] == 0 and seen[1] == (1, 0) and seen[2] == (1, 0), seen   # the same batch again: predicted, no miss
        assert seen[3][1] == 1 and seen[4] == (1, 0), seen                        # other huge reads under the same shape: a miss, then none
        assert seen[5][1] == 1 and seen[6] == (1, 0), seen
