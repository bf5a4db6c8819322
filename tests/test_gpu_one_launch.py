"""YACRD_F_ONE_LAUNCH (csrc/one_batch.h): a short batch as ONE kernel launch — slab-owning workgroups screen their reads,
sort what the screen leaves, scan, compact and classify.  Bit-exact against the oracle through the C ABI; batches the
launch does not handle (a read beyond 256 intervals, a degenerate interval, more regions than the buffer holds) come
out of the default path, bit-exact as well.  Needs an MI355X."""
import numpy as np
import pytest

import oracle
import yacrd_amd
from cases import assert_same, make_csr
from test_gpu_parity import _crafted_screen_reads

pytestmark = pytest.mark.gpu

ONE = yacrd_amd.F_ONE_LAUNCH


def _csr_of(reads):
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(iv) for iv, _ in reads])
    intervals = np.array([p for iv, _ in reads for p in iv], dtype=np.uint32).reshape(-1, 2)
    lengths = np.array([L for _, L in reads], dtype=np.uint32)
    return offsets, intervals, lengths


@pytest.mark.parametrize("prof,R,O,cov", [(0, 10000, 500000, 4), (1, 6000, 600000, 3), (0, 100000, 5000000, 4),
                                          (0, 16, 800, 4), (0, 127, 6000, 0), (0, 129, 6000, 9), (0, 4097, 200000, 4)])
def test_generator_batches(prof, R, O, cov):
    """SURVEY 8d's profiles (configs[1] among them), clamped and jittered (sigma 30 / 100): one launch, bit-exact."""
    from yacrd_amd import host
    for sflags in (0, host.SYNTH_F_JITTER, host.SYNTH_F_JITTER | host.synth_f_sigma(100)):
        o, iv, ln = host.synth_csr(prof, R, O, 5 + cov, flags=sflags)
        want = oracle.run(o, iv, ln.astype(np.uint64), cov, 0.4, n_threads=4)
        huge = bool((np.diff(o.astype(np.int64)) > 256).any())
        with yacrd_amd.Engine(flags=ONE | yacrd_amd.F_COUNT_PREFILTERED) as e:
            for rep in range(2):
                assert_same(e.run(o, iv, ln, cov, 0.4), want, "profile %d R %d synth flags %d rep %d" % (prof, R, sflags, rep))
                t = e.timing()
                assert t["one_launch"] == (0 if huge else 1), t
                if not huge:
                    assert t["prefiltered_reads"] + t["deferred_reads"] <= R  # (reads of < 2 intervals: neither)
                    assert t["deferred_intervals"] >= t["deferred_reads"]


@pytest.mark.parametrize("cov", [0, 1, 3, 4, 5, 8, 300, 0xFFFFFFFF])
def test_screen_edges(cov):
    """The crafted reads on the edges of the screen (test_gpu_parity._crafted_screen_reads) plus short ones (0 .. 64
    intervals: the default path sorts those in registers, here they go through the screen or the 64-lane sort)."""
    reads = _crafted_screen_reads()
    reads += _crafted_screen_reads(ns=(2, 3, 5, 16, 33, 64), Ls=(7, 1000, 65537), steps=(1, 31, 33))
    reads += [([], 100), ([(5, 50)], 100), ([(0, 100)], 100), ([], 0), ([(0, 0)], 0)]
    rng = np.random.default_rng(cov & 0xFFFF)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    o, iv, ln = _csr_of(reads)
    want = oracle.run(o, iv.reshape(-1), ln.astype(np.uint64), cov, 0.4, n_threads=4)
    with yacrd_amd.Engine(flags=ONE) as e:
        assert_same(e.run(o, iv, ln, cov, 0.4), want, "cov %d" % cov)
        # (some of these reads reach the 64-lane sort with intervals its keys cannot express — zero-length pairs at one position —
        # and are handed to the exact path: such a batch comes out of the default route, bit-exact all the same)


def test_random_batches_of_every_slab_shape():
    """Random reads (not screen-friendly: most are sorted), batch sizes around the slab (128) and wavefront (16) edges."""
    rng = np.random.default_rng(11)
    with yacrd_amd.Engine(flags=ONE) as e:
        for R in (1, 2, 15, 16, 17, 127, 128, 129, 255, 256, 257, 1000, 5000):
            reads = []
            for _ in range(R):
                L = int(rng.integers(1, 3000))
                n = int(rng.choice([0, 1, 2, 7, 40, 64, 65, 100, 128, 129, 200, 256]))
                s = rng.integers(0, L, n)
                t = np.minimum(L, s + rng.integers(0, L, n))  # (zero-length and clamped intervals included)
                reads.append((list(zip(s.tolist(), t.tolist())), L))
            o, iv, ln = _csr_of(reads)
            for cov in (0, 2, 6):
                want = oracle.run(o, iv.reshape(-1), ln.astype(np.uint64), cov, 0.4, n_threads=4)
                assert_same(e.run(o, iv, ln, cov, 0.4), want, "R %d cov %d" % (R, cov))


def test_batches_the_launch_does_not_take_go_the_default_way():
    from yacrd_amd import host
    o, iv, ln = host.synth_csr(host.SYNTH_ONT, 3000, 150000, 3)
    n_iv = int(o[-1])
    # (a) one read of 300 intervals in the middle: found on the device, the batch is run again
    n = np.insert(np.diff(o.astype(np.int64)), 1500, 300)
    o2 = np.zeros(len(n) + 1, np.uint64)
    o2[1:] = np.cumsum(n)
    cut = int(o[1500])
    iv2 = np.concatenate([iv.reshape(-1, 2)[:cut], np.array([(0, 900)] * 300, np.uint32), iv.reshape(-1, 2)[cut:]])
    ln2 = np.insert(ln, 1500, 1000).astype(np.uint32)
    assert int(o2[-1]) == n_iv + 300 and len(ln2) == 3001
    want = oracle.run(o2, iv2.reshape(-1), ln2.astype(np.uint64), 4, 0.4, n_threads=4)
    with yacrd_amd.Engine(flags=ONE) as e:
        assert_same(e.run(o2, iv2, ln2, 4, 0.4), want, "a read beyond 256 intervals")
        assert e.timing()["one_launch"] == 0
        # ... and the engine is as good as new for a batch it does take
        want1 = oracle.run(o, iv, ln.astype(np.uint64), 4, 0.4, n_threads=4)
        assert_same(e.run(o, iv, ln, 4, 0.4), want1, "the next batch")
        assert e.timing()["one_launch"] == 1
    # (b) a degenerate interval (start > end): the 64-lane sort rejects the read -> exact path of the default route
    iv3 = iv.reshape(-1, 2).copy()
    k = int(o[700]) + 3
    iv3[k] = (iv3[k][1] + 5, iv3[k][1])
    want = oracle.run(o, iv3.reshape(-1), ln.astype(np.uint64), 4, 0.4, n_threads=4)
    with yacrd_amd.Engine(flags=ONE) as e:
        assert_same(e.run(o, iv3, ln, 4, 0.4), want, "a degenerate interval")
    # (c) more regions than bad_regions holds (4 per read + 1024): every read is a comb of 100 teeth
    R, teeth = 400, 100
    comb = [(10 * j, 10 * j + 5) for j in range(teeth)]
    o4, iv4, ln4 = _csr_of([(comb, 10 * teeth)] * R)
    want = oracle.run(o4, iv4.reshape(-1), ln4.astype(np.uint64), 0, 0.4, n_threads=4)
    assert len(want[1]) > 4 * R + 1024
    with yacrd_amd.Engine(flags=ONE) as e:
        assert_same(e.run(o4, iv4, ln4, 0, 0.4), want, "region overflow")
        assert_same(e.run(o4, iv4, ln4, 0, 0.4), want, "region overflow, buffer grown")
        assert e.timing()["one_launch"] == 1


def test_submit_wait_and_pipelined_engines():
    """Through yacrd_engine_submit_device / yacrd_engine_wait, two engines taking turns."""
    import torch
    from yacrd_amd import host
    batches = [host.synth_csr(host.SYNTH_ONT, 4000 + 300 * i, 200000, 20 + i, flags=host.SYNTH_F_JITTER if i & 1 else 0) for i in range(6)]
    wants = [oracle.run(b[0], b[1], b[2].astype(np.uint64), 4, 0.4, n_threads=4) for b in batches]
    dev = torch.device("cuda", 0)
    keep, dev_batches = [], []
    for o, iv, ln in batches:
        t = [torch.from_numpy(x).to(dev) for x in (o.view(np.int64), iv.view(np.int32).reshape(-1), ln.view(np.int32))]
        keep.append(t)
        dev_batches.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), len(ln), int(o[-1]), 4, 0.4))
    torch.cuda.synchronize()
    engs = [yacrd_amd.Engine(flags=ONE) for _ in range(2)]
    yacrd_amd.run_device_batches(engs, dev_batches, lambda i, e: assert_same(e.fetch(), wants[i], "pipelined batch %d" % i))
    for e in engs:
        assert e.timing()["one_launch"] == 1
        e.close()
