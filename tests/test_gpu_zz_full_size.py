"""BASELINE.json's configs[3] and configs[2] at FULL size, every read against the oracle, in the run the driver sees
(missing #3: until now full-size every-read parity was builder-side only, tools/scale_check.py).  The file's
name sorts it last: it is the suite's longest test (~1 min).  Reference semantics: src/stack.rs:61-139 (compute_bad_part),
src/editor/mod.rs:85-100 (type_of_read); partition: src/stack.rs:151-156 (reads are independent)."""
import numpy as np
import pytest

import oracle
import yacrd_amd
from yacrd_amd import dist as ydist, host
from cases import assert_same

pytestmark = pytest.mark.gpu


def _eight_way(e, off, iv, ln, cov, nc, want):
    """What 8 GPUs would each compute (yacrd_partition_reads: contiguous read ranges balanced by interval count), merged
    on the host: the whole input's result."""
    cuts = yacrd_amd.partition_reads(off, 8)
    assert cuts[0] == 0 and cuts[-1] == len(ln) and all(cuts[i] <= cuts[i + 1] for i in range(8))
    bo, br, rt = [np.zeros(1, np.uint64)], [], []
    base = np.uint64(0)
    for k in range(8):
        o, v, l = ydist.local_csr(off, iv, ln, int(cuts[k]), int(cuts[k + 1]))
        got = e.run(np.ascontiguousarray(o), np.ascontiguousarray(v), np.ascontiguousarray(l), cov, nc)
        bo.append(got.bad_offsets[1:] + base)
        base = base + got.bad_offsets[-1]
        br.append(got.bad_regions)
        rt.append(got.read_type)
    merged = (np.concatenate(bo), np.concatenate(br, axis=0), np.concatenate(rt))
    for a, b, what in zip(merged, want, ("bad_offsets", "bad_regions", "read_type")):
        assert np.array_equal(a, b), "8-way partition: " + what
    iv_share = [int(off[cuts[k + 1]] - off[cuts[k]]) for k in range(8)]
    return max(iv_share) / max(1, min(iv_share))


def test_configs3_full_size_every_read():
    """configs[3]: 10 000 ultra-long reads / 30 M overlaps (57 M intervals, the largest read > 700 k), -c 4 -n 0.4."""
    off, iv, ln = host.synth_csr(host.SYNTH_SKEWED, 10_000, 30_000_000, 20241108 + 4)
    assert len(ln) == 10_000 and int(np.diff(off.astype(np.int64)).max()) > 16384  # (the device-wide class is in it)
    want = oracle.run(off, iv, ln.astype(np.uint64), 4, 0.4, n_threads=16)
    with yacrd_amd.Engine() as e:
        assert_same(e.run(off, iv, ln, 4, 0.4), want, "configs[3], full size")
        assert e.timing()["fused_reruns"] == 0
        _eight_way(e, off, iv, ln, 4, 0.4, want)  # (read counts this skewed: the balance is by intervals, looked at on configs[2])


def test_configs2_full_size_every_read():
    """configs[2]: 2 M reads / 200 M overlaps (400 M intervals, 3.2 GB), Sequel-like, -c 3 -n 0.4."""
    off, iv, ln = host.synth_csr(host.SYNTH_SEQUEL, 2_000_000, 200_000_000, 20241108 + 3)
    assert len(ln) == 2_000_000 and int(off[-1]) == 400_000_000
    want = oracle.run(off, iv, ln.astype(np.uint64), 3, 0.4, n_threads=16)
    with yacrd_amd.Engine() as e:
        assert_same(e.run(off, iv, ln, 3, 0.4), want, "configs[2], full size")
        t = e.timing()
        assert t["screened"] == 1 and t["deferred_reads"] < 0.05 * len(ln), t  # (the product path: the screen, the list-driven follow-on)
        imbalance = _eight_way(e, off, iv, ln, 3, 0.4, want)
        assert imbalance < 1.01, imbalance
