"""Seeded CSR generators for the parity tests (numpy only; no oracle, no engine)."""
import numpy as np


def make_read(rng, n, length, mode="regular"):
    """n intervals on a read of `length`, pile-up like: dovetails anchored at an end + internal."""
    L = int(length)
    if n == 0:
        return np.zeros((0, 2), dtype=np.uint32)
    kind = rng.random(n)
    ell = rng.integers(1, max(2, int(0.8 * L)), size=n)
    start = np.empty(n, dtype=np.int64)
    end = np.empty(n, dtype=np.int64)
    left = kind < 0.3
    right = (kind >= 0.3) & (kind < 0.6)
    inner = kind >= 0.6
    jit = np.rint(rng.normal(0, 30, size=n)).astype(np.int64)
    start[left] = jit[left]
    end[left] = start[left] + ell[left]
    end[right] = L + jit[right]
    start[right] = end[right] - ell[right]
    s_in = rng.integers(0, max(1, L), size=n)
    start[inner] = s_in[inner]
    end[inner] = start[inner] + np.minimum(ell[inner], max(1, L // 2))
    start = np.clip(start, 0, max(L - 1, 0))
    end = np.clip(end, 0, L)
    end = np.maximum(end, start + 1)  # regular: start < end (may exceed L by 1 when L tiny)
    iv = np.stack([start, end], axis=1)

    if mode == "regular":
        pass
    elif mode == "abutting":  # chains e_i == s_j -> zero-length gaps
        k = max(1, n // 4)
        idx = rng.choice(n, size=k, replace=False)
        src = rng.choice(n, size=k)
        iv[idx, 0] = iv[src, 1]
        iv[idx, 1] = iv[idx, 0] + rng.integers(1, max(2, L // 4), size=k)
    elif mode == "dups":
        k = max(1, n // 3)
        idx = rng.choice(n, size=k, replace=False)
        src = rng.choice(n, size=k)
        iv[idx] = iv[src]
    elif mode == "beyond":  # ends (and some starts) past the read length
        k = max(1, n // 5)
        idx = rng.choice(n, size=k, replace=False)
        iv[idx, 1] = L + rng.integers(0, 50, size=k)
        idx2 = rng.choice(n, size=max(1, k // 3), replace=False)
        iv[idx2, 0] = L + rng.integers(0, 20, size=idx2.size)
        iv[idx2, 1] = iv[idx2, 0] + rng.integers(1, 20, size=idx2.size)
    elif mode == "degenerate":  # start == end and start > end, incl. value 0 (sentinel reset)
        k = max(1, n // 6)
        idx = rng.choice(n, size=k, replace=False)
        which = rng.random(k)
        eq = which < 0.5
        iv[idx[eq], 1] = iv[idx[eq], 0]
        iv[idx[~eq], 1] = rng.integers(0, np.maximum(iv[idx[~eq], 0], 1))
        if rng.random() < 0.5:
            iv[idx[0]] = (rng.integers(0, max(1, L)), 0)
        if rng.random() < 0.3:
            iv[idx[-1]] = iv[idx[0]]
    elif mode == "zero_len":  # start == end only (the fast paths take these)
        k = max(1, n // 6)
        idx = rng.choice(n, size=k, replace=False)
        iv[idx, 1] = iv[idx, 0]
        if rng.random() < 0.3:
            iv[idx[0]] = (0, 0)
        if rng.random() < 0.3:
            iv[idx[-1]] = (L, L)
        if rng.random() < 0.2 and k > 1:
            iv[idx[1]] = iv[idx[0]]  # duplicate zero-length interval -> exact path
    elif mode == "huge_pos":  # positions >= 2^31 (32-bit event keys cannot hold them)
        k = max(1, n // 8)
        idx = rng.choice(n, size=k, replace=False)
        iv[idx, 0] = rng.integers(2**31 - 5, 2**32 - 10, size=k)
        iv[idx, 1] = np.minimum(iv[idx, 0] + rng.integers(1, 9, size=k), 2**32 - 1)
    elif mode == "sparse":  # low coverage: many holes
        iv[:, 1] = np.minimum(iv[:, 0] + rng.integers(1, max(2, L // (n + 1) + 2), size=n), L)
        iv[:, 1] = np.maximum(iv[:, 1], iv[:, 0] + 1)
    else:
        raise ValueError(mode)
    return iv.astype(np.uint32)


def make_csr(seed, sizes, modes=("regular",), len_lo=500, len_hi=60000, lengths=None, mode_block=1):
    """sizes: iterable of interval counts per read; read r is of mode modes[(r // mode_block) %
    len(modes)].  Returns offsets u64, intervals u32[I,2], lengths u32."""
    rng = np.random.default_rng(seed)
    sizes = list(sizes)
    R = len(sizes)
    if lengths is None:
        lengths = rng.integers(len_lo, len_hi, size=R)
    lengths = np.asarray(lengths, dtype=np.uint64)
    offsets = np.zeros(R + 1, dtype=np.uint64)
    parts = []
    for r, n in enumerate(sizes):
        mode = modes[(r // mode_block) % len(modes)]
        parts.append(make_read(rng, int(n), int(lengths[r]), mode))
        offsets[r + 1] = offsets[r] + np.uint64(n)
    intervals = (np.concatenate(parts, axis=0) if parts else np.zeros((0, 2), np.uint32))
    return offsets, intervals.astype(np.uint32), lengths.astype(np.uint32)


def assert_same(got, want, ctx=""):
    """got: yacrd_amd.Result; want: (bad_offsets, bad_regions, read_type) from the oracle."""
    bo, br, rt = want
    if not np.array_equal(got.bad_offsets, bo):
        bad = int(np.nonzero(np.diff(got.bad_offsets.astype(np.int64)) !=
                             np.diff(bo.astype(np.int64)))[0][0])
        raise AssertionError("%s: region count differs first at read %d: got %d want %d" % (
            ctx, bad, int(got.bad_offsets[bad + 1] - got.bad_offsets[bad]),
            int(bo[bad + 1] - bo[bad])))
    if not np.array_equal(got.bad_regions, br):
        k = int(np.nonzero((got.bad_regions != br).any(axis=1))[0][0])
        r = int(np.searchsorted(bo, k, side="right") - 1)
        a, b = int(bo[r]), int(bo[r + 1])
        raise AssertionError("%s: regions differ at read %d: got %s want %s" % (
            ctx, r, got.bad_regions[a:b].tolist(), br[a:b].tolist()))
    if not np.array_equal(got.read_type, rt):
        r = int(np.nonzero(got.read_type != rt)[0][0])
        raise AssertionError("%s: type differs at read %d: got %d want %d" % (
            ctx, r, int(got.read_type[r]), int(rt[r])))
