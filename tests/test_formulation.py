"""Fuzz the GPU formulations (emulated sequentially in Python) against the oracle.  CPU only."""
import numpy as np
import pytest

import oracle
from cases import make_read
from formulation import general_events, regular_events

MODES = ("regular", "abutting", "dups", "beyond", "sparse", "degenerate", "huge_pos", "zero_len")


@pytest.mark.parametrize("mode", MODES)
def test_formulations_match_oracle(mode):
    rng = np.random.default_rng(hash(mode) % 2**31)
    n_reg = 0
    for it in range(400):
        n = int(rng.integers(0, 40)) if it % 4 else int(rng.integers(40, 300))
        L = int(rng.integers(1, 3000)) if it % 3 else int(rng.integers(1, 60))
        iv = [tuple(int(x) for x in p) for p in make_read(rng, n, L, mode)]
        for cov in (0, 1, 2, 4):
            want = oracle.compute_bad_part(iv, L, cov)
            assert general_events(iv, L, cov) == want, (mode, iv, L, cov)
            reg = regular_events(iv, L, cov)
            if reg is not None:
                n_reg += 1
                assert reg == want, (mode, iv, L, cov)
    if mode not in ("huge_pos",):
        assert n_reg > 0


def test_tiny_exhaustive():
    """All multisets of <= 3 intervals over positions 0..4, len 0..4, c 0..2, degenerate included."""
    import itertools
    pairs = [(s, e) for s in range(5) for e in range(5)]
    for k in range(0, 4):
        for iv in itertools.combinations_with_replacement(pairs, k):
            for L in range(0, 5):
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    assert general_events(list(iv), L, cov) == want, (iv, L, cov)
                    reg = regular_events(list(iv), L, cov)
                    if reg is not None:
                        assert reg == want, (iv, L, cov)
