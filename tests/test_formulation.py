"""Fuzz the GPU formulations (emulated sequentially in Python) against the oracle.  CPU only."""
import numpy as np
import pytest

import oracle
from cases import make_read
from formulation import general_events, prefilter_keys, prefiltered_events, regular_events

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
            for nb in (4, 16, 32):  # may accept reads whose duplicate zero-length pair sits in a safe bin
                pre = prefiltered_events(iv, L, cov, nb)
                assert pre is None or pre == want, (mode, iv, L, cov, nb)
                assert reg is None or pre is not None
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
                    pre = prefiltered_events(list(iv), L, cov, 4)
                    assert pre is None or pre == want, (iv, L, cov)
                    assert reg is None or pre is not None


def test_prefilter_deep_pileups():
    """Deep pile-ups at small c: most bins are safe, the filter drops most events and the result
    must not change (modes with ties, zero-length intervals and ends beyond the read included)."""
    rng = np.random.default_rng(77)
    dropped = total = 0
    for it in range(600):
        n = int(rng.integers(20, 260))
        L = int(rng.integers(16, 40000)) if it % 5 else int(rng.integers(1, 64))
        mode = ("regular", "abutting", "dups", "beyond", "zero_len")[it % 5]
        iv = [tuple(int(x) for x in p) for p in make_read(rng, n, L, mode)]
        if it % 7 == 0:  # positions on a coarse grid: many ties on bin boundaries
            g = max(1, L // 16)
            iv = [((s // g) * g, max((e // g) * g, (s // g) * g + (1 if e > s else 0))) for s, e in iv]
        for cov in (0, 1, 4, 9):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb in (8, 16, 32):
                got = prefiltered_events(iv, L, cov, nb)
                if got is not None:
                    assert got == want, (mode, iv, L, cov, nb)
            if regular_events(iv, L, cov) is not None:
                total += 2 * len(iv)
                dropped += 2 * len(iv) - len(prefilter_keys(iv, L, cov, 16))
    assert dropped > total // 4  # the filter really fires in this test


def _pile_read(rng, n, L, jitter):
    """Dovetail-style read: piles of starts near 0 and of ends near L, some internal intervals."""
    iv = []
    for _ in range(n):
        u = rng.random()
        ell = int(rng.integers(1, max(2, L)))
        if u < 0.35:
            s = max(0, int(round(jitter * rng.normal())))
            e = s + ell
        elif u < 0.7:
            e = L + int(round(jitter * rng.normal()))
            s = e - ell
        else:
            s = int(rng.integers(0, max(1, L)))
            e = s + ell
        s = min(max(s, 0), L - 1)
        e = min(max(e, s + 1), L)
        iv.append((s, e))
    return iv


def test_pile_trimming_matches_oracle():
    from formulation import healthy_read_regions, trim_keys, trimmed_events, trimmed_minmax_events
    rng = np.random.default_rng(2024)
    kept = total = n_zl_checked = n_healthy = 0
    for it in range(1500):
        L = int(rng.integers(2, 200)) if it % 3 == 0 else int(rng.integers(200, 50000))
        n = int(rng.integers(1, 30)) if it % 4 == 0 else int(rng.integers(30, 400))
        jitter = (0.0, 1.0, 5.0, 30.0)[it % 4]
        iv = _pile_read(rng, n, L, jitter)
        if it % 5 == 1:  # some zero-length intervals, in the piles and elsewhere
            for _ in range(int(rng.integers(1, 4))):
                j = int(rng.integers(0, len(iv)))
                p0 = (iv[j][0], iv[j][1], 0, L, L // 2)[int(rng.integers(0, 5))]
                iv[j] = (p0, p0)
        if it % 11 == 0:  # everything on a coarse grid: ties everywhere
            g = max(1, L // 8)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        has_zl = any(s == e for s, e in iv)
        for cov in (0, 1, 4, 9, 50):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, F in ((16, 32), (16, 0), (4, 3), (256, 128), (16, 64), (64, 32)):  # (64, 32): finish_compact.h trim_item
                got = trimmed_events(iv, L, cov, nb, F)
                assert got == want or (got is None and has_zl), (iv, L, cov, nb, F)
                n_zl_checked += has_zl and got is not None
                gw = trimmed_events(iv, L, cov, nb, F, wave=True)  # coarse bins aligned at 0
                assert gw == want or (gw is None and has_zl), (iv, L, cov, nb, F, "wave")
                if not has_zl:  # sweep_wave.h's deferring build: bins at the smallest start / largest end
                    assert trimmed_minmax_events(iv, L, cov, nb) == want, (iv, L, cov, nb, "minmax")
                    hr = healthy_read_regions(iv, L, cov, nb)  # its closed form, where it applies
                    assert hr is None or hr == want, (iv, L, cov, nb, "healthy")
                    n_healthy += hr is not None
                    if it % 7 == 3:  # a read covered only inside a window: piles at the window's edges
                        w0, w1 = L // 3, max(L // 3 + 2, 2 * L // 3)
                        win = [(min(max(s, w0), w1 - 1), min(max(e, min(max(s, w0), w1 - 1) + 1), w1)) for s, e in iv]
                        assert trimmed_minmax_events(win, L, cov, nb) == oracle.compute_bad_part(win, L, cov), (win, L, cov, nb)
            if cov == 4:
                total += 2 * len(iv)
                kept += len(trim_keys(iv, L, cov, 16, 32))
    assert kept < total // 3 and n_zl_checked > 500 and n_healthy > 500


def test_pile_trimming_tiny_exhaustive():
    import itertools
    from formulation import healthy_read_regions, trimmed_events, trimmed_minmax_events
    for L in range(1, 6):
        pairs = [(s, e) for s in range(L + 1) for e in range(s, L + 1)]  # zero-length ones included
        for k in range(1, 4):
            for iv in itertools.combinations_with_replacement(pairs, k):
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, F in ((2, 1), (2, 2), (4, 0), (4, 2)):
                        got = trimmed_events(list(iv), L, cov, nb, F)
                        assert got is None or got == want, (iv, L, cov, nb, F)
                        gw = trimmed_events(list(iv), L, cov, nb, min(F, 1), wave=True)
                        assert gw is None or gw == want, (iv, L, cov, nb, F, "wave")
                        if all(s < e for s, e in iv):
                            assert trimmed_minmax_events(list(iv), L, cov, nb) == want, (iv, L, cov, nb, "minmax")
                            hr = healthy_read_regions(list(iv), L, cov, nb)
                            assert hr is None or hr == want, (iv, L, cov, nb, "healthy")
                        assert got is not None or regular_events(list(iv), L, cov) is None


def test_window_screen_matches_oracle():
    """The order-statistics screen of sweep_wave.h (round 3): wherever it applies it equals the oracle —
    piles on one exact position (jitter 0), spread piles, windows, ties on a coarse grid, every window
    size the kernel can be built with and tiny ones that make the windows collide with everything."""
    from formulation import window_screen_regions
    rng = np.random.default_rng(31337)
    n_fired = n_total = 0
    for it in range(1500):
        L = int(rng.integers(2, 300)) if it % 3 == 0 else int(rng.integers(300, 50000))
        n = int(rng.integers(1, 30)) if it % 4 == 0 else int(rng.integers(30, 300))
        jitter = (0.0, 1.0, 5.0, 30.0, 100.0)[it % 5]
        iv = _pile_read(rng, n, L, jitter)
        if it % 6 == 1:  # long intervals only: the screen's usual case
            iv = [(s, min(max(e, s + 70), L)) for s, e in iv if s + 70 <= L] or iv
        if it % 11 == 0:  # everything on a coarse grid: ties everywhere
            g = max(1, L // 8)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        if it % 7 == 3:  # covered only inside a window: the order statistics sit at the window's edges
            w0, w1 = L // 3, max(L // 3 + 2, 2 * L // 3)
            iv = [(min(max(s, w0), w1 - 1), min(max(e, min(max(s, w0), w1 - 1) + 1), w1)) for s, e in iv]
        for cov in (0, 1, 4, 9, 50, 400):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W in ((16, 32), (32, 32), (16, 64), (16, 1), (4, 2), (8, 5)):
                got = window_screen_regions(iv, L, cov, nb, W)
                n_total += 1
                if got is not None:
                    n_fired += 1
                    assert got == want, (iv, L, cov, nb, W)
    assert n_fired > n_total // 5


def test_slid_window_screen_matches_oracle():
    """Round 4: the same screen with windows that slide on by W when the order statistic lies beyond them
    (dovetail ends spread by sigma = 100 .. 300 positions): wherever it decides — after any number of slides — it
    equals the oracle; what it decides without a slide it decides the same way with slides allowed; and it decides
    MORE reads."""
    from formulation import slid_window_screen_regions, window_screen_regions
    rng = np.random.default_rng(777)
    fired = {0: 0, 4: 0}
    slid = 0
    spotted = {2: 0, 8: 0, 1 << 20: 0}
    mirrored = {False: 0, True: 0}
    jumped = [0, 0, 0, 0]
    for it in range(1500):
        L = int(rng.integers(2, 400)) if it % 3 == 0 else int(rng.integers(400, 50000))
        n = int(rng.integers(1, 30)) if it % 4 == 0 else int(rng.integers(30, 300))
        jitter = (0.0, 5.0, 30.0, 100.0, 300.0)[it % 5]
        iv = _pile_read(rng, n, L, jitter)
        if it % 6 == 1:
            iv = [(s, min(max(e, s + 200), L)) for s, e in iv if s + 200 <= L] or iv
        if it % 11 == 0:
            g = max(1, L // 8)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        if it % 7 == 3:
            w0, w1 = L // 3, max(L // 3 + 2, 2 * L // 3)
            iv = [(min(max(s, w0), w1 - 1), min(max(e, min(max(s, w0), w1 - 1) + 1), w1)) for s, e in iv]
        for cov in (0, 1, 4, 9, 50):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W in ((16, 32), (32, 32), (16, 8), (4, 2)):
                base = window_screen_regions(iv, L, cov, nb, W)
                # (the coarse blocks are counted from the window here and from position 0 there: now and then one of
                # the two partitions decides a read the other leaves to the sort; both equal the oracle where they decide)
                zero = slid_window_screen_regions(iv, L, cov, nb, W, 0)
                assert zero is None or (zero[0] == want and zero[1] == 0), (iv, L, cov, nb, W)
                got = slid_window_screen_regions(iv, L, cov, nb, W, 4)
                if base is not None:
                    assert base == want
                    fired[0] += 1
                if zero is not None:
                    assert got is not None and got == zero
                if got is not None:
                    assert got[0] == want, (iv, L, cov, nb, W, got)
                    fired[4] += 1
                    slid += got[1] > 0
                ramp = slid_window_screen_regions(iv, L, cov, nb, W, 4, ramp_always=True)
                assert ramp is None or ramp[0] == want, (iv, L, cov, nb, W, ramp)
                if got is not None:  # (what the plain form decides the ramp form decides too: it only adds open intervals)
                    assert ramp is not None
                # round 6: SPOT CHECKS of the coarse-counted starts of the blocks that fail the depth test (up to 2 / 8 / any)
                for sp in (2, 8, 1 << 20):
                    for ra in (False, True):
                        spot = slid_window_screen_regions(iv, L, cov, nb, W, 4, ramp_always=ra, spot=sp)
                        assert spot is None or spot[0] == want, (iv, L, cov, nb, W, sp, ra, spot)
                        if (ramp if ra else got) is not None:
                            assert spot is not None
                        spotted[sp] += spot is not None and (ramp if ra else got) is None
                # round 6: the ramp's mirror — ends behind the largest start are in no block's count
                for ra in (False, True):
                    tr = slid_window_screen_regions(iv, L, cov, nb, W, 4, ramp_always=ra, tail_ramp=True)
                    assert tr is None or tr[0] == want, (iv, L, cov, nb, W, ra, tr)
                    if (ramp if ra else got) is not None:
                        assert tr is not None
                    mirrored[ra] += tr is not None and (ramp if ra else got) is None
                    # ... and windows that JUMP to the next event instead of sliding by W
                    jp = slid_window_screen_regions(iv, L, cov, nb, W, 4, ramp_always=ra, tail_ramp=True, jump=True)
                    assert jp is None or jp[0] == want, (iv, L, cov, nb, W, ra, "jump", jp)
                    jumped[0] += jp is not None
                    jumped[1] += tr is not None
                    if jp is not None and tr is not None:
                        jumped[2] += jp[1]
                        jumped[3] += tr[1]
    assert jumped[0] >= jumped[1] and jumped[2] < jumped[3], jumped  # (decides no fewer reads, in fewer passes)
    assert fired[4] > fired[0] and slid > 100, (fired, slid)
    assert spotted[8] > 200 and spotted[1 << 20] >= spotted[8] >= spotted[2], spotted
    assert mirrored[True] > 50, mirrored


def test_slid_window_screen_tiny_exhaustive():
    import itertools
    from formulation import slid_window_screen_regions
    for L in range(1, 8):
        pairs = [(s, e) for s in range(L + 1) for e in range(s + 1, L + 1)]
        for k in range(1, 4):
            for iv in itertools.combinations_with_replacement(pairs, k):
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W in ((2, 1), (4, 1), (4, 2)):
                        for ramp in (False, True):
                            for sp in (0, 3):
                                got = slid_window_screen_regions(list(iv), L, cov, nb, W, 3, ramp_always=ramp, spot=sp)
                                assert got is None or got[0] == want, (iv, L, cov, nb, W, sp, got)
                            got = slid_window_screen_regions(list(iv), L, cov, nb, W, 3, ramp_always=ramp, tail_ramp=True)
                            assert got is None or got[0] == want, (iv, L, cov, nb, W, "tail ramp", got)
                            got = slid_window_screen_regions(list(iv), L, cov, nb, W, 3, ramp_always=ramp, tail_ramp=True, jump=True)
                            assert got is None or got[0] == want, (iv, L, cov, nb, W, "jump", got)


def test_window_screen_tiny_exhaustive():
    import itertools
    from formulation import window_screen_regions
    for L in range(1, 7):
        pairs = [(s, e) for s in range(L + 1) for e in range(s + 1, L + 1)]
        for k in range(1, 4):
            for iv in itertools.combinations_with_replacement(pairs, k):
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W in ((2, 1), (2, 2), (4, 1), (4, 3)):
                        got = window_screen_regions(list(iv), L, cov, nb, W)
                        assert got is None or got == want, (iv, L, cov, nb, W)


def test_unified_screen_matches_oracle():
    """The workgroup / device-wide screen (one position map for starts and ends): wherever it decides, it
    equals the oracle — short intervals anywhere, starts inside the tail window, ends inside the head
    window, ties, windows, tiny W."""
    from formulation import unified_screen_regions
    rng = np.random.default_rng(4242)
    n_fired = n_total = 0
    for it in range(1500):
        L = int(rng.integers(2, 600)) if it % 3 == 0 else int(rng.integers(600, 50000))
        n = int(rng.integers(1, 30)) if it % 4 == 0 else int(rng.integers(30, 400))
        jitter = (0.0, 1.0, 5.0, 30.0, 100.0)[it % 5]
        iv = _pile_read(rng, n, L, jitter)
        if it % 6 == 1:  # a few one-position intervals at the very ends and in the middle
            iv += [(L - 1, L), (0, 1), (L // 2, L // 2 + 1)][: int(rng.integers(1, 4))]
        if it % 5 == 2:  # zero-length intervals: in the piles, in the middle, doubled, at 0 and at len
            for _ in range(int(rng.integers(1, 5))):
                j = int(rng.integers(0, len(iv)))
                p0 = (iv[j][0], iv[j][1], 0, 0, L, L // 2, L // 2, min(s for s, e in iv), min(s for s, e in iv))[int(rng.integers(0, 9))]
                iv.append((p0, p0))
        if it % 11 == 0:
            g = max(1, L // 8)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        if it % 7 == 3:
            w0, w1 = L // 3, max(L // 3 + 2, 2 * L // 3)
            iv = [(min(max(s, w0), w1 - 1), min(max(e, min(max(s, w0), w1 - 1) + 1), w1)) for s, e in iv]
        for cov in (0, 1, 4, 9, 50, 400):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W in ((256, 128), (16, 32), (64, 4), (4, 1), (8, 5), (1024, 512)):
                got = unified_screen_regions(iv, L, cov, nb, W)
                n_total += 1
                if got is not None:
                    n_fired += 1
                    assert got == want, (iv, L, cov, nb, W)
    assert n_fired > n_total // 6


def test_unified_screen_tiny_exhaustive():
    import itertools
    from formulation import unified_screen_regions
    for L in range(1, 8):
        pairs = [(s, e) for s in range(L + 1) for e in range(s, L + 1)]  # zero-length ones included
        for k in range(1, 4):
            for iv in itertools.combinations_with_replacement(pairs, k):
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W in ((2, 1), (4, 1), (4, 2), (8, 3)):
                        got = unified_screen_regions(list(iv), L, cov, nb, W)
                        assert got is None or got == want, (iv, L, cov, nb, W)


def _survey_read(rng, n, L, jitter):
    """SURVEY.md 8d's shape: 60 % dovetails anchored within N(0, jitter) of one end (reflected into the read), 40 %
    internal, every interval at least min(500, L / 4) long."""
    lo = max(1, min(500, L // 4))
    iv = []
    for _ in range(n):
        if rng.random() < 0.6:
            ell = lo + int(rng.integers(0, max(1, int(0.8 * L) - lo)))
            j = abs(int(round(jitter * rng.normal())))
            if rng.random() < 0.5:
                s, e = j, j + ell
            else:
                e = L - j
                s = e - ell
        else:
            ell = lo + int(rng.integers(0, max(1, L // 2 - lo)))
            s = int(rng.integers(0, max(1, L - ell)))
            e = s + ell
        s = min(max(s, 0), L - 1)
        e = min(max(e, s + 1), L)
        iv.append((s, e))
    return iv


def _chimera_read(rng, n, L, jitter, base=None):
    """SURVEY.md 8d's chimera: a junction j; every interval that crosses it is cut back to the side holding its
    midpoint, 10..100 positions short of j — here also with a few intervals left spanning, gaps of any width and
    intervals ending / starting exactly at the junction."""
    iv = (base or _pile_read)(rng, n, L, jitter)
    j = int(rng.integers(max(1, L // 5), max(2, 4 * L // 5)))
    span_left = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0
    out = []
    for s, e in iv:
        if s < j < e:
            if span_left > 0 and rng.random() < 0.1:
                span_left -= 1
                out.append((s, e))
                continue
            gap = int(rng.integers(0, 100)) if rng.random() < 0.8 else 0
            if (s + e) // 2 < j:
                e = max(s + 1, j - gap)
            else:
                s = min(e - 1, j + gap)
        out.append((s, e))
    return out


def test_hole_screen_matches_oracle():
    """Round 4: the closed form for a read with ONE stretch of low coverage inside (formulation.hole_screen_regions):
    wherever it decides it equals the oracle — chimeras with gaps of any width, intervals left spanning the junction,
    spread piles, coarse grids — and it decides most of the generator's chimeras."""
    from formulation import hole_fast_regions, hole_screen_regions, slid_window_screen_regions
    rng = np.random.default_rng(90210)
    fired = total = decided = decided_fast = 0
    for it in range(2500):
        L = int(rng.integers(50, 600)) if it % 5 == 0 else int(rng.integers(600, 60000))
        n = int(rng.integers(2, 40)) if it % 4 == 0 else int(rng.integers(40, 256))
        jitter = (0.0, 5.0, 30.0, 100.0)[it % 4]
        base = _survey_read if it % 2 else _pile_read
        iv = _chimera_read(rng, n, L, jitter, base) if it % 3 else base(rng, n, L, jitter)
        if it % 13 == 0:
            g = max(1, L // 16)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        for cov in (0, 1, 3, 4, 9):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W in ((16, 32), (32, 32), (16, 8), (4, 4)):
                got = hole_screen_regions(iv, L, cov, nb, W, 4)
                assert got is None or got == want, (iv, L, cov, nb, W, got, want)
                decided += got is not None
                fast = hole_fast_regions(iv, L, cov, nb, W)
                assert fast is None or fast == want, (iv, L, cov, nb, W, fast, want)
                decided_fast += fast is not None
                if it % 3 and it % 2 and nb >= 16 and W == 32 and cov in (3, 4) and L >= 4000 and n >= 80 and it % 13:
                    total += 1
                    fired += got is not None or slid_window_screen_regions(iv, L, cov, nb, W, 4, True) is not None
    assert fired > total * 8 // 10 and decided > 3000 and decided_fast > 1500, (fired, total, decided, decided_fast)


def test_hole_screen_tiny_exhaustive():
    import itertools
    from formulation import hole_fast_regions, hole_screen_regions
    for L in range(2, 9):
        pairs = [(s, e) for s in range(L + 1) for e in range(s + 1, L + 1)]
        for k in range(2, 5):
            for iv in itertools.combinations_with_replacement(pairs, k):
                if k == 4 and (sum(a for a, b in iv) + L) % 3:
                    continue  # (a third of the quadruples)
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W, nbf in ((4, 1, 8), (2, 1, 4), (4, 2, 64)):
                        got = hole_screen_regions(list(iv), L, cov, nb, W, 2, nbf)
                        assert got is None or got == want, (iv, L, cov, nb, W, got, want)
                    for nb, W, nbf in ((4, 1, 8), (2, 1, 4), (2, 2, 8)):
                        got = hole_fast_regions(list(iv), L, cov, nb, W, nbf)
                        assert got is None or got == want, (iv, L, cov, nb, W, nbf, got, want)


def test_filtered_sweep_matches_oracle():
    """Round 5: the follow-on step's filtered exact sweep (formulation.filtered_sweep_regions): wherever it decides it
    equals the oracle — healthy reads, chimeras with any gap, several holes, spread piles, coarse grids — and it
    decides nearly all of the generator's chimeras within its key budget."""
    from formulation import filtered_sweep_regions
    rng = np.random.default_rng(5150)
    fired = total = decided = 0
    for it in range(2500):
        L = int(rng.integers(50, 600)) if it % 5 == 0 else int(rng.integers(600, 60000))
        n = int(rng.integers(2, 40)) if it % 4 == 0 else int(rng.integers(40, 256))
        jitter = (0.0, 5.0, 30.0, 100.0)[it % 4]
        base = _survey_read if it % 2 else _pile_read
        iv = _chimera_read(rng, n, L, jitter, base) if it % 3 else base(rng, n, L, jitter)
        if it % 7 == 0:  # a second junction
            iv = _chimera_read(rng, n, L, jitter, lambda *_: iv)
        if it % 13 == 0:
            g = max(1, L // 16)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        for cov in (0, 1, 3, 4, 9):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W, cap in ((16, 32, 64), (32, 32, 128), (16, 8, 64), (4, 4, 16), (32, 32, 1 << 20)):
                got = filtered_sweep_regions(iv, L, cov, nb, W, cap)
                assert got is None or got == want, (iv, L, cov, nb, W, cap, got, want)
                decided += got is not None
                if it % 3 and it % 2 and nb == 32 and cap == 128 and cov in (3, 4) and L >= 4000 and n >= 80 and it % 13:
                    total += 1
                    fired += got is not None
    assert fired > total * 8 // 10 and decided > 10000, (fired, total, decided)


def test_filtered_sweep_tiny_exhaustive():
    import itertools
    from formulation import filtered_sweep_regions
    for L in range(2, 9):
        pairs = [(s, e) for s in range(L + 1) for e in range(s + 1, L + 1)]
        for k in range(2, 5):
            for iv in itertools.combinations_with_replacement(pairs, k):
                if k == 4 and (sum(a for a, b in iv) + L) % 3:
                    continue  # (a third of the quadruples)
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W in ((4, 1), (2, 1), (4, 2), (8, 1)):
                        got = filtered_sweep_regions(list(iv), L, cov, nb, W, 1 << 20)
                        assert got is None or got == want, (iv, L, cov, nb, W, got, want)


def _short_mix(rng, iv, L, k):
    """k short intervals (1 .. 40 positions) anywhere: what a read of thousands of intervals always has."""
    out = list(iv)
    for _ in range(k):
        s = int(rng.integers(0, max(1, L - 1)))
        out.append((s, min(L, s + int(rng.integers(1, 41)))))
    return out


def test_unified_filtered_matches_oracle():
    """Round 6: the workgroup classes' filtered exact sweep (formulation.unified_filtered_regions; screen_wg.h): wherever it
    decides it equals the oracle (src/stack.rs:61-139) — healthy reads, chimeras with any gap, several holes, short intervals
    anywhere (starts inside the tail window, ends inside the head window), zero-length intervals, spread piles, coarse grids —
    and it decides nearly all of the generator's chimeras within its key budget."""
    from formulation import unified_filtered_regions, unified_screen_regions
    rng = np.random.default_rng(6160)
    fired = total = decided = beyond_screen = 0
    for it in range(2500):
        L = int(rng.integers(50, 600)) if it % 5 == 0 else int(rng.integers(600, 60000))
        n = int(rng.integers(2, 40)) if it % 4 == 0 else int(rng.integers(40, 320))
        jitter = (0.0, 5.0, 30.0, 100.0)[it % 4]
        base = _survey_read if it % 2 else _pile_read
        iv = _chimera_read(rng, n, L, jitter, base) if it % 3 else base(rng, n, L, jitter)
        if it % 7 == 0:  # a second junction
            iv = _chimera_read(rng, n, L, jitter, lambda *_: iv)
        if it % 4 == 1:
            iv = _short_mix(rng, iv, L, int(rng.integers(1, 12)))
        if it % 9 == 2:  # zero-length intervals: in the piles, in the middle, doubled, at 0 and at len
            for _ in range(int(rng.integers(1, 4))):
                j = int(rng.integers(0, len(iv)))
                p0 = (iv[j][0], iv[j][1], 0, 0, L, L // 2, L // 2, min(s for s, e in iv), min(s for s, e in iv))[int(rng.integers(0, 9))]
                iv.append((p0, p0))
        if it % 13 == 0:
            g = max(1, L // 16)
            iv = [(min((s // g) * g, L - 1), min(max((e // g) * g, (s // g) * g + 1), L)) for s, e in iv]
            iv = [(s, max(e, s + 1)) for s, e in iv]
        for cov in (0, 1, 3, 4, 9):
            want = oracle.compute_bad_part(iv, L, cov)
            for nb, W, cap in ((256, 128, 512), (16, 32, 64), (32, 32, 128), (16, 8, 64), (4, 4, 16), (8, 5, 1 << 20), (32, 32, 1 << 20)):
                got = unified_filtered_regions(iv, L, cov, nb, W, cap)
                assert got is None or got == want, (iv, L, cov, nb, W, cap, got, want)
                decided += got is not None
                beyond_screen += got is not None and unified_screen_regions(iv, L, cov, nb, W) is None
                if it % 3 and it % 2 and nb == 32 and cap == 128 and cov in (3, 4) and L >= 4000 and n >= 80 and it % 13:
                    total += 1
                    fired += got is not None
    assert fired > total * 7 // 10 and decided > 10000 and beyond_screen > 3000, (fired, total, decided, beyond_screen)


def test_unified_filtered_tiny_exhaustive():
    import itertools
    from formulation import unified_filtered_regions
    for L in range(2, 9):
        pairs = [(s, e) for s in range(L + 1) for e in range(s, L + 1)]  # zero-length ones included
        for k in range(2, 5):
            for iv in itertools.combinations_with_replacement(pairs, k):
                if k == 4 and (sum(a for a, b in iv) + L) % 4:
                    continue  # (a quarter of the quadruples)
                for cov in range(0, 3):
                    want = oracle.compute_bad_part(list(iv), L, cov)
                    for nb, W in ((4, 1), (2, 1), (4, 2), (8, 1), (8, 3)):
                        got = unified_filtered_regions(list(iv), L, cov, nb, W, 1 << 20)
                        assert got is None or got == want, (iv, L, cov, nb, W, got, want)
