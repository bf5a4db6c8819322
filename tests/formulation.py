"""Sequential Python emulation of the two GPU formulations, line for line what
yacrd_amd/csrc/sweep_lds.h and sweep_general.h compute, minus the parallel scans.  Lets the
formulations be fuzzed against the oracle on CPU (tests/test_formulation.py)."""

NO = 0xFFFFFFFF


SH = 2  # key = pos << 2 | class: 0 end, 1 zero-length start, 2 zero-length end, 3 start


def finish_read(slot, mf_t, ml_t, min_ge, length):
    lcf = min_ge if min_ge != NO else (mf_t >> SH)
    if ml_t > mf_t:
        b, e = mf_t >> SH, ml_t >> SH
        if mf_t == 0:
            if e != 0 and length != 0:
                slot.append((0, max(e, length)))
            elif e != 0:
                slot.append((0, e))
            elif length != 0:
                slot.append((0, length))
        else:
            slot.append((b, max(e, length) if lcf != length else e))
    elif lcf != length:
        slot.append((lcf, length))
    return slot


def regular_keys(intervals):
    """Sorted event keys of a regular read; None if the read must take the exact general path."""
    keys = []
    for s, e in intervals:
        if s > e or e >= 2**30 - 1:
            return None
        if s == e:
            keys.append((s << SH) | 1)
            keys.append((s << SH) | 2)
        else:
            keys.append((s << SH) | 3)
            keys.append(e << SH)
    keys.sort()
    for a, b in zip(keys, keys[1:]):
        if a == b and (a & 3) == 1 and a != 1:
            return None  # duplicate zero-length interval: S,S,E,E would not be S,E,S,E
        # (0,0) intervals are inert in the reference (their pop re-assigns last_covered = 0 while it
        # still is 0) and in the keys (flipped end key 0 never beats "none" = 1): no exception needed
    return keys


def sweep_keys(keys, length, cov):
    """The post-sort sweep over sorted event keys (sweep_wave.h passes 1-3 + finish_read)."""
    # Flagged ends are tracked in a "flipped" domain tk = key ^ 2 (class 0 <-> 2): at one position
    # a regular end then beats a zero-length interval's own end under max, i.e. the FIRST flagged
    # end of a position is the effective one (later ones leave last_covered unchanged, so an open
    # run of low starts at that position must stay open).  tc = 1 means "none": its true key 1 ^ 2
    # = 3 is the key of a start at position 0, which makes `ml > true(tc)` the reference's
    # `first_covered != 0` test.
    n_starts = sum(1 for k in keys if k & 1)
    d = 0
    tc = 1
    ml = 0
    min_ge = NO
    slot = []
    any_flag = False
    seen = 0
    for key in keys:
        if key & 1:
            if d <= cov:
                ml = key
            d += 1
            seen += 1
        else:
            if d > cov:
                tk = key ^ 2
                if tk > tc:
                    if ml > (tc ^ 2):
                        slot.append(((tc ^ 2) >> SH, ml >> SH))
                    tc = tk
                    any_flag = True
                if seen == n_starts and (key >> SH) >= length:  # tail: every start precedes it
                    min_ge = min(min_ge, key >> SH)
            d -= 1
    return finish_read(slot, (tc ^ 2) if any_flag else 0, ml, min_ge, length)


def regular_events(intervals, length, cov):
    """Event formulation for regular reads (start <= end < 2^30 - 1, no two zero-length
    intervals at one position); None if the read must take the exact general path."""
    if len(intervals) == 0:
        return [(0, length)] if length != 0 else []
    keys = regular_keys(intervals)
    return None if keys is None else sweep_keys(keys, length, cov)


def bin_shift(length, nb):
    """Smallest shift with (length >> shift) < nb: bins of 2^shift positions, the bin that holds
    `length` (and every later one) exists inside the table."""
    sh = 0
    while (length >> sh) >= nb:
        sh += 1
    return sh


def prefilter_keys(intervals, length, cov, nb):
    """Coverage pre-filter (sweep_wave.h `prefilter`): drop every event inside a *safe* bin — one
    that more than `cov` intervals span completely — and stand in for each maximal run of safe
    bins with |net| start (net > 0) or end (net < 0) keys at the run's first position, so that the
    depth of every surviving event is unchanged.  Returns the (unsorted) surviving keys."""
    sh = bin_shift(length, nb)
    S, E = [0] * nb, [0] * nb
    keys = []
    for s, e in intervals:
        S[min(s >> sh, nb - 1)] += 1
        E[min(e >> sh, nb - 1)] += 1
        keys += [(s << SH) | 1, (s << SH) | 2] if s == e else [(s << SH) | 3, e << SH]
    first_unsafe = length >> sh           # bins holding positions >= length are never safe
    safe, depth_at, cs, ce = [], [], 0, 0
    for b in range(nb):
        depth_at.append(cs - ce)          # depth at the bin's first position
        ce += E[b]
        safe.append(cs - ce > cov and b < first_unsafe)  # <= #intervals spanning the whole bin
        cs += S[b]
    depth_at.append(cs - ce)
    out = [k for k in keys if not safe[min((k >> SH) >> sh, nb - 1)]]
    b = 0
    while b < nb:
        if safe[b]:
            b2 = b
            while b2 + 1 < nb and safe[b2 + 1]:
                b2 += 1
            net = depth_at[b2 + 1] - depth_at[b]
            pos = b << sh
            out += [(pos << SH) | 3] * net if net > 0 else [pos << SH] * (-net)
            b = b2 + 1
        else:
            b += 1
    return out


def prefiltered_events(intervals, length, cov, nb=16):
    """regular_events with the coverage pre-filter in front of the sort.  As on the GPU, duplicate
    zero-length intervals are only looked for among the keys that survive the filter."""
    if len(intervals) == 0:
        return [(0, length)] if length != 0 else []
    if any(s > e or e >= 2**30 - 1 for s, e in intervals):
        return None
    keys = sorted(prefilter_keys(intervals, length, cov, nb))
    for a, b in zip(keys, keys[1:]):
        if a == b and (a & 3) == 1 and a != 1:
            return None
    return sweep_keys(keys, length, cov)


def general_events(intervals, length, cov):
    """Exact formulation for arbitrary u32 input."""
    n = len(intervals)
    if n == 0:
        return [(0, length)] if length != 0 else []
    K = sorted((s << 32) | e for s, e in intervals)
    starts = [k >> 32 for k in K]
    import bisect
    EV = []
    for j, k in enumerate(K):
        s, e = k >> 32, k & NO
        t = bisect.bisect_left(starts, e) if e > s else j + 1
        EV.append(((2 * j + 1) << 32) | s)
        EV.append(((2 * t) << 32) | e)
    EV.sort()
    M = 2 * n
    d = 0
    lc = 0
    lf_val = None
    brk = None
    fc = 0
    raw = []
    for q, ev in enumerate(EV):
        hi, val = ev >> 32, ev & NO
        if hi & 1:
            if d <= cov:
                if lc != 0:
                    raw.append((lc, val))
                else:
                    fc = val
            d += 1
        else:
            if d > cov:
                lc = val
                lf_val = val
                if hi == M and val >= length and brk is None:
                    brk = val
            d -= 1
    lcf = brk if brk is not None else (lf_val if lf_val is not None else 0)
    lst = ([(0, fc)] if fc != 0 else []) + raw + ([(lcf, length)] if lcf != length else [])
    out = []
    for t, (b, e) in enumerate(lst):
        if t + 1 == len(lst) or lst[t + 1][0] != b:
            if t > 0 and lst[t - 1][0] == b:
                e = max(e, lst[t - 1][1])
            out.append((b, e))
    return out


# --------------------------------------------------------------------------------------------------
# Pile trimming: the pre-filter with position-exact bins at both ends of the read.
#
# An event is DEEP when the depth is above `cov` on both sides of it (start: depth_before >= cov+1;
# end: depth_before >= cov+2).  Deep starts are never low; a deep flagged end is always followed by
# another flagged end before the next low start and before the end of the read (the end that takes
# the depth from cov+1 to cov is not deep), so it never supplies a region's begin, and if it was
# the first flagged end after a run of low starts the next kept flagged end closes the same region.
# Any set of deep events may therefore be dropped as long as the depth of every kept event is
# unchanged: each maximal run of dropped events (contiguous in key order) is stood in for by |net|
# keys of one type placed between the kept events on either side of it.
# Which events are deep is known exactly, without sorting, in two kinds of bins:
#   * a coarse bin that more than `cov` intervals span completely: all of its events (the old rule);
#   * a bin holding ONE position p: its keys are E ends followed by S starts, all equal within a
#     type, so with the depth D at the bin's head the first max(0, D - cov - 1) ends and all but the
#     first max(0, cov + 1 - (D - E)) starts are deep.  Dovetail overlaps pile hundreds of starts
#     within a few dozen positions of 0 and as many ends near `length`: the first F and the last F
#     positions get such bins, and of a pile only its cov + 1 outermost events survive.
# Only for plain reads (every interval 0 <= start < end <= length).

def trim_keys(intervals, length, cov, nb, F, wave=False):
    """wave=True: the bin geometry of sweep_wave.h (coarse bins aligned at 0, fine zones only when
    length >= nb * F, sequence index = min(pos, F) + (pos >> sh) + max(pos - (length - F), 0)).
    Zero-length intervals (start == end) are taken too: in a coarse bin they count as a start and
    an end of that bin (kept or dropped with it); at a one-position bin their two keys sit between
    the bin's regular ends and its regular starts, at depth D - E, so the pair is deep — dropped —
    exactly when the bin's first regular start would be (D - E >= cov + 1), and kept otherwise."""
    if wave:
        if length < nb * F:
            F = 0
        sh = bin_shift(length, nb)
        H = length - F
        nbins = F + nb + F
        idx_h = F + (H >> sh)

        def bin_of(pos):
            return min(pos, F) + (pos >> sh) + max(pos - H, 0)

        fp_table = {}
        for pos in range(0, min(F, length + 1)):
            fp_table.setdefault(bin_of(pos), pos)
        for pos in range(H + 1, length + 1):
            fp_table.setdefault(bin_of(pos), pos)

        def first_pos(b):
            if b in fp_table:
                return fp_table[b]
            return max((b - F) << sh, F)      # a coarse bin

        uniform = [b < F or b > idx_h for b in range(nbins)]
    else:
        if length < 2 * F + 2:
            F = 0
        sh = bin_shift(max(length - 2 * F, 0), nb)
        hi0 = length - F + 1                  # first position of the right fine zone
        nbins = F + nb + F

        def bin_of(pos):
            if pos < F:
                return pos
            if F and pos >= hi0:
                return F + nb + (pos - hi0)
            return F + ((pos - F) >> sh)

        def first_pos(b):
            if b < F:
                return b
            if b >= F + nb:
                return hi0 + (b - F - nb)
            return F + ((b - F) << sh)

        uniform = [b < F or b >= F + nb for b in range(nbins)]
    S, E, Z = [0] * nbins, [0] * nbins, [0] * nbins
    keys = []
    for s, e in intervals:
        assert 0 <= s <= e <= length
        if s == e:
            b = bin_of(s)
            if uniform[b]:
                Z[b] += 1
            else:
                S[b] += 1
                E[b] += 1
            keys += [(s << SH) | 1, (s << SH) | 2]
        else:
            S[bin_of(s)] += 1
            E[bin_of(e)] += 1
            keys += [(s << SH) | 3, e << SH]
    # per bin: kept ends / kept starts (uniform), or everything / nothing (coarse)
    D = 0
    keep_e, keep_s, keep_z, d_in = [0] * nbins, [0] * nbins, [0] * nbins, [0] * nbins
    for b in range(nbins):
        d_in[b] = D
        if uniform[b]:
            n_de = min(max(D - cov - 1, 0), E[b])
            keep_e[b] = E[b] - n_de
            keep_s[b] = min(max(cov + 1 - (D - E[b]), 0), S[b])
            keep_z[b] = Z[b] if D - E[b] <= cov else 0
        elif D - E[b] > cov:                  # spanned by more than cov intervals: all deep
            keep_e[b] = keep_s[b] = 0
        else:
            keep_e[b], keep_s[b] = E[b], S[b]
        D += S[b] - E[b]
    assert D == 0
    out = []
    took_e, took_s = [0] * nbins, [0] * nbins
    for k in keys:                            # any keep_x[b] of a uniform bin's equal keys
        b = bin_of(k >> SH)
        cls = k & 3
        if uniform[b] and cls in (1, 2):
            if keep_z[b]:
                out.append(k)
        elif k & 1:
            if took_s[b] < keep_s[b]:
                took_s[b] += 1
                out.append(k)
        elif took_e[b] < keep_e[b]:
            took_e[b] += 1
            out.append(k)
    # stand-ins: in front of every bin that keeps something, the net of the dropped run before it
    A = 0                                     # depth after the kept block of the last such bin
    for b in range(nbins):
        opaque = keep_e[b] + keep_s[b] + keep_z[b] > 0 or (not uniform[b] and d_in[b] - E[b] <= cov)
        if not opaque:
            continue
        n_de = E[b] - keep_e[b]
        net = (d_in[b] - n_de) - A
        fp = first_pos(b)
        if net > 0:
            assert fp >= 1
            out += [(fp << SH) - 1] * net     # starts at fp - 1: after everything before the bin
        elif net < 0:
            out += [fp << SH] * (-net)        # ends at fp: before everything else in the bin
        A = d_in[b] - E[b] + keep_s[b]
    assert A == 0
    return out


def trimmed_events(intervals, length, cov, nb=16, F=32, wave=False):
    """sweep over the keys trim_keys leaves; None unless the read is plain."""
    if len(intervals) == 0:
        return [(0, length)] if length != 0 else []
    if any(not (0 <= s <= e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    keys = sorted(trim_keys(intervals, length, cov, nb, F, wave))
    for a, b in zip(keys, keys[1:]):
        if a == b and (a & 3) == 1 and a != 1:
            return None  # two zero-length intervals at one position survive: exact general path
    return sweep_keys(keys, length, cov)


def trim_keys_minmax(intervals, length, cov, nb):
    """The filter of sweep_wave.h's deferring build (`trimfilter`): nb coarse bins of 2^sh positions
    aligned at 0, plus one bin for the read's SMALLEST START and one for its LARGEST END — where the
    clamped halves of the dovetail piles sit, at 0 and `length` for a healthy read, at the edges of
    the covered window for a read that is only covered in part.  Nothing else can lie at those two
    positions (an end is above its own start, a start below its own end), so in key order the
    sequence is [starts at pmin][coarse bins][ends at pmax].  Plain reads only (0 <= s < e <= length)."""
    sh = bin_shift(length, nb)
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    S, E = [0] * nb, [0] * nb
    S0 = E1 = 0
    keys = []
    for s, e in intervals:
        assert 0 <= s < e <= length
        if s == pmin:
            S0 += 1
        else:
            S[s >> sh] += 1
        if e == pmax:
            E1 += 1
        else:
            E[e >> sh] += 1
        keys += [(s << SH) | 3, e << SH]
    ks0, ke1 = min(S0, cov + 1), min(E1, cov + 1)
    out = [(pmin << SH) | 3] * ks0 + [pmax << SH] * ke1
    kmin = (pmin << SH) | 3
    D, A = S0, ks0
    for b in range(nb):
        deep = D - E[b] > cov
        if not deep and S[b] + E[b] > 0:      # keeps everything; empty bins keep nothing
            net = D - A
            pk = (b << sh) << SH
            out += [max(pk, kmin + 1) - 1] * net if net > 0 else [pk] * (-net)
            out += [k for k in keys if (k >> SH) >> sh == b and k != kmin and k != (pmax << SH)]
            A = D - E[b] + S[b]
        D += S[b] - E[b]
    assert D == E1
    net1 = ke1 - A
    out += [(pmax << SH) - 1] * net1 if net1 > 0 else [pmax << SH] * (-net1)
    return out


def trimmed_minmax_events(intervals, length, cov, nb=16):
    if len(intervals) == 0:
        return [(0, length)] if length != 0 else []
    if any(not (0 <= s < e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    return sweep_keys(sorted(trim_keys_minmax(intervals, length, cov, nb)), length, cov)


def healthy_read_regions(intervals, length, cov, nb):
    """sweep_wave.h's closed form for the healthy read (trimfilter tier 3): when every coarse bin
    that holds an event is deep and min(S0, cov + 1) == min(E1, cov + 1) =: k, what trim_keys_minmax
    keeps is k copies of the smallest start key and k of the largest end key, and the sweep over
    those gives: the whole read when k <= cov (nothing exceeds the threshold), otherwise the part in
    front of pmin and the part behind pmax.  Returns None when the read is not of that kind (the
    kernel then runs pass 2, the sort and the sweep).  Plain reads only (0 <= s < e <= length)."""
    if len(intervals) == 0:
        return [(0, length)] if length != 0 else []
    sh = bin_shift(length, nb)
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    S, E = [0] * nb, [0] * nb
    S0 = E1 = 0
    for s, e in intervals:
        assert 0 <= s < e <= length
        if s == pmin:
            S0 += 1
        else:
            S[s >> sh] += 1
        if e == pmax:
            E1 += 1
        else:
            E[e >> sh] += 1
    D = S0
    for b in range(nb):
        if not (D - E[b] > cov) and S[b] + E[b] > 0:
            return None
        D += S[b] - E[b]
    k = min(S0, cov + 1)
    if k != min(E1, cov + 1):
        return None
    if k <= cov:
        return [(0, length)]
    return ([(0, pmin)] if pmin != 0 else []) + ([(pmax, length)] if pmax != length else [])


def window_screen_regions(intervals, length, cov, nb, W):
    """sweep_wave.h's healthy-read screen on ORDER STATISTICS (round 3; replaces the exact-position
    piles of healthy_read_regions).  With a = the (cov+1)-th smallest start and b = the (cov+1)-th
    largest end, a plain read whose starts beyond the first cov+1 all find more than cov intervals open
    is bad exactly in front of a and behind b: src/stack.rs:83-89 assigns first_covered at the first
    cov+1 starts (heap sizes 0..cov, nothing popped yet) and never opens a gap afterwards, and the tail
    loop :93-105 pops down to cov open intervals, i.e. ends on the (cov+1)-th largest end (or breaks on
    an end == len, which then is that end as well).
    The kernel finds a and b without a sort: W one-position bins counted from the read's smallest start
    pmin upwards (starts only) and W from its largest end pmax downwards (ends only), next to the nb
    coarse bins of 2^sh positions that hold every other event.  When no end lies inside the head window
    and no start inside the tail window (smallest end >= pmin + W, largest start <= pmax - W), in event
    order the read is [head window: F starts][coarse bins][tail window: G ends], and a coarse-counted
    start of bin i has at least F + (starts of bins < i) - (ends of bins <= i) intervals open in front of it.
    Returns None when the screen does not apply (the kernel then defers the read to the sort)."""
    n = len(intervals)
    if n == 0:
        return [(0, length)] if length != 0 else []
    if any(not (0 <= s < e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    if n <= cov:  # never more than cov intervals open: nothing is flagged, the whole read is bad
        return [(0, length)]
    if n < 2:
        return None
    sh = bin_shift(length, nb)
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    if min(e for s, e in intervals) - pmin < W or pmax - max(s for s, e in intervals) < W:
        return None
    S, E = [0] * nb, [0] * nb
    FH, FT = [0] * W, [0] * W
    for s, e in intervals:
        if s - pmin < W:
            FH[s - pmin] += 1
        else:
            S[s >> sh] += 1
        if pmax - e < W:
            FT[pmax - e] += 1
        else:
            E[e >> sh] += 1
    F, G = sum(FH), sum(FT)
    if F < cov + 1 or G < cov + 1:
        return None
    D = F
    for b in range(nb):
        if S[b] > 0 and not (D - E[b] > cov):
            return None
        D += S[b] - E[b]
    acc, a = 0, None
    for i in range(W):
        acc += FH[i]
        if acc >= cov + 1:
            a = pmin + i
            break
    acc, bb = 0, None
    for i in range(W):
        acc += FT[i]
        if acc >= cov + 1:
            bb = pmax - i
            break
    return ([(0, a)] if a != 0 else []) + ([(bb, length)] if bb != length else [])


def _sub_screen(intervals, length, cov, nb, W, lo0, hi0, P0, Q0, max_slides, ramp_always, s_hi=None, e_lo=None, spot=0, tail_ramp=False, jump=False):
    """The order-statistics screen over the events inside [lo0, hi0] of a read (starts and ends outside are not
    counted; P0 starts in front of lo0 and Q0 ends behind hi0 are carried as counts: intervals open across the
    border), with windows that slide by W up to max_slides times.  Returns (a, b, slides) — a: where P0 + the starts
    counted upwards from lo0 reach cov + 1, b: where Q0 + the ends counted downwards from hi0 do — or None.
    s_hi / e_lo: only starts below s_hi / ends from e_lo on belong to the sub-read (the halves of a read with a hole:
    a start AT the hole's far side is the other half's, an end AT it this half's)."""
    sh = bin_shift(length, nb)
    s_hi = hi0 + 1 if s_hi is None else s_hi
    e_lo = lo0 if e_lo is None else e_lo
    intervals = [(s if s < s_hi else 2**40, e if e >= e_lo else -1) for s, e in intervals]  # (out of every range below)
    ends_in = [e for s, e in intervals if e >= lo0]
    starts_in = [s for s, e in intervals if s <= hi0]
    if not ends_in or not starts_in:
        return None
    emin, smax = min(ends_in), max(starts_in)
    h0 = t0 = 0
    for slide in range(max_slides + 1):
        if emin - lo0 < h0 + W or hi0 - smax < t0 + W or (hi0 - lo0) - h0 - t0 < 2 * W:
            return None
        lo, hi = lo0 + h0, hi0 - t0
        P = P0 + sum(1 for s, e in intervals if lo0 <= s < lo)
        Q = Q0 + sum(1 for s, e in intervals if hi < e <= hi0)
        if P > cov or Q > cov:
            return None
        S, E = [0] * (nb + 1), [0] * (nb + 1)
        FH, FT = [0] * W, [0] * W
        ramp = 0  # starts behind the head window but in front of the smallest end: nothing has been popped when they
        #           arrive, the heap holds every start before them — more than cov once a has passed — so they are never
        #           low and, like the window's starts, precede every coarse-counted start and every end
        for s, e in intervals:
            if lo <= s <= hi:
                if s - lo < W:
                    FH[s - lo] += 1
                elif (slide > 0 or ramp_always) and s < emin:
                    ramp += 1
                else:
                    S[min((s - lo) >> sh, nb)] += 1
            if lo <= e <= hi:
                if hi - e < W:
                    FT[hi - e] += 1
                elif tail_ramp and (slide > 0 or ramp_always) and e > smax:
                    # THE RAMP'S MIRROR (round 6): an end behind the read's largest start is popped after every start has
                    # arrived — no start finds it gone — so it is in no block's count of "ends that may precede my starts".
                    # (What it was costing: dovetail ends spread over the last coarse blocks of a read, every one of them
                    # counted as popped before the block's starts: 8 % of configs[1]'s reads at sigma = 300.)
                    pass
                else:
                    E[min((e - lo) >> sh, nb)] += 1
        F, G = P + sum(FH), Q + sum(FT)
        if F < cov + 1 or G < cov + 1:
            if slide == max_slides:
                return None
            if jump:
                # WINDOWS THAT JUMP (round 6): a window that came up short moves to the next event it has not seen — the head
                # window begins AT the smallest start behind it, the tail window ends AT the largest end in front of it —
                # instead of by W: nothing lies in between, so what it has passed is what it counted (P = F, Q = G), every
                # pass gains at least one event, and cov + 1 passes always reach cov + 1.
                if F < cov + 1:
                    nxt = [s for s, e in intervals if lo + W <= s <= hi0]
                    if not nxt:
                        return None
                    h0 = min(nxt) - lo0
                if G < cov + 1:
                    prv = [e for s, e in intervals if lo0 <= e <= hi - W]
                    if not prv:
                        return None
                    t0 = hi0 - max(prv)
            else:
                h0 += W if F < cov + 1 else 0
                t0 += W if G < cov + 1 else 0
            continue
        D = F + ramp
        failing = []
        for b in range(nb + 1):
            if S[b] > 0 and not (D - E[b] > cov):
                failing.append(b)
            D += S[b] - E[b]
        if failing:
            # SPOT CHECKS (round 6, `spot` > 0): the block test counts ALL of a block's ends as before its starts; where that
            # is too coarse — dovetail ends spread over a whole block at ONT depth — the few coarse-counted starts of the
            # failing blocks are looked at one by one: a start s has at least (starts at positions < s) - (ends at positions
            # <= s) intervals open in front of it (src/stack.rs:72-83: the ends at or before s are popped first); more than
            # cov for each of them, and no start beyond the first cov + 1 is low after all.
            if not spot:
                return None
            cand = [s for s, e in intervals if lo <= s <= hi and s - lo >= W and not ((slide > 0 or ramp_always) and s < emin)
                    and min((s - lo) >> sh, nb) in failing]
            if len(cand) > spot:
                return None
            for s0 in cand:
                if not (P0 + sum(1 for s, e in intervals if lo0 <= s < s0) - sum(1 for s, e in intervals if e <= s0) > cov):
                    return None
        acc, a = P, None
        for i in range(W):
            acc += FH[i]
            if acc >= cov + 1:
                a = lo + i
                break
        acc, bb = Q, None
        for i in range(W):
            acc += FT[i]
            if acc >= cov + 1:
                bb = hi - i
                break
        return a, bb, slide
    return None


def slid_window_screen_regions(intervals, length, cov, nb, W, max_slides, ramp_always=False, spot=0, tail_ramp=False, jump=False):
    """window_screen_regions with windows that SLIDE (round 4): when the first W positions hold fewer than cov + 1
    starts (or the last W fewer than cov + 1 ends) — dovetail ends spread wider than the window — the screen is
    repeated with that window moved on by W, the events it has passed carried as a count: P starts in front of the
    head window, Q ends behind the tail window.  Nothing else changes: in event order the read is
    [P passed starts][head window][coarse blocks, re-based at the head window][tail window][Q passed ends], the passed
    starts precede every end (smallest end >= pmin + h0 + W is required), the passed ends follow every start
    (largest start <= pmax - t0 - W), so a coarse-counted start still has at least
    (P + window starts) + (coarse starts before its block) - (ends through its block) intervals open, a is where
    P + the window's running count reaches cov + 1 and b likewise from the top.  max_slides = 0 is
    window_screen_regions.  ramp_always: the starts between the head window and the read's smallest end count as
    "already open" from the first pass on (the kernel's second look at a read whose window holds few starts), not
    only after a slide.  tail_ramp: on those same passes the ends behind the read's largest start (and in front of
    the tail window) are left out of the coarse blocks — the ramp's mirror.  jump: a window that came up short moves to
    the next event instead of by W.  Returns (regions, slides used) or None."""
    n = len(intervals)
    if n == 0:
        return ([(0, length)] if length != 0 else []), 0
    if any(not (0 <= s < e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    if n <= cov:
        return [(0, length)], 0
    if n < 2:
        return None
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    r = _sub_screen(intervals, length, cov, nb, W, pmin, pmax, 0, 0, max_slides, ramp_always, spot=spot, tail_ramp=tail_ramp, jump=jump)
    if r is None:
        return None
    a, bb, slide = r
    return ([(0, a)] if a != 0 else []) + ([(bb, length)] if bb != length else []), slide


def hole_screen_regions(intervals, length, cov, nb, W, max_slides, nbf=64):
    """The OTHER closed form (round 4): a read whose coverage drops to cov or less in ONE stretch inside — what a
    chimera looks like, the reads yacrd exists to find.  If a position sp inside the stretch has no event in
    [sp, sp + w) and at most cov intervals across it, the read falls into the intervals that end before sp (L), those
    that start behind it (R) and k <= cov that span it; with hiL = the largest end below sp, loR = the smallest start
    at or above it and no event strictly between the two, the reference's sweep (src/stack.rs:61-139) does on L and
    the k spanning intervals what it does on a healthy read whose last k intervals never end — its tail stops at
    x = where the ends counted down from hiL, the k included, reach cov + 1 (the pops from there on leave cov or
    fewer in the heap: not flagged, :77-79) — then finds the heap at k when R's first start arrives, opens a gap
    (x, s) at each of R's first cov + 1 - k starts (:83-89; merged by equal begin, :119-136, to (x, y) with y where
    the starts counted up from loR, the k included, reach cov + 1) and goes on as on a healthy read: the regions are
    (0, a), (x, y), (b, len).  Both halves are the sub-read screen (_sub_screen) with the spanning intervals carried.
    How loR is found (what the kernel can do without a sort): coarse blocks of 2^sh positions from the smallest
    start; the first block, behind a block whose depth bound exceeds cov, whose bound does not; nbf sub-bins over that
    block and the next; the first sub-bin that holds a start and whose depth bound (its own ends taken first) is <= cov;
    its smallest start.  Everything else follows from loR: L = the intervals that end at or before it, R = those that
    start at or behind it, the k others span it.  A wrong guess costs nothing but the closed form: k > cov, or one of
    the two halves fails its screen.  Returns the regions or None."""
    n = len(intervals)
    if n < 2 or n <= cov or any(not (0 <= s < e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    sh = bin_shift(length, nb)
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    S, E = [0] * (nb + 2), [0] * (nb + 2)
    for s, e in intervals:
        S[min((s - pmin) >> sh, nb + 1)] += 1
        E[min((e - pmin) >> sh, nb + 1)] += 1
    seen_deep, istar, sb, eb = False, None, 0, 0
    for i in range(nb + 2):
        x = sb - (eb + E[i])  # starts before block i - ends through it
        if x > cov:
            seen_deep = True
        elif seen_deep:
            istar = i
            break
        sb += S[i]
        eb += E[i]
    if istar is None:
        return None
    B0 = pmin + (istar << sh)
    w = max(1, (2 << sh) // nbf)
    FS, FE = [0] * nbf, [0] * nbf
    D0 = 0
    for s, e in intervals:
        if s < B0:
            D0 += 1
        elif (s - B0) // w < nbf:
            FS[(s - B0) // w] += 1
        if e < B0:
            D0 -= 1
        elif (e - B0) // w < nbf:
            FE[(e - B0) // w] += 1
    jstar, D = None, D0
    for j in range(nbf):
        if FS[j] > 0 and D - FE[j] <= cov:  # a start that may find cov or fewer intervals open
            jstar = j
            break
        D += FS[j] - FE[j]
    if jstar is None:
        return None
    loR = min(s for s, e in intervals if s >= B0 and (s - B0) // w == jstar)  # R = the intervals that start at or behind it
    Lends = [e for s, e in intervals if e <= loR]  # (an end AT loR is popped before that start is looked at, :72-81)
    if not Lends:
        return None
    hiL = max(Lends)
    k = sum(1 for s, e in intervals if s < loR < e)
    if k > cov:
        return None
    if any(hiL < s < loR for s, e in intervals):
        return None  # a spanning interval that starts after L's last end: it would arrive at a heap already drained
    left = _sub_screen(intervals, length, cov, nb, W, pmin, hiL, 0, k, max_slides, True, s_hi=loR)
    if left is None:
        return None
    right = _sub_screen(intervals, length, cov, nb, W, loR, pmax, k, 0, max_slides, True, e_lo=loR + 1)
    if right is None:
        return None
    a, x, _ = left
    y, bb, _ = right
    return ([(0, a)] if a != 0 else []) + [(x, y)] + ([(bb, length)] if bb != length else [])


def hole_fast_regions(intervals, length, cov, nb, W, nbf=64):
    """hole_screen_regions the way sweep_wave.h computes it (round 4): no second screen of the two halves — a and b,
    the windows' counts F and G and the coarse blocks' depth bounds are the FIRST screen's (window_screen_regions: it
    found them and failed on the depth test of some block), and what the hole adds is looked at where it lies:
      1. istar = the first coarse block whose bound fails; only it and its successor may fail;
      2. nbf sub-bins over those two blocks, the depth D0 in front of them (> cov required: the hole lies behind the
         covered part) carried through them; jstar = the first sub-bin with a start whose bound (its own ends first)
         is <= cov; loR = its smallest start;
      3. hiL = the largest end at or before loR, k = the intervals across loR (<= cov), no start between hiL and loR;
      4. x from W one-position bins downwards from hiL (k carried; no start of the left half inside them),
         y from W one-position bins upwards from loR (k carried; no end inside them);
      5. behind jstar every sub-bin that may hold a start at or behind the right half's smallest end must be deep
         again (the starts in front of that end are the ramp: nothing popped yet).
    Returns the regions or None."""
    n = len(intervals)
    if n < 2 or n <= cov or any(not (0 <= s < e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    sh = bin_shift(length, nb)
    sh = max(sh, (W - 1).bit_length())  # blocks of at least W positions, like the kernel
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    if min(e - s for s, e in intervals) < W or pmax - pmin < 2 * W:
        return None
    span, T = pmax - pmin, pmax - pmin - W
    NBIN = 2 * W + nb
    FH, FT = [0] * W, [0] * W
    CS, CE = [0] * NBIN, [0] * NBIN
    for s, e in intervals:
        ds, dx = s - pmin, e - pmin
        if ds < W:
            FH[ds] += 1
        else:
            CS[min(W + (ds >> sh), NBIN - 1)] += 1
        if span - dx < W:
            FT[span - dx] += 1
        CE[min(W + (dx >> sh) + max(dx - T, 0), NBIN - 1)] += 1
    F, G = sum(FH), sum(FT)
    if F < cov + 1 or G < cov + 1:
        return None
    acc, a = 0, None
    for i in range(W):
        acc += FH[i]
        if acc >= cov + 1:
            a = pmin + i
            break
    acc, bb = 0, None
    for i in range(W):
        acc += FT[i]
        if acc >= cov + 1:
            bb = pmax - i
            break
    # 1. the coarse blocks' bounds (bins W .. W + nb - 1)
    failing, sb, eb = [], 0, 0
    for i in range(nb):
        eb += CE[W + i]
        if CS[W + i] > 0 and not (F + sb - eb > cov):
            failing.append(i)
        sb += CS[W + i]
    if not failing or failing[0] == 0 or any(i > failing[0] + 1 for i in failing):
        return None
    istar = failing[0]
    # 2. sub-bins over blocks istar, istar + 1
    B0 = pmin + (istar << sh)
    fsh = sh + 1 - (nbf.bit_length() - 1)
    if fsh < 0:
        return None
    FS, FE = [0] * nbf, [0] * nbf
    D0 = 0
    for s, e in intervals:
        if s < B0:
            D0 += 1
        elif (s - B0) >> fsh < nbf:
            FS[(s - B0) >> fsh] += 1
        if e < B0:
            D0 -= 1
        elif (e - B0) >> fsh < nbf:
            FE[(e - B0) >> fsh] += 1
    if D0 <= cov:
        return None
    jstar, D, Dj = None, D0, [0] * nbf
    for j in range(nbf):
        Dj[j] = D
        if jstar is None and FS[j] > 0 and D - FE[j] <= cov:
            jstar = j
        D += FS[j] - FE[j]
    if jstar is None:
        return None
    loR = min(s for s, e in intervals if s >= B0 and (s - B0) >> fsh == jstar)
    # 3.
    Lends = [e for s, e in intervals if e <= loR]
    if not Lends:
        return None
    hiL = max(Lends)
    k = sum(1 for s, e in intervals if s < loR < e)
    if k > cov or any(hiL < s < loR for s, e in intervals):
        return None
    # 4. the two windows at the hole
    if any(hiL - W < s < loR for s, e in intervals):  # a start of the left half inside its tail window
        return None
    emin_r = min([e for s, e in intervals if e > loR] or [None])
    if emin_r is None or emin_r - loR < W:  # an end inside the right half's head window
        return None
    acc, x = k, None
    for d in range(W):
        acc += sum(1 for s, e in intervals if e == hiL - d)
        if acc >= cov + 1:
            x = hiL - d
            break
    acc, y = k, None
    for d in range(W):
        acc += sum(1 for s, e in intervals if s == loR + d)
        if acc >= cov + 1:
            y = loR + d
            break
    if x is None or y is None:
        return None
    # 5. behind the hole: the ramp (starts in front of the right half's smallest end), then deep again
    if B0 + ((jstar + 1) << fsh) - 1 >= emin_r:  # the sub-bin of loR reaches the right half's smallest end
        return None
    for j in range(jstar + 1, nbf):
        last = B0 + ((j + 1) << fsh) - 1
        if FS[j] > 0 and last >= emin_r and not (Dj[j] - FE[j] > cov):
            return None
    return ([(0, a)] if a != 0 else []) + [(x, y)] + ([(bb, length)] if bb != length else [])


def drop_inert_at_pmin(intervals, cov):
    """Zero-length intervals AT the read's smallest start pmin, when a regular interval starts there too and cov >= 1, change
    nothing (round 6; the screens leave them out of every count): sorted first among the intervals at pmin, each is pushed
    at depth 0 — a low start: first_covered = pmin, as the regular start at pmin will assign it again — and popped, alone in
    the heap, in front of the next push: flagged only if 1 > cov (src/stack.rs:72-89).  A regular interval cannot END at
    pmin, so every end there is such an interval's.  (SURVEY.md 8d's generator clamps a degenerate interval of a read
    that is covered only inside a window onto the window's first position: without this rule an end `at or before a`
    sent the read to the sort.)"""
    live = [iv for iv in intervals if iv != (0, 0)]
    if cov < 1 or not live:
        return intervals
    pmin = min(s for s, e in live)
    if not any(s == pmin and e > s for s, e in live):
        return intervals
    return [iv for iv in intervals if iv != (pmin, pmin)]


def unified_screen_regions(intervals, length, cov, nb, W):
    """screen_wg.h / screen_big.h: the order-statistics screen with ONE position map for starts and ends,
        idx(x) = min(dx, W) + (dx >> sh) + max(dx - T, 0),   dx = x - pmin,  T = (pmax - pmin) - W,  2^sh >= W:
    a bin per position in the first W and the last W positions of the covered span, coarse blocks of 2^sh
    positions in between, monotone in x.  With cs / ce the running counts of starts / ends in bin order:
      * a = position of the bin where cs reaches cov + 1: it must be a head-window bin, with no end at or
        before it (src/stack.rs:83-89 then assigns first_covered at exactly the first cov + 1 starts);
      * b = position of the bin where the count of ends from the top reaches cov + 1: a tail-window bin
        (the tail loop :93-105 stops there);
      * every bin that holds a start beyond the first cov + 1 must have more than cov intervals open even
        after all of its own ends: cs_before - ce_through > cov.
    Unlike window_screen_regions (the register classes) it tolerates intervals shorter than W anywhere —
    a read of thousands of intervals nearly always has one — because starts inside the tail window and ends
    inside the head window have bins of their own.  None = not decided here."""
    n = len(intervals)
    if n == 0:
        return [(0, length)] if length != 0 else []
    # (zero-length intervals are taken: where more than cov intervals are open on both sides of one it changes
    # nothing — its start is not low, its flagged end is superseded before the next low start — and the
    # tests below put it nowhere else: as an end it may not lie at or before a, as a start its bin must be deep)
    if any(not (0 <= s <= e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    # (0, 0) intervals are inert in the reference — popped at once, their pop re-assigns last_covered = 0 while
    # it still is 0, their low start sets first_covered = 0 — and are left out of every count
    intervals = [iv for iv in drop_inert_at_pmin(intervals, cov) if iv != (0, 0)]
    n = len(intervals)
    if n < 2 or n <= cov:
        return None
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    span = pmax - pmin
    if span < 2 * W:
        return None
    sh = bin_shift(length, nb)
    while (1 << sh) < W:
        sh += 1
    T = span - W

    def idx(x):
        dx = x - pmin
        return min(dx, W) + (dx >> sh) + max(dx - T, 0)

    nbins = idx(pmax) + 1
    assert nbins <= 2 * W + nb
    S, E = [0] * nbins, [0] * nbins
    for s, e in intervals:
        S[idx(s)] += 1
        E[idx(e)] += 1
    k1 = cov + 1
    # a / b: the kernel walks the head window upwards from pmin (start bins) and the tail window downwards
    # from pmax (end bins, idx(pmax - d)) until the running count reaches cov + 1
    a = sorted(s for s, e in intervals)[cov]
    b = sorted(e for s, e in intervals)[n - 1 - cov]
    if a - pmin >= W or pmax - b >= W or any(e <= a for s, e in intervals):
        return None
    cs = ce = 0
    for i in range(nbins):
        cs_ex = cs
        cs += S[i]
        ce += E[i]
        if S[i] and cs_ex >= k1 and not (cs_ex - ce > cov):
            return None
    return ([(0, a)] if a != 0 else []) + ([(b, length)] if b != length else [])


def filtered_sweep_regions(intervals, length, cov, nb, W, cap, stats=None):
    """finish_compact.h's follow-on for the reads the screen defers (round 5): the screen's closed form for the two ENDS
    of the read, an exact sweep over what lies in its UNSAFE coarse blocks for the inside — any number of holes, no
    sort of the whole read.  The table is healthy_screen's: W one-position bins from the smallest start pmin upwards
    (F starts) and from the largest end pmax downwards (G ends), nb coarse blocks of 2^sh positions for the rest
    (start -> block (s - pmin) >> sh when s - pmin >= W; end -> block (e - pmin) >> sh when it lies in front of the
    tail window).  Every interval at least W long: no end inside the head window, no start inside the tail window;
    F > cov and G > cov: a = the (cov+1)-th smallest start and b = the (cov+1)-th largest end lie in their windows and
    the reference (src/stack.rs:83-113) gives (0, a) in front and (b, len) behind.
    D_i = F + (coarse starts - coarse ends of the blocks before i) is the EXACT depth on entry to block i, D_i - E_i
    the least depth any of its events sees: a block with D_i - E_i > cov is safe — its starts are never low
    (:83), its ends are all flagged (:77-79).  A low start s (depth <= cov) therefore lies in an unsafe block, and
    the flagged end the reference pairs it with — the last one in front of it — lies in an unsafe block too or is
    the largest end of the nearest block in front that holds an end (every end of a safe block is flagged).  Kept:
    the unsafe blocks and, for each, the nearest block in front that holds an end.  The kept events are sorted and
    swept with their true depths (D of the block + the kept events in front inside it); a run of low starts still
    open when the keys end is closed by the tail (G > cov: the (cov+1)-th largest end is flagged behind it).
    None: not this kind of read, or more than `cap` kept events — the caller sorts the read whole."""
    n = len(intervals)
    if n < 2 or length >= 2**30 - 1 or n <= cov:
        return None
    if any(not (0 <= s < e <= length) or e - s < W for s, e in intervals):
        return None
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    span = pmax - pmin
    T = span - W
    sh = bin_shift(length, nb)
    while (1 << sh) < W:
        sh += 1
    FH, FT = [0] * W, [0] * W
    S, E = [0] * (nb + 1), [0] * (nb + 1)
    for s, e in intervals:
        ds, dx = s - pmin, e - pmin
        if ds < W:
            FH[ds] += 1
        else:
            S[ds >> sh] += 1
        if dx > T:
            FT[span - dx] += 1
        else:
            E[dx >> sh] += 1
    F, G = sum(FH), sum(FT)
    if F <= cov or G <= cov:
        return None
    acc, a = 0, None
    for i in range(W):
        acc += FH[i]
        if acc >= cov + 1:
            a = pmin + i
            break
    acc, b = 0, None
    for i in range(W):
        acc += FT[i]
        if acc >= cov + 1:
            b = pmax - i
            break
    iT = T >> sh
    D, d = [0] * (iT + 1), F
    for i in range(iT + 1):
        D[i] = d
        d += S[i] - E[i]
    unsafe = [S[i] + E[i] > 0 and D[i] - E[i] <= cov for i in range(iT + 1)]
    kept = list(unsafe)
    for i in range(iT + 1):
        if not unsafe[i] and E[i] > 0:
            nxt = next((j for j in range(i + 1, iT + 1) if unsafe[j] or E[j] > 0), None)
            kept[i] = nxt is not None and unsafe[nxt]
    m = sum(S[i] + E[i] for i in range(iT + 1) if kept[i])
    if stats is not None:
        stats.append(m)
    if m > cap:
        return None
    corr, before = {}, 0
    for i in range(iT + 1):
        if kept[i]:
            corr[i] = D[i] - before
            before += S[i] - E[i]
    keys = []
    for s, e in intervals:
        ds, dx = s - pmin, e - pmin
        if ds >= W and kept[ds >> sh]:
            keys.append((s << SH) | 3)
        if dx <= T and kept[dx >> sh]:
            keys.append(e << SH)
    keys.sort()
    assert len(keys) == m
    out, run, tc, cml = [], 0, None, None
    for key in keys:
        pos = key >> SH
        depth = run + corr[(pos - pmin) >> sh]
        if key & 1:
            if depth <= cov:
                if tc is None:
                    return None  # (cannot happen with F > cov: kept for the kernel's guard)
                cml = pos
            run += 1
        else:
            if depth > cov:
                if cml is not None and cml >= tc:
                    out.append((tc, cml))
                    cml = None
                tc = pos
            run -= 1
    if cml is not None and cml >= tc:
        out.append((tc, cml))
    return ([(0, a)] if a != 0 else []) + out + ([(b, length)] if b != length else [])


def unified_filtered_regions(intervals, length, cov, nb, W, cap, stats=None):
    """screen_wg.h's fallback for the reads its screen cannot decide (round 6): filtered_sweep_regions' idea on
    unified_screen_regions' table — ONE position map for starts and ends, idx(x) = min(dx, W) + (dx >> sh) + max(dx - T, 0),
    so the bins are in event order, every position of the first / last W a bin of its own — for reads of thousands of
    intervals, short ones anywhere.
    The two ENDS by the screen's closed form: a = the position where the starts counted upwards from pmin reach cov + 1 (a
    head-window bin, no end at or before it), b = the position where the ends counted downwards from pmax reach cov + 1 (a
    tail-window bin): src/stack.rs:83-113 gives (0, a) in front and (b, len) behind.
    The INSIDE exactly, from the bins that can matter: D_i = starts - ends of the bins in front of i is the depth on entry
    to bin i, D_i - E_i the least depth any of its events sees (inside a bin its ends count as before its starts).  A bin
    with D_i - E_i > cov is SAFE: none of its starts is low (:83), all of its ends are flagged (:77-79).  A low start lies
    in an unsafe bin, and the flagged end the reference pairs it with — the last one in front of it — lies in an unsafe bin
    too or is the largest end of the nearest bin in front that holds an end.  Kept: the unsafe bins behind a's and in front
    of b's, and for each the nearest bin in front that holds an end; their events sorted and swept with their true depths.
    A bin at or behind b's with a shallow start, a zero-length interval in a kept bin with cov or fewer intervals open in
    front of it, more than `cap` kept events, or anything the screen itself would not take: None — the caller sorts the read."""
    n = len(intervals)
    if n == 0:
        return [(0, length)] if length != 0 else []
    if any(not (0 <= s <= e <= length) for s, e in intervals) or length >= 2**30 - 1:
        return None
    intervals = [iv for iv in drop_inert_at_pmin(intervals, cov) if iv != (0, 0)]
    n = len(intervals)
    if n < 2 or n <= cov:
        return None
    pmin = min(s for s, e in intervals)
    pmax = max(e for s, e in intervals)
    span = pmax - pmin
    if span < 2 * W:
        return None
    sh = bin_shift(length, nb)
    while (1 << sh) < W:
        sh += 1
    T = span - W

    def idx(x):
        dx = x - pmin
        return min(dx, W) + (dx >> sh) + max(dx - T, 0)

    nbins = idx(pmax) + 1
    S, E = [0] * nbins, [0] * nbins
    for s, e in intervals:
        S[idx(s)] += 1
        E[idx(e)] += 1
    a = sorted(s for s, e in intervals)[cov]
    b = sorted(e for s, e in intervals)[n - 1 - cov]
    if a - pmin >= W or pmax - b >= W or any(e <= a for s, e in intervals):
        return None
    ia, ib = idx(a), idx(b)
    D, d = [0] * nbins, 0
    for i in range(nbins):
        D[i] = d
        d += S[i] - E[i]
    unsafe = [i > ia and S[i] + E[i] > 0 and D[i] - E[i] <= cov for i in range(nbins)]
    if any(unsafe[i] and S[i] > 0 for i in range(ib, nbins)):
        return None  # a start that may be low at or behind b: the closed form for the tail does not hold
    inside = range(ia + 1, ib)
    kept = [False] * nbins
    for i in inside:
        if unsafe[i]:
            kept[i] = True
        elif E[i] > 0:
            nxt = next((j for j in range(i + 1, ib) if unsafe[j] or E[j] > 0), None)
            kept[i] = nxt is not None and unsafe[nxt]
    m = sum(S[i] + E[i] for i in inside if kept[i])
    if stats is not None:
        stats.append(m)
    if m > cap:
        return None
    corr, before = {}, 0
    for i in inside:
        if kept[i]:
            corr[i] = D[i] - before
            before += S[i] - E[i]
    keys = []
    for s, e in intervals:
        if s == e:  # a zero-length interval: its keys sort between the position's ends and its regular starts (regular_keys)
            if kept[idx(s)]:
                keys += [(s << SH) | 1, (s << SH) | 2]
            continue
        if kept[idx(s)]:
            keys.append((s << SH) | 3)
        if kept[idx(e)]:
            keys.append(e << SH)
    keys.sort()
    assert len(keys) == m
    out, run, tc, cml = [], 0, None, None
    for key in keys:
        pos = key >> SH
        depth = run + corr[idx(pos)]
        if key & 1:
            if (key & 3) == 1 and depth <= cov:
                return None  # a zero-length interval whose start is low: the sort's (one with more than cov open around it is an
                             # ordinary pair of keys: its start is not low, its end is flagged where it stands)
            if depth <= cov:
                if tc is None:
                    return None  # a low start with no flagged end in front of it among the kept (cannot happen: kept for the kernel's guard)
                cml = pos
            run += 1
        else:
            if depth > cov:
                if cml is not None and cml >= tc:
                    out.append((tc, cml))
                    cml = None
                tc = pos
            run -= 1
    if cml is not None and cml >= tc:
        out.append((tc, cml))
    return ([(0, a)] if a != 0 else []) + out + ([(b, length)] if b != length else [])
