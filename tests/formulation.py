"""Sequential Python emulation of the two GPU formulations, line for line what
yacrd_amd/csrc/sweep_lds.h and sweep_general.h compute, minus the parallel scans.  Lets the
formulations be fuzzed against the oracle on CPU (tests/test_formulation.py)."""

NO = 0xFFFFFFFF


def finish_read(slot, mf_t, ml_t, min_ge, length):
    lcf = min_ge if min_ge != NO else (mf_t >> 1)
    if ml_t > mf_t:
        b, e = mf_t >> 1, ml_t >> 1
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


def regular_events(intervals, length, cov):
    """Event formulation for regular reads (start < end < 2^31); None if not regular."""
    n = len(intervals)
    if n == 0:
        return [(0, length)] if length != 0 else []
    keys = []
    max_start = 0
    for s, e in intervals:
        if s >= e or e >= 2**31:
            return None
        keys.append((s << 1) | 1)
        keys.append(e << 1)
        max_start = max(max_start, (s << 1) | 1)
    keys.sort()
    d = 0
    mf = ml = 0
    min_ge = NO
    slot = []
    for key in keys:
        if key & 1:
            if d <= cov:
                ml = key
            d += 1
        else:
            if d > cov:
                if ml > mf and not (mf == 0 and (ml >> 1) == 0):
                    slot.append((mf >> 1, ml >> 1))
                mf = key
                if key > max_start and (key >> 1) >= length:
                    min_ge = min(min_ge, key >> 1)
            d -= 1
    return finish_read(slot, mf, ml, min_ge, length)


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
