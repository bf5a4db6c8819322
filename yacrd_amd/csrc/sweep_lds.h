// sweep_lds.h — one read per workgroup, events sorted in LDS.
//
// Replaces reference src/stack.rs:61-139 (FromOverlap::compute_bad_part) for *regular* reads
// (every interval start < end < 2^31).  Instead of the reference's sort + binary-heap sweep it
// uses the event formulation of DESIGN.md §3:
//   key = position<<2 | class (device_common.h): an end sorts before a start at the same position
//   (the reference pops `head <= interval.0`, stack.rs:72-81); zero-length intervals sit between
//   depth_before(event) = exclusive prefix sum of +1/-1 over the sorted keys
//   end flagged   <=> depth_before > c   (stack.rs:77-79 and the tail loop :93-105)
//   start is low  <=> depth_before <= c  (stack.rs:83)
// A region is closed by the first flagged end after a run of low starts; its begin is the
// previous flagged end (0 = none -> first_covered, stack.rs:84-88) and its end the last low start
// of the run (the equal-begin merge of stack.rs:119-136 keeps exactly that one).
// Reads with a degenerate interval are handed to the general queue untouched.
#pragma once
#include "device_common.h"
#include "sweep_wave.h"

namespace yk {

// ---- register-resident pieces of the workgroup sort ------------------------------------------
// A wavefront holds a 1024-key block striped over its lanes (element = r*64 + lane, r < 16), so
// LDS loads/stores are conflict-free, strides >= 64 are register-to-register and strides < 64
// are DPP / ds_swizzle exchanges (same primitives as sweep_wave.h).  Blocks that must come out
// descending are complemented before and after, so every step below is ascending-only at the
// block level; directions inside a block are compile-time (register bits) or lane constants.
template <int M, int J, int XM>
__device__ __forceinline__ void striped_step(u32 (&x)[16], const LaneConst &lc)
{
    constexpr int K = 16, P = 64 * K;
    if constexpr (J >= 64) {
        constexpr int R = J / 64;
#pragma unroll
        for (int r = 0; r < K; r++) {
            if ((r & R) == 0) {
                const u32 a = x[r], b = x[r | R];
                const bool desc = (M < P) && ((r & (M / 64)) != 0);
                x[r] = desc ? max(a, b) : min(a, b);
                x[r | R] = desc ? min(a, b) : max(a, b);
            }
        }
    } else {
        const u32 kj = lc.k[ilog2c(J)];
        const u32 dir_lane = (M < 64) ? lc.k[ilog2c(M)] : 0u;
#pragma unroll
        for (int r = 0; r < K; r++) {
            const bool desc_r = (M >= 64) && (M < P) && ((r & (M / 64)) != 0);
            const u32 sel = desc_r ? ~kj : (kj ^ dir_lane);
            const u32 t = lane_xor<J, XM>(x[r], lc.addr32);
            x[r] = umed3(x[r], t, sel);
        }
    }
}
template <int M, int J, int XM>
__device__ __forceinline__ void striped_level(u32 (&x)[16], const LaneConst &lc)
{
    striped_step<M, J, XM>(x, lc);
    if constexpr (J > 1) striped_level<M, J / 2, XM>(x, lc);
}
template <int M, int XM>
__device__ __forceinline__ void striped_sort(u32 (&x)[16], const LaneConst &lc)
{
    striped_level<M, M / 2, XM>(x, lc);
    if constexpr (M < 1024) striped_sort<M * 2, XM>(x, lc);
}

// Workgroup sort of P >= 1024 keys in LDS: 1024-key runs sorted in registers, then per level the
// strides >= 1024 as LDS compare-exchange stages and the strides < 1024 again in registers.
// For P = 32768 that is 15 LDS stages instead of 120.
template <int T>
__device__ __forceinline__ void hybrid_sort_lds(u32 *keys, u32 P, const LaneConst &lc)
{
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    constexpr u32 NW = T / 64;
    const u32 n_blocks = P >> 10;
    for (u32 b = wave; b < n_blocks; b += NW) {
        u32 *kb = keys + (b << 10) + lane;
        const u32 flip = (b & 1u) ? 0xFFFFFFFFu : 0u; // level 1024 direction = bit 10 of the index
        u32 x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = kb[r * 64] ^ flip;
        striped_sort<2, 0>(x, lc);
#pragma unroll
        for (int r = 0; r < 16; r++) kb[r * 64] = x[r] ^ flip;
    }
    __syncthreads();
    for (u32 M = 2048; M <= P; M <<= 1) {
        for (u32 j = M >> 1; j >= 1024; j >>= 1) {
            for (u32 p = threadIdx.x; p < (P >> 1); p += T) {
                const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const bool up = (i & M) == 0;
                const u32 a = keys[i], c = keys[l];
                if ((a > c) == up) {
                    keys[i] = c;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
        for (u32 b = wave; b < n_blocks; b += NW) {
            u32 *kb = keys + (b << 10) + lane;
            const u32 flip = ((b << 10) & M) ? 0xFFFFFFFFu : 0u;
            u32 x[16];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = kb[r * 64] ^ flip;
            striped_level<1024, 512, 0>(x, lc);
#pragma unroll
            for (int r = 0; r < 16; r++) kb[r * 64] = x[r] ^ flip;
        }
        __syncthreads();
    }
}

template <int T>
__device__ __forceinline__ void bitonic_sort_lds(u32 *keys, u32 P)
{
    for (u32 k = 2; k <= P; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 p = threadIdx.x; p < (P >> 1); p += T) {
                const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const u32 l = i | j;
                const bool up = (i & k) == 0;
                const u32 a = keys[i], b = keys[l];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

// ---- coverage pre-filter with pile trimming for the workgroup classes (DESIGN.md §3.4 / §3.5;
// emulated and fuzzed in tests/formulation.py::trim_keys) -------------------------------------------
// An event is DEEP when the depth is above c on both sides of it; deep events never matter (a deep
// start is never low, a deep flagged end is always superseded by a later flagged end before the
// next low start), so any of them may be dropped as long as each maximal run of dropped events is
// stood in for by |net| keys of one type.  Which events are deep is known without sorting in two
// kinds of bins:
//   * a coarse bin (2^sh positions) that more than c intervals span completely: all of it;
//   * a bin that holds ONE position: its keys are E ends then S starts, equal within a type, so with
//     the depth D at its head the first max(0, D - c - 1) ends and all but the first
//     max(0, c + 1 - (D - E)) starts are deep.  The first F and the last F positions of the read
//     get such bins: dovetail overlaps pile their starts within a few dozen positions of 0 and their
//     ends near `len`, and of a pile only the c + 1 outermost events survive (configs[3]: a read of
//     5 600 intervals keeps ~10 keys instead of ~3 500).
// Bin sequence in key order: F fine | M + 1 coarse | F fine, idx(pos) below; 2 T bins at most, two
// per thread.  Every bin has four counters; pass 1 counts starts (low half) and ends (high half)
// in the copy of lane & 3; pass 2 re-uses them as packed (quota << 16 | next slot) cursors — the
// copies of a coarse bin keep everything (quota 0x7FFF) or nothing, a fine bin uses copy 0 for its
// ends and copy 1 for its starts with the quotas above — so one LDS atomic per key decides whether
// it is kept and where it goes.  Only for plain reads (every interval start < end <= len).
// Bin geometry: F fine (one-position) bins | up to NB coarse bins | F fine bins, in key order.
template <int NB, int F>
struct TrimGeo {
    u32 f;      // fine positions on either side (F, or 0 for reads too short for fine zones)
    u32 hk;     // key of position H = len - f: the last coarse position
    u32 ksh;    // coarse bin = (key - fk) >> ksh
    u32 m_last; // coarse index of H
    static __device__ __forceinline__ TrimGeo make(u32 len)
    {
        TrimGeo g;
        g.f = len >= 2u * F + 2u ? (u32)F : 0u;
        const u32 span = len - 2u * g.f; // coarse positions f .. len - f
        const i32 bits = 32 - (i32)__builtin_clz(span | 1u) - ilog2c(NB) + (span != 0 ? 0 : -1);
        g.ksh = (u32)max(bits, 0) + kKeyShift;
        g.hk = (len - g.f) << kKeyShift;
        g.m_last = ((len - g.f - g.f) << kKeyShift) >> g.ksh;
        return g;
    }
    __device__ __forceinline__ u32 n_seq() const { return 2u * f + m_last + 1u; }
    // sequence index of the bin that holds `key` (monotone in the key, the class bits do not matter)
    __device__ __forceinline__ u32 idx(u32 key) const
    {
        const u32 pos = key >> kKeyShift, fk = f << kKeyShift;
        const u32 lo = min(pos, f);                               // fine zone at the head
        const u32 mid = (min(max(key, fk), hk) - fk) >> ksh;      // coarse, re-based at f
        const u32 hi = max(key, hk | 3u) - (hk | 3u);             // > 0 beyond position H
        return lo + mid + ((hi + 3u) >> kKeyShift);
    }
    __device__ __forceinline__ u32 first_pos(u32 i) const
    {
        if (i < f) return i;
        if (i <= f + m_last) return f + (((i - f) << ksh) >> kKeyShift);
        return (hk >> kKeyShift) + (i - f - m_last);
    }
    __device__ __forceinline__ bool uniform(u32 i) const { return i < f || i > f + m_last; }
};

template <int T, int CAP>
struct LdsTrim {
    static constexpr int NB = T, F = T / 2, NSEQ = 2 * T;
    static constexpr u32 kTakeOne = 0xFFFF0001u; // quota - 1, slot + 1
    static_assert(T < 256 || CAP / 2 + 5 * NSEQ + 8 <= CAP, "survivors and counters must not overlap");
    using Geo = TrimGeo<NB, F>;
    static __device__ __forceinline__ Geo geo(u32 len) { return Geo::make(len); }
    static __device__ __forceinline__ u32 idx(const Geo &g, u32 key) { return g.idx(key); }
    static __device__ __forceinline__ u32 first_pos(const Geo &g, u32 i) { return g.first_pos(i); }
    static __device__ __forceinline__ bool uniform(const Geo &g, u32 i) { return g.uniform(i); }
    static __device__ __forceinline__ u32 *tab(u32 *keys) { return keys + (CAP - 4 * NSEQ); }
    // zero-length intervals per one-position bin (by sequence index; coarse bins count them as a
    // start and an end of their own)
    static __device__ __forceinline__ u32 *ztab(u32 *keys) { return keys + (CAP - 5 * NSEQ); }
};

// Turns the pass-1 histogram into the pass-2 cursors, writes the stand-in keys and returns the
// number of keys that survive (0 = do not filter).  syn_start = largest stand-in start key.
template <int T, int CAP>
__device__ __forceinline__ u32 lds_trim_plan(u32 *keys, u32 len, u32 cov, u32 *sc, u32 &syn_start)
{
    using F = LdsTrim<T, CAP>;
    const typename F::Geo g = F::geo(len);
    const u32 tid = T == 64 ? lane_id() : threadIdx.x; // (T = 64: one wavefront of a larger workgroup may be the "workgroup": finish_compact.h)
    const i32 c = (i32)min(cov, 0x3FFFFFFFu);
    uint4 *bins = reinterpret_cast<uint4 *>(F::tab(keys));
    const u32 *zt = F::ztab(keys);
    syn_start = 0;
    const u32 n_seq = g.n_seq();

    // this thread's two bins: counts, depth at their heads
    uint4 w4[2];
    i32 S[2], E[2];
    u32 delta = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        w4[k] = bins[2 * tid + k];
        const u32 w = w4[k].x + w4[k].y + w4[k].z + w4[k].w;
        S[k] = (i32)(w & 0xFFFFu);
        E[k] = (i32)(w >> 16);
        delta += (u32)(S[k] - E[k]);
    }
    u32 tot;
    i32 D = (i32)block_excl_add<T>(delta, sc, tot);

    // per bin: what is kept, the depth before / after its kept block
    i32 keep_e[2], keep_s[2], keep_z[2], B[2], A[2];
    bool opaque[2];
    u32 last_own = 0; // (seq index + 1) << 16 | A of this thread's last opaque bin
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const u32 i = 2 * tid + k;
        if (F::uniform(g, i)) {
            const i32 n_de = min(max(D - c - 1, 0), E[k]);
            keep_e[k] = E[k] - n_de;
            keep_s[k] = min(max(c + 1 - (D - E[k]), 0), S[k]);
            // a zero-length interval's two keys sit between the bin's ends and its starts, at depth
            // D - E: deep exactly when the first regular start would be
            keep_z[k] = (D - E[k] <= c) ? 2 * (i32)zt[i] : 0;
            opaque[k] = keep_e[k] + keep_s[k] + keep_z[k] > 0;
        } else {
            const bool deep = D - E[k] > c; // spanned by more than c intervals
            keep_e[k] = deep ? 0 : E[k];
            keep_s[k] = deep ? 0 : S[k];
            keep_z[k] = 0;
            opaque[k] = !deep && i < n_seq;
        }
        B[k] = D - (E[k] - keep_e[k]);
        A[k] = D - E[k] + keep_s[k];
        if (opaque[k]) last_own = ((i + 1u) << 16) | (u32)A[k];
        D += S[k] - E[k];
    }
    u32 any;
    const u32 prev = block_excl_max<T>(last_own, sc, any);
    if (any == 0) return 0; // (cannot happen for a non-empty read; keeps the caller's fallback honest)

    // stand-ins in front of every opaque bin: the net of the dropped run before it
    u32 nsyn[2], synkey[2], mine = 0;
    i32 a_prev = (i32)(prev & 0xFFFFu);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        nsyn[k] = 0;
        synkey[k] = 0;
        if (opaque[k]) {
            const i32 net = B[k] - a_prev;
            const u32 fpk = F::first_pos(g, 2 * tid + k) << kKeyShift;
            nsyn[k] = (u32)(net < 0 ? -net : net);
            synkey[k] = net > 0 ? fpk - 1u : fpk; // starts just before the bin, ends at its head
            a_prev = A[k];
        }
        mine += (u32)(keep_e[k] + keep_s[k] + keep_z[k]) + nsyn[k];
    }
    u32 m_new;
    u32 base = block_excl_add<T>(mine, sc, m_new);
    if (m_new > (u32)(CAP / 2)) return 0; // would not fit / shrink the sort: the caller falls back
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const u32 i = 2 * tid + k;
        uint4 cur;
        if (F::uniform(g, i)) { // copy 0: ends, copy 1: starts, copy 2: keys of zero-length intervals
            cur.x = ((u32)keep_e[k] << 16) | base;
            cur.y = ((u32)keep_s[k] << 16) | (base + (u32)keep_e[k]);
            cur.z = ((u32)keep_z[k] << 16) | (base + (u32)(keep_e[k] + keep_s[k]));
            cur.w = 0;
        } else {
            const u32 q = (keep_e[k] + keep_s[k] > 0) ? 0x7FFF0000u : 0u;
            cur.x = q | base;
            cur.y = q | (base + (w4[k].x & 0xFFFFu) + (w4[k].x >> 16));
            cur.z = q | ((cur.y & 0xFFFFu) + (w4[k].y & 0xFFFFu) + (w4[k].y >> 16));
            cur.w = q | ((cur.z & 0xFFFFu) + (w4[k].z & 0xFFFFu) + (w4[k].z >> 16));
        }
        bins[i] = cur;
        base += (u32)(keep_e[k] + keep_s[k] + keep_z[k]);
#pragma unroll 1
        for (u32 t = 0; t < nsyn[k]; t++) keys[base + t] = synkey[k];
        base += nsyn[k];
        if (nsyn[k] && (synkey[k] & 1u)) syn_start = max(syn_start, synkey[k]);
    }
    if (T == 64) wave_lds_sync();
    else __syncthreads();
    return m_new;
}

// ---- everything after the event keys of a read are in LDS: pad, sort, the four sweep passes,
// finish_read.  Shared by sweep_lds_kernel and the device-wide trimmed path (sweep_big.h).
template <int T>
__device__ __forceinline__ void lds_sort_and_sweep(u32 *keys, u32 *sc, u32 m_sort, u32 len, u32 cov,
                                                   u32 max_start, u32 nz_total, const LaneConst &lc,
                                                   uint2 *slot, u32 *counts, u32 r, u32 *rej_list,
                                                   u32 *rej_count)
{
    const u32 tid = threadIdx.x;
    u32 P;
    P = 2;
    while (P < m_sort) P <<= 1;
    for (u32 i = m_sort + tid; i < P; i += T) keys[i] = kNoKey;
    __syncthreads();

    if (T >= 256 && P >= 1024) hybrid_sort_lds<T>(keys, P, lc);
    else bitonic_sort_lds<T>(keys, P);

    if (nz_total > 2) { // two zero-length intervals at one position > 0: exact path (see keys)
        u32 dup = 0;
        for (u32 i = tid; i + 1 < m_sort; i += T)
            dup |= (keys[i] == keys[i + 1] && (keys[i] & 3u) == 1u && keys[i] != 1u);
        if (block_max<T>(dup, sc)) {
            if (tid == 0) {
                rej_list[atomicAdd(rej_count, 1u)] = r;
                counts[r] = 0;
            }
            __syncthreads();
            return;
        }
    }

    // ---- blocked chunks: thread t owns events [t*K, t*K+K) of the sorted sequence
    const u32 K = (P >= (u32)T) ? P / T : 1;
    const u32 q0 = min(tid * K, m_sort), q1 = min(q0 + K, m_sort);

    // pass A: depth carried into each chunk
    u32 delta = 0;
    for (u32 q = q0; q < q1; q++) delta += (keys[q] & 1u) ? 1u : 0xFFFFFFFFu;
    u32 tot;
    const u32 depth_in = block_excl_add<T>(delta, sc, tot);

    // pass B: last flagged end (flipped domain, device_common.h) / last low start per chunk
    u32 d = depth_in, mf = 0, ml = 0;
    for (u32 q = q0; q < q1; q++) {
        const u32 key = keys[q];
        if (key & 1u) {
            if (d <= cov) ml = key;
            d++;
        } else {
            if (d > cov) mf = max(mf, key ^ 2u);
            d--;
        }
    }
    u32 mf_t, ml_t;
    const u32 mf_in = max(block_excl_max<T>(mf, sc, mf_t), kNoFlag);
    const u32 ml_in = block_excl_max<T>(ml, sc, ml_t);

    // pass C: count the regions this chunk closes; tail rule (stack.rs:93-105)
    u32 cnt = 0, min_ge = kNoKey;
    d = depth_in;
    mf = mf_in;
    ml = ml_in;
    for (u32 q = q0; q < q1; q++) {
        const u32 key = keys[q];
        if (key & 1u) {
            if (d <= cov) ml = key;
            d++;
        } else {
            if (d > cov) {
                if ((key ^ 2u) > mf) {
                    if (ml > (mf ^ 2u)) cnt++;
                    mf = key ^ 2u;
                }
                if (key > max_start && (key >> kKeyShift) >= len)
                    min_ge = min(min_ge, key >> kKeyShift);
            }
            d--;
        }
    }
    u32 g_closed;
    u32 pos = block_excl_add<T>(cnt, sc, g_closed);
    min_ge = block_min<T>(min_ge, sc);

    // pass D: write them, in event order
    if (cnt) {
        d = depth_in;
        mf = mf_in;
        ml = ml_in;
        for (u32 q = q0; q < q1; q++) {
            const u32 key = keys[q];
            if (key & 1u) {
                if (d <= cov) ml = key;
                d++;
            } else {
                if (d > cov && (key ^ 2u) > mf) {
                    if (ml > (mf ^ 2u))
                        slot[pos++] = make_uint2((mf ^ 2u) >> kKeyShift, ml >> kKeyShift);
                    mf = key ^ 2u;
                }
                d--;
            }
        }
    }
    if (tid == 0)
        counts[r] = finish_read(slot, g_closed, mf_t ? (mf_t ^ 2u) : 0u, ml_t, min_ge, len);
}

// One read through the workgroup path: T threads, `keys` = CAP words of LDS (CAP = max events, a power of two,
// CAP % T == 0), `sc` = T / 64 + 1 words.  Every thread of the workgroup calls it with the same r; the caller
// synchronises before the arrays are used again.  A read with more events than the array holds even after the
// pre-filter is appended to a.over_list; one with a degenerate interval to a.rej_list.
template <int T, int CAP>
__device__ __forceinline__ void sweep_lds_read(const SweepArgs &a, u32 r, u32 *keys, u32 *sc, const LaneConst &lc)
{
    const u32 tid = threadIdx.x;
    const u64 o = a.off[r];
    const u32 n = (u32)(a.off[r + 1] - o);
    const u32 len = a.len[r];
    uint2 *slot = a.stage + (o + 2 * (u64)r);

    if (n == 0) { // only reachable through add_length (stack.rs: no loop, tail empty)
        if (tid == 0) {
            u32 g = 0;
            if (len != 0) slot[g++] = make_uint2(0, len);
            a.counts[r] = g;
        }
        return;
    }

    const u32 m = 2 * n;
    // A read with more events than the LDS array holds is still taken when the pre-filter
    // leaves at most CAP / 2 of them (configs[3]'s reads of ~5 600 intervals keep ~30 %: they
    // fit a 256-thread workgroup, five of which share a CU, instead of owning a whole CU as a
    // 1024-thread one); otherwise it goes to over_list for the kernel with the larger array.
    const bool over = m > (u32)CAP;

    // ---- first pass over the intervals (coalesced 8 B/lane loads): degenerate ones, the
    // largest start key, and the pre-filter's histogram (into the tail of the empty key array)
    using F = LdsTrim<T, CAP>;
    const typename F::Geo geo = F::geo(len);
    u32 *tab = F::tab(keys);
    const bool try_filter = T >= 256 && a.prefilter != 0 && len <= kMaxKeyPos;
    if (over && !try_filter) { // uniform
        if (tid == 0) a.over_list[atomicAdd(a.over_count, 1u)] = r;
        return;
    }
    if (try_filter) {
        reinterpret_cast<uint4 *>(tab)[2 * tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4 *>(tab)[2 * tid + 1] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint2 *>(F::ztab(keys))[tid] = make_uint2(0u, 0u);
        __syncthreads();
    }
    u32 bad = 0, max_start = 0, nz = 0; // bad bit 1: an end beyond the read (no pre-filter)
    const uint2 *iv = a.iv + o;
    // (four loads in flight per thread: one memory latency per 4 T intervals instead of per T)
    for (u32 i0 = tid; i0 < n; i0 += 4 * T) {
        uint2 v4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v4[j] = iv[min(i0 + (u32)j * T, n - 1u)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + (u32)j * T >= n) break;
            u32 ks, ke;
            const uint2 v = v4[j];
            make_event_keys(v, ks, ke, bad, nz);
            bad |= v.y > len ? 2u : 0u;
            max_start = max(max_start, ks);
            if (try_filter && v.x <= v.y && v.y <= len) { // (anything else switches the filter off below)
                const u32 is = F::idx(geo, ks);
                if (v.x == v.y && F::uniform(geo, is)) atomicAdd(F::ztab(keys) + is, 1u);
                else {
                    atomicAdd(tab + is * 4u + (tid & 3u), 1u);
                    atomicAdd(tab + F::idx(geo, ke) * 4u + (tid & 3u), 0x10000u);
                }
            }
        }
    }
    bad = block_or<T>(bad, sc);
    const bool beyond = (bad & 2u) != 0;
    bad &= 1u;
    u32 nz_total;
    block_excl_add<T>(nz, sc, nz_total);
    if (bad) { // degenerate interval: exact general path takes the read
        if (tid == 0) {
            a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
            a.counts[r] = 0; // keeps the compaction well defined until the exact path ran
        }
        __syncthreads();
        return;
    }
    max_start = block_max<T>(max_start, sc);

    // ---- second pass: the event keys go to LDS — the survivors of the pre-filter to the
    // slots they are handed, or all of them
    u32 m_sort = 0, syn_start = 0; // events to sort and sweep
    if (try_filter && !beyond)
        m_sort = lds_trim_plan<T, CAP>(keys, len, a.cov, sc, syn_start);
    if (!m_sort && over) { // uniform: too much survives
        if (tid == 0) a.over_list[atomicAdd(a.over_count, 1u)] = r;
        __syncthreads();
        return;
    }
    if (m_sort) {
        u32 ms = syn_start;
        // A cursor whose quota is used up never gets one back, so it is looked at before it is
        // asked (a plain read: lanes reading one address are a broadcast): of a pile of hundreds of
        // equal keys only the first few still do the atomic.
        auto take = [&](u32 *cur) -> u32 {
            return (i32)*reinterpret_cast<volatile u32 *>(cur) >= 0x10000 ? atomicAdd(cur, F::kTakeOne) : 0u;
        };
        for (u32 i0 = tid; i0 < n; i0 += 4 * T) {
            uint2 v4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v4[j] = iv[min(i0 + (u32)j * T, n - 1u)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (i0 + (u32)j * T >= n) break;
                const uint2 v = v4[j];
                u32 ks, ke, b2 = 0, z2 = 0;
                make_event_keys(v, ks, ke, b2, z2);
                const u32 is = F::idx(geo, ks), ie = F::idx(geo, ke);
                const u32 ps = take(tab + is * 4u + (F::uniform(geo, is) ? (z2 ? 2u : 1u) : (tid & 3u)));
                const u32 pe = take(tab + ie * 4u + (F::uniform(geo, ie) ? (z2 ? 2u : 0u) : (tid & 3u)));
                if ((i32)ps >= 0x10000) { // quota left: kept, at the slot in the low half
                    keys[ps & 0xFFFFu] = ks;
                    ms = max(ms, ks);
                }
                if ((i32)pe >= 0x10000) keys[pe & 0xFFFFu] = ke;
            }
        }
        max_start = block_max<T>(ms, sc);
    } else {
        m_sort = m;
        __syncthreads(); // the counters are dead, the keys may overwrite them
        for (u32 i = tid; i < n; i += T) {
            u32 ks, ke, b2 = 0, z2 = 0;
            make_event_keys(iv[i], ks, ke, b2, z2);
            keys[2 * i] = ks;
            keys[2 * i + 1] = ke;
        }
    }
    lds_sort_and_sweep<T>(keys, sc, m_sort, len, a.cov, max_start, nz_total, lc, slot, a.counts, r,
                          a.rej_list, a.rej_count);
}

// T threads per read, CAP = max events (power of two, CAP % T == 0).
template <int T, int CAP>
__global__ __launch_bounds__(T) void sweep_lds_kernel(SweepArgs a)
{
    __shared__ u32 keys[CAP];
    __shared__ u32 sc[T / 64 + 1];

    const u32 tid = threadIdx.x;
    const u32 list_n = *a.list_n;
    LaneConst lc;
#pragma unroll
    for (int i = 0; i < 6; i++) lc.k[i] = (tid & (1u << i)) ? 0xFFFFFFFFu : 0u;
    lc.k[6] = 0;
    lc.addr32 = ((tid & 63u) ^ 32u) << 2;

    for (u32 b = blockIdx.x; b < list_n; b += gridDim.x) {
        sweep_lds_read<T, CAP>(a, a.list[b], keys, sc, lc);
        __syncthreads(); // keys / sc reused by the next read
    }
}

} // namespace yk
