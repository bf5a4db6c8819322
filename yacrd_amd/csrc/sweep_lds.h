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

// ---- coverage pre-filter for the workgroup classes (same rule as sweep_wave.h) --
// NB = T bins, one per thread.  The staging loop histograms the read's events into the last
// 4 * T words of the (still empty) key array while it looks for degenerate intervals;
// lds_filter_plan() turns the counters into slot counters, writes the stand-in keys of the safe
// runs and says how many keys survive; a second pass over the intervals (L2-resident) then writes
// each surviving key to the slot it is handed.  Used when the read is plain (no zero-length
// interval, every end <= len) and keeps at most CAP / 2 keys.
template <int T, int CAP>
struct LdsFilter {
    static constexpr int NB = T;
    static constexpr u32 kNowhere = 0x80000000u;
    static_assert(CAP / 2 <= CAP - 5 * NB, "survivors and counters must not overlap");
    static __device__ __forceinline__ u32 shift_of(u32 len)
    {
        const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(NB) + (len != 0 ? 0 : -1);
        return (u32)max(bits, 0) + kKeyShift;
    }
    static __device__ __forceinline__ u32 *hist(u32 *keys) { return keys + (CAP - 4 * NB); }
    // Add `val` to this thread's counter of the bin that holds `key` (four counters per bin, one
    // per lane & 3: dovetail overlaps pile thousands of starts into bin 0 and as many ends into
    // the bin of `len`) and return the counter's previous value.  (One wave-aggregated atomic
    // for the lanes that hit those two bins was tried: no measurable change.)
    static __device__ __forceinline__ u32 add(u32 *keys, u32 key, u32 ksh, u32 val)
    {
        return atomicAdd(hist(keys) + min(key >> ksh, (u32)(NB - 1)) * 4u + (threadIdx.x & 3u), val);
    }
};

// Returns the number of keys that survive (0 = do not filter).  syn_start = largest stand-in
// start key this thread wrote (for the tail rule's max_start), else 0.
template <int T, int CAP>
__device__ __forceinline__ u32 lds_filter_plan(u32 *keys, u32 len, u32 cov, u32 *sc, u32 &syn_start)
{
    using F = LdsFilter<T, CAP>;
    constexpr int NB = T;
    const u32 tid = threadIdx.x;
    const i32 c = (i32)min(cov, 0x7FFFFFFFu);
    const u32 ksh = F::shift_of(len), sh = ksh - kKeyShift;
    u32 *flags = keys + (CAP - 5 * NB); // [NB]: safe bits, for the neighbours
    uint4 *my_bin = reinterpret_cast<uint4 *>(F::hist(keys)) + tid;
    syn_start = 0;

    const uint4 w4 = *my_bin;
    const u32 w = w4.x + w4.y + w4.z + w4.w; // starts in the low half, ends in the high half
    u32 w_tot;
    const u32 incl = block_excl_add<T>(w, sc, w_tot) + w;
    const i32 S = (i32)(w & 0xFFFFu), E = (i32)(w >> 16);
    const i32 cs = (i32)(incl & 0xFFFFu), ce = (i32)(incl >> 16);
    const i32 depth_after = cs - ce, depth_at = depth_after - (S - E);
    const bool safe = (cs - S) - ce > c && tid < (len >> sh);
    flags[tid] = safe ? 1u : 0u;
    if (block_max<T>(safe ? 1u : 0u, sc) == 0) return 0; // nothing to drop (ends with a barrier)
    const bool head = safe && !(tid > 0 && flags[tid - 1]);
    const bool tail = safe && !(tid + 1 < (u32)NB && flags[tid + 1]);
    const u32 hv_own = head ? (((tid + 1u) << 16) | (u32)depth_at) : 0u;
    u32 hv_tot;
    const u32 hv = max(block_excl_max<T>(hv_own, sc, hv_tot), hv_own);
    const i32 net = tail ? depth_after - (i32)(hv & 0xFFFFu) : 0;
    const u32 nsyn = (u32)(net < 0 ? -net : net);
    const u32 synkey = (((hv >> 16) - 1u) << ksh) | (net > 0 ? 3u : 0u);
    const u32 mine = safe ? nsyn : (u32)(S + E);
    u32 m_new;
    const u32 base = block_excl_add<T>(mine, sc, m_new);
    if (m_new > (u32)(CAP / 2)) return 0; // would not shrink the sort
    uint4 b4;
    b4.x = base;
    b4.y = b4.x + (w4.x & 0xFFFFu) + (w4.x >> 16);
    b4.z = b4.y + (w4.y & 0xFFFFu) + (w4.y >> 16);
    b4.w = b4.z + (w4.z & 0xFFFFu) + (w4.z >> 16);
    *my_bin = safe ? make_uint4(F::kNowhere, F::kNowhere, F::kNowhere, F::kNowhere) : b4;
#pragma unroll 1
    for (u32 t = 0; t < nsyn; t++) keys[base + t] = synkey;
    if (tail && net > 0) syn_start = synkey;
    __syncthreads();
    return m_new;
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
        const u32 r = a.list[b];
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
            continue;
        }

        const u32 m = 2 * n;
        u32 P = 2;
        while (P < m) P <<= 1;

        // ---- first pass over the intervals (coalesced 8 B/lane loads): degenerate ones, the
        // largest start key, and the pre-filter's histogram (into the tail of the empty key array)
        using F = LdsFilter<T, CAP>;
        const bool try_filter = T >= 256 && a.prefilter != 0;
        const u32 ksh = F::shift_of(len);
        if (try_filter) {
            reinterpret_cast<uint4 *>(F::hist(keys))[tid] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        u32 bad = 0, max_start = 0, nz = 0; // bad bit 1: an end beyond the read (no pre-filter)
        const uint2 *iv = a.iv + o;
        for (u32 i = tid; i < n; i += T) {
            u32 ks, ke;
            const uint2 v = iv[i];
            make_event_keys(v, ks, ke, bad, nz);
            bad |= v.y > len ? 2u : 0u;
            max_start = max(max_start, ks);
            if (try_filter) {
                F::add(keys, ks, ksh, 1u);
                F::add(keys, ke, ksh, 0x10000u);
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
            continue;
        }
        max_start = block_max<T>(max_start, sc);

        // ---- second pass: the event keys go to LDS — the survivors of the pre-filter to the
        // slots they are handed, or all of them
        u32 m_sort = 0, syn_start = 0; // events to sort and sweep
        if (try_filter && nz_total == 0 && !beyond)
            m_sort = lds_filter_plan<T, CAP>(keys, len, a.cov, sc, syn_start);
        if (m_sort) {
            u32 ms = syn_start;
            for (u32 i = tid; i < n; i += T) {
                const uint2 v = iv[i];
                const u32 ks = (v.x << kKeyShift) | 3u, ke = v.y << kKeyShift;
                const u32 ps = F::add(keys, ks, ksh, 1u);
                const u32 pe = F::add(keys, ke, ksh, 1u);
                if (ps < F::kNowhere) {
                    keys[ps] = ks;
                    ms = max(ms, ks);
                }
                if (pe < F::kNowhere) keys[pe] = ke;
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
                    a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
                    a.counts[r] = 0;
                }
                __syncthreads();
                continue;
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
                if (d <= a.cov) ml = key;
                d++;
            } else {
                if (d > a.cov) mf = max(mf, key ^ 2u);
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
                if (d <= a.cov) ml = key;
                d++;
            } else {
                if (d > a.cov) {
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
                    if (d <= a.cov) ml = key;
                    d++;
                } else {
                    if (d > a.cov && (key ^ 2u) > mf) {
                        if (ml > (mf ^ 2u))
                            slot[pos++] = make_uint2((mf ^ 2u) >> kKeyShift, ml >> kKeyShift);
                        mf = key ^ 2u;
                    }
                    d--;
                }
            }
        }
        if (tid == 0)
            a.counts[r] = finish_read(slot, g_closed, mf_t ? (mf_t ^ 2u) : 0u, ml_t, min_ge, len);
        __syncthreads(); // keys / sc reused by the next read
    }
}

} // namespace yk
