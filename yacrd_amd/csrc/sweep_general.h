// sweep_general.h — exact path for ANY input: degenerate intervals (start >= end), positions
// >= 2^31 and reads too large for LDS.  One 1024-thread workgroup per read, 64-bit keys in a
// global-memory scratch.  It is the slow, always-correct route; the LDS / register sweeps are
// fast paths for regular reads and must agree with it (tests force every read through here).
//
// Formulation, exact for the reference's sequential semantics
// (src/stack.rs:61-139) on arbitrary u32 input:
//   1. sort intervals by (start,end)  == ovls.sort_unstable(), stack.rs:66
//   2. interval j is popped at step t(j) = first i > j with start_i >= end_j
//        regular (end > start):  t(j) = lower_bound(starts, end_j)
//        degenerate:             t(j) = j + 1   (it is the heap minimum right after its push)
//   3. event keys  start_i -> (2i+1, start_i),  pop_j -> (2 t(j), end_j); sorting them gives the
//      reference's exact push/pop order (pops of one step ascend by value: BinaryHeap order)
//   4. depth_before = prefix sum; pop flagged <=> depth_before > c (stack.rs:77-79, :93);
//      last_covered = value of the last flagged pop (by position, not by max: values may
//      decrease when degenerate intervals are present); start low <=> depth_before <= c
//   5. raw gaps (stack.rs:83-89), tail break (:93-105), prepend/append (:107-113),
//      equal-begin merge with "max of the last two" (:119-136) done literally.
#pragma once
#include "device_common.h"

namespace yk {

struct GeneralArgs {
    const u64 *off;
    const uint2 *iv;
    const u32 *len;
    const u32 *list;        // read ids
    const u64 *scratch_off; // per list entry, in u64 elements
    u64 *scratch;           // per read: K[n+2] then EV[2n]
    u32 cov;
    uint2 *stage;
    u32 *counts;
};

constexpr int kGenThreads = 1024;

// Ascending bitonic network in the "flip" form: every compare-exchange puts the minimum at the
// lower index, so virtual +inf padding above n never moves and pairs reaching past n are skipped.
template <int T>
__device__ __forceinline__ void bitonic_sort_global_u64(u64 *a, u32 n)
{
    if (n < 2) return;
    u32 P = 2;
    while (P < n) P <<= 1;
    for (u32 k = 2; k <= P; k <<= 1) {
        const u32 h = k >> 1;
        for (u32 p = threadIdx.x; p < (P >> 1); p += T) { // flip stage: i <-> block mirror
            const u32 blk = p / h, x = p - blk * h;
            const u32 i = blk * k + x, l = blk * k + (k - 1 - x);
            if (l < n) {
                const u64 va = a[i], vb = a[l];
                if (va > vb) {
                    a[i] = vb;
                    a[l] = va;
                }
            }
        }
        __syncthreads();
        for (u32 j = h >> 1; j > 0; j >>= 1) { // half cleaners
            for (u32 p = threadIdx.x; p < (P >> 1); p += T) {
                const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const u32 l = i + j;
                if (l < n) {
                    const u64 va = a[i], vb = a[l];
                    if (va > vb) {
                        a[i] = vb;
                        a[l] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// One read, T threads, scratch K[n+2] / EV[2n] of u64 anywhere (global memory or LDS).
template <int T>
__device__ __forceinline__ void general_read(const uint2 *iv, u32 n, u32 len, u32 cov, uint2 *slot,
                                             u32 *count_out, u64 *K, u64 *EV, u32 *sc, u32 *s_misc)
{
    const u32 tid = threadIdx.x;
    if (n == 0) {
        if (tid == 0) {
            u32 g = 0;
            if (len != 0) slot[g++] = make_uint2(0, len);
            *count_out = g;
        }
        return;
    }
    const u32 M = 2 * n;

    // 1. interval keys
    for (u32 i = tid; i < n; i += T) {
        const uint2 v = iv[i];
        K[i] = ((u64)v.x << 32) | v.y;
    }
    __syncthreads();
    bitonic_sort_global_u64<T>(K, n);

    // 2-3. event keys
    for (u32 j = tid; j < n; j += T) {
        const u64 key = K[j];
        const u32 s = (u32)(key >> 32), e = (u32)key;
        u32 t;
        if (e > s) { // first i with start_i >= e
            u32 lo = 0, hi = n;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if ((u32)(K[mid] >> 32) >= e) hi = mid;
                else lo = mid + 1;
            }
            t = lo;
        } else {
            t = j + 1;
        }
        EV[2 * j] = ((u64)(2 * j + 1) << 32) | s;
        EV[2 * j + 1] = ((u64)(2 * t) << 32) | e;
    }
    __syncthreads();
    bitonic_sort_global_u64<T>(EV, M);

    // 4. blocked chunks over the sorted events
    const u32 Kc = (M + T - 1) / T;
    const u32 q0 = min(tid * Kc, M), q1 = min(q0 + Kc, M);

    u32 delta = 0;
    for (u32 q = q0; q < q1; q++) delta += ((u32)(EV[q] >> 32) & 1u) ? 1u : 0xFFFFFFFFu;
    u32 tot;
    const u32 depth_in = block_excl_add<T>(delta, sc, tot);

    // last flagged pop (index + 1) per chunk; first flagged tail pop with value >= len
    u32 d = depth_in, lf = 0, brk = kNoKey;
    for (u32 q = q0; q < q1; q++) {
        const u64 ev = EV[q];
        const u32 hi = (u32)(ev >> 32), val = (u32)ev;
        if (hi & 1u) {
            d++;
        } else {
            if (d > cov) {
                lf = q + 1;
                if (hi == M && val >= len) brk = min(brk, q);
            }
            d--;
        }
    }
    u32 lf_t;
    const u32 lf_in = block_excl_max<T>(lf, sc, lf_t);
    brk = block_min<T>(brk, sc);

    // raw gaps per chunk; last low start that still sees last_covered == 0
    u32 cnt = 0, fci = 0;
    u32 lc = lf_in ? (u32)EV[lf_in - 1] : 0u;
    d = depth_in;
    for (u32 q = q0; q < q1; q++) {
        const u64 ev = EV[q];
        const u32 hi = (u32)(ev >> 32), val = (u32)ev;
        if (hi & 1u) {
            if (d <= cov) {
                if (lc != 0) cnt++;
                else fci = q + 1;
            }
            d++;
        } else {
            if (d > cov) lc = val;
            d--;
        }
    }
    u32 n_raw;
    u32 pos = block_excl_add<T>(cnt, sc, n_raw);
    fci = block_max<T>(fci, sc);

    uint2 *RAW = (uint2 *)K; // K is dead: EV carries every value.  RAW[0] = prepend slot.
    if (cnt) {
        lc = lf_in ? (u32)EV[lf_in - 1] : 0u;
        d = depth_in;
        for (u32 q = q0; q < q1; q++) {
            const u64 ev = EV[q];
            const u32 hi = (u32)(ev >> 32), val = (u32)ev;
            if (hi & 1u) {
                if (d <= cov && lc != 0) RAW[1 + pos++] = make_uint2(lc, val);
                d++;
            } else {
                if (d > cov) lc = val;
                d--;
            }
        }
    }
    if (tid == 0) {
        const u32 fc = fci ? (u32)EV[fci - 1] : 0u;                       // stack.rs:87
        const u32 lcf = (brk != kNoKey) ? (u32)EV[brk]                     // stack.rs:101-103
                                        : (lf_t ? (u32)EV[lf_t - 1] : 0u);
        u32 base = 1, total = n_raw;
        if (fc != 0) {                                                     // stack.rs:107-109
            RAW[0] = make_uint2(0, fc);
            base = 0;
            total++;
        }
        if (lcf != len) {                                                  // stack.rs:111-113
            RAW[base + total] = make_uint2(lcf, len);
            total++;
        }
        s_misc[0] = base;
        s_misc[1] = total;
    }
    __syncthreads();

    // 5. equal-begin merge (stack.rs:119-136): a run's region is (begin, max of its last two ends)
    const uint2 *L = RAW + s_misc[0];
    const u32 total = s_misc[1];
    const u32 Kr = (total + T - 1) / T;
    const u32 t0 = min(tid * Kr, total), t1 = min(t0 + Kr, total);
    u32 tails = 0;
    for (u32 t = t0; t < t1; t++)
        tails += (t + 1 == total) || (L[t + 1].x != L[t].x);
    u32 g;
    u32 w = block_excl_add<T>(tails, sc, g);
    for (u32 t = t0; t < t1; t++) {
        const uint2 cur = L[t];
        if ((t + 1 == total) || (L[t + 1].x != cur.x)) {
            u32 e = cur.y;
            if (t > 0 && L[t - 1].x == cur.x) e = max(e, L[t - 1].y);
            slot[w++] = make_uint2(cur.x, e);
        }
    }
    if (tid == 0) *count_out = g;
    __syncthreads(); // scratch may be reused by the caller's next read
}

// Global-memory scratch: any read size.  One workgroup per list entry.
__global__ __launch_bounds__(kGenThreads) void sweep_general_kernel(GeneralArgs a)
{
    __shared__ u32 sc[kGenThreads / 64];
    __shared__ u32 s_misc[4];
    const u32 r = a.list[blockIdx.x];
    const u64 o = a.off[r];
    const u32 n = (u32)(a.off[r + 1] - o);
    u64 *K = a.scratch + a.scratch_off[blockIdx.x];
    general_read<kGenThreads>(a.iv + o, n, a.len[r], a.cov, a.stage + (o + 2 * (u64)r),
                              a.counts + r, K, K + (n + 2), sc, s_misc);
}

// LDS scratch: the exact path for reads a sweep rejected (degenerate interval), without a host
// round trip.  T threads per read, reads of at most CAPN intervals; grid-strides over the
// device-side rejection list.  Reads larger than CAPN are forwarded to `big_list`.
template <int T, int CAPN>
__global__ __launch_bounds__(T) void sweep_general_lds_kernel(SweepArgs a)
{
    __shared__ u64 scratch[3 * CAPN + 2];
    __shared__ u32 sc[T / 64 + 1];
    __shared__ u32 s_misc[4];
    const u32 list_n = *a.list_n;
    for (u32 b = blockIdx.x; b < list_n; b += gridDim.x) {
        const u32 r = a.list[b];
        const u64 o = a.off[r];
        const u32 n = (u32)(a.off[r + 1] - o);
        if (n > (u32)CAPN) {
            if (threadIdx.x == 0) {
                a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
                a.counts[r] = 0;
            }
            continue;
        }
        general_read<T>(a.iv + o, n, a.len[r], a.cov, a.stage + (o + 2 * (u64)r), a.counts + r,
                        scratch, scratch + (n + 2), sc, s_misc);
    }
}

// what the host needs to lay out scratch / key buffers for the listed reads
struct GatherOut {
    u64 iv_off;
    u64 n;
    u32 len;
    u32 read;
};
__global__ __launch_bounds__(256) void gather_general_sizes_kernel(const u64 *off, const u32 *len,
                                                                   const u32 *list, u32 count,
                                                                   GatherOut *out)
{
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const u32 r = list[i];
    GatherOut g;
    g.iv_off = off[r];
    g.n = off[r + 1] - off[r];
    g.len = len[r];
    g.read = r;
    out[i] = g;
}

} // namespace yk
