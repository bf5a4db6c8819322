// screen_big.h — the healthy-read screen (tests/formulation.py::unified_screen_regions) for
// reads of more than 16 384 intervals, run by the WHOLE device: a read of 763 000 intervals is 6 MB, one
// workgroup would stream it for 100 us.
//   setup    one workgroup: the BIG class list -> one segment record per read and the chunk -> segment map
//            (chunks of 4096 intervals), built on the device (rounds 1-2 built such tables on the host: a
//            gather kernel, a read-back and two uploads per batch)
//   minmax   grid over the chunks: smallest start, largest end, largest start of every read (atomics on its
//            record), whether it is plain (no start > end, no position beyond the read / the key range)
//   hist     grid over the chunks: starts and ends per bin of screen_wg.h's position map (a bin per position in
//            the first / last W = 512 positions of the covered span, 1024 coarse blocks in between),
//            privatised in LDS, the non-zero counters flushed to the read's table
//   verdict  one workgroup per read: the tests of screen_wg.h over the 2048 bins; the read's two regions in
//            closed form, or the read goes to the fallback list (sweep_big_trim.h / sweep_big.h take it after
//            the run's final sync: rare)
// No sort, no key buffer, two passes over the intervals (the second one from L2 for all but the largest reads).
// Reference semantics: src/stack.rs:61-139 via the screen's rule (see screen_wg.h).
#pragma once
#include "device_common.h"
#include "sweep_wave.h"

namespace yk {

constexpr int kBsW = 512, kBsNB = 1024, kBsBins = 2 * kBsW + kBsNB; // window positions, coarse blocks, bins per read
constexpr int kBsT = 256;       // threads of minmax / hist
constexpr int kBsChunk = 4096;  // intervals per workgroup of minmax / hist
constexpr int kBsVT = 1024;     // threads of the verdict kernel: two bins each

struct BsSeg {
    u64 iv_off;    // first interval in the CSR
    u32 n, len, read, chunk_off;
    u32 pmin, pmax, smax, bad; // device-written by minmax (bad: 1 = not plain)
    u32 pad[2];
};

struct BsArgs {
    const u64 *off;
    const uint2 *iv;
    const u32 *len;
    const u32 *list;    // the BIG class list
    const u32 *list_n;
    BsSeg *seg;         // [reads of the class]
    u32 *chunk_seg;     // [max_chunks]
    u32 *n_chunks;      // device word: chunks in all
    u32 *hist;          // [reads][2][kBsBins]: starts, ends
    u32 max_chunks, max_segs;
    u32 cov, count_healthy;
    uint2 *stage;
    u32 *counts;
    u32 *fb_list;       // reads the screen leaves to the trimming filter / the segmented sort
    u32 *fb_count;
    Counters *ctr;
};

__global__ __launch_bounds__(1024) void bs_setup_kernel(BsArgs a)
{
    __shared__ u32 sc[1024 / 64 + 1];
    __shared__ u32 s_carry;
    const u32 n_segs = min(*a.list_n, a.max_segs);
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_segs; base += 1024u) { // (uniform)
        const u32 si = base + threadIdx.x;
        u32 r = 0, n = 0, nch = 0;
        u64 o = 0;
        if (si < n_segs) {
            r = a.list[si];
            o = a.off[r];
            n = (u32)min(a.off[r + 1] - o, (u64)0xFFFFFFFFu);
            nch = (n + (u32)kBsChunk - 1u) / (u32)kBsChunk;
        }
        u32 tot;
        const u32 first = s_carry + block_excl_add<1024>(nch, sc, tot);
        if (si < n_segs) {
            BsSeg s;
            s.iv_off = o, s.n = n, s.len = a.len[r], s.read = r, s.chunk_off = first;
            s.pmin = 0xFFFFFFFFu, s.pmax = 0, s.smax = 0, s.bad = 0, s.pad[0] = s.pad[1] = 0;
            a.seg[si] = s;
            for (u32 c = 0; c < nch && first + c < a.max_chunks; c++) a.chunk_seg[first + c] = si;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *a.n_chunks = min(s_carry, a.max_chunks);
}

__global__ __launch_bounds__(kBsT) void bs_minmax_kernel(BsArgs a)
{
    constexpr int R = kBsChunk / kBsT;
    __shared__ u32 red[kBsT / 64][4];
    const u32 c = blockIdx.x;
    if (c >= *a.n_chunks) return;
    const u32 si = a.chunk_seg[c];
    const BsSeg s = a.seg[si];
    const u32 i0 = (c - s.chunk_off) * (u32)kBsChunk, i1 = min(i0 + (u32)kBsChunk, s.n);
    const uint2 *iv = a.iv + s.iv_off;
    uint2 v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = iv[min(i0 + threadIdx.x + (u32)(j * kBsT), i1 - 1u)]; // (copies beyond the chunk)
    u32 smin = 0xFFFFFFFFu, smax = 0, emax = 0, bad = 0;
#pragma unroll
    for (int j = 0; j < R; j++) {
        smin = min(smin, v[j].y != 0u ? v[j].x : 0xFFFFFFFFu); // ((0, 0) intervals are inert: left out)
        smax = max(smax, v[j].x);
        emax = max(emax, v[j].y);
        bad |= v[j].x > v[j].y ? 1u : 0u;
    }
    smin = wave_min(smin);
    smax = wave_max(smax);
    emax = wave_max(emax);
    bad = wave_or(bad);
    const u32 wv = threadIdx.x >> 6;
    if (lane_id() == 0) red[wv][0] = smin, red[wv][1] = smax, red[wv][2] = emax, red[wv][3] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBsT / 64; w++) {
            smin = min(smin, red[w][0]);
            smax = max(smax, red[w][1]);
            emax = max(emax, red[w][2]);
            bad |= red[w][3];
        }
        BsSeg *g = a.seg + si;
        atomicMin(&g->pmin, smin);
        atomicMax(&g->smax, smax);
        atomicMax(&g->pmax, emax);
        if (bad) atomicOr(&g->bad, 1u);
    }
}

// whether the verdict kernel can decide the read at all (the same test in hist, which then skips its chunks)
__device__ __forceinline__ bool bs_screenable(const BsSeg &s)
{
    return s.bad == 0u && s.len <= kMaxKeyPos && s.pmax <= s.len && s.smax <= kMaxKeyPos && s.pmax >= s.pmin &&
           s.pmax - s.pmin >= (u32)(2 * kBsW);
}
__device__ __forceinline__ u32 bs_shift(u32 len)
{
    const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(kBsNB) + (len != 0 ? 0 : -1);
    return (u32)max(bits, ilog2c(kBsW));
}

__global__ __launch_bounds__(kBsT) void bs_hist_kernel(BsArgs a)
{
    constexpr int R = kBsChunk / kBsT;
    __shared__ u32 hs[kBsBins], he[kBsBins];
    const u32 c = blockIdx.x;
    if (c >= *a.n_chunks) return;
    const u32 si = a.chunk_seg[c];
    const BsSeg s = a.seg[si];
    if (!bs_screenable(s)) return; // (uniform)
    const u32 sh = bs_shift(s.len), span = s.pmax - s.pmin, Tt = span - (u32)kBsW;
    for (u32 b = threadIdx.x; b < (u32)kBsBins; b += kBsT) hs[b] = he[b] = 0;
    __syncthreads();
    const u32 i0 = (c - s.chunk_off) * (u32)kBsChunk, i1 = min(i0 + (u32)kBsChunk, s.n);
    const uint2 *iv = a.iv + s.iv_off;
    uint2 v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = iv[min(i0 + threadIdx.x + (u32)(j * kBsT), i1 - 1u)];
#pragma unroll
    for (int j = 0; j < R; j++) { // (every load is issued before the first atomic)
        const u32 ds = v[j].x - s.pmin, dx = v[j].y - s.pmin;
        const u32 is = min(ds, (u32)kBsW) + (ds >> sh) + __builtin_elementwise_sub_sat(ds, Tt);
        const u32 ie = min(dx, (u32)kBsW) + (dx >> sh) + __builtin_elementwise_sub_sat(dx, Tt);
        if (i0 + threadIdx.x + (u32)(j * kBsT) < i1 && v[j].y != 0u) {
            atomicAdd(&hs[is], 1u);
            atomicAdd(&he[ie], 1u);
        }
    }
    __syncthreads();
    u32 *gh = a.hist + (size_t)si * 2 * kBsBins;
    for (u32 b = threadIdx.x; b < (u32)kBsBins; b += kBsT) {
        if (hs[b]) atomicAdd(&gh[b], hs[b]);
        if (he[b]) atomicAdd(&gh[kBsBins + b], he[b]);
    }
}

__global__ __launch_bounds__(kBsVT) void bs_verdict_kernel(BsArgs a)
{
    constexpr int T = kBsVT, W = kBsW;
    static_assert(kBsBins == 2 * kBsVT, "two bins per thread");
    __shared__ u32 sc[T / 64 + 1];
    const u32 si = blockIdx.x, tid = threadIdx.x;
    if (si >= min(*a.list_n, a.max_segs)) return;
    const BsSeg s = a.seg[si];
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    bool healthy = false;
    u32 ra = 0, rb = 0;
    if (bs_screenable(s) && s.n > (u32)c) { // (uniform)
        const u32 sh = bs_shift(s.len), span = s.pmax - s.pmin, Tt = span - (u32)W;
        const u32 *gs = a.hist + (size_t)si * 2 * kBsBins, *ge = gs + kBsBins;
        // ---- windows: thread d < W looks at window position d: the starts at pmin + d, the ends at pmax - d
        u32 hS = 0, hE = 0, tE = 0;
        if (tid < (u32)W) {
            const u32 dx = span - tid;
            hS = gs[tid];
            hE = ge[tid];
            tE = ge[min((u32)W + (dx >> sh) + (dx - Tt), (u32)(kBsBins - 1))];
        }
        u32 F, G;
        const u32 hs_ex = block_excl_add<T>(hS, sc, F);
        const u32 te_ex = block_excl_add<T>(tE, sc, G);
        const u32 k1 = (u32)c + 1u;
        u32 notyet = 0;
        bool spoiled = false;
        if (tid < (u32)W) {
            notyet = (hs_ex + hS < k1 ? 1u : 0u) | (te_ex + tE < k1 ? 0x10000u : 0u);
            spoiled = hE != 0u && hs_ex < k1; // an end at or before a
        }
        u32 ntot;
        block_excl_add<T>(notyet, sc, ntot);
        // ---- depth: a bin that holds a start beyond the first c + 1 needs more than c intervals open after all
        // of its own ends; two consecutive bins per thread
        const u32 S0 = gs[2 * tid], S1 = gs[2 * tid + 1], E0 = ge[2 * tid], E1 = ge[2 * tid + 1];
        u32 st, et;
        const u32 cs = block_excl_add<T>(S0 + S1, sc, st);
        const u32 ce = block_excl_add<T>(E0 + E1, sc, et);
        bool shallow = S0 != 0u && cs >= k1 && !((long long)cs - (long long)(ce + E0) > (long long)c);
        shallow |= S1 != 0u && cs + S0 >= k1 && !((long long)(cs + S0) - (long long)(ce + E0 + E1) > (long long)c);
        const u32 any_bad = block_or<T>((shallow || spoiled) ? 1u : 0u, sc);
        healthy = any_bad == 0u && F >= k1 && G >= k1;
        ra = s.pmin + (ntot & 0xFFFFu);
        rb = s.pmax - (ntot >> 16);
    }
    if (tid == 0) {
        if (healthy) {
            uint2 *slot = a.stage + (s.iv_off + 2 * (u64)s.read);
            u32 g = 0;
            if (ra != 0) slot[g++] = make_uint2(0u, ra);
            if (rb != s.len) slot[g++] = make_uint2(rb, s.len);
            a.counts[s.read] = g;
            if (a.count_healthy) atomicAdd(&a.ctr->prefiltered, 1u);
        } else {
            a.counts[s.read] = 0; // (well defined until the fallback has run)
            a.fb_list[atomicAdd(a.fb_count, 1u)] = s.read;
        }
    }
}

} // namespace yk
