// screen_wg.h — the healthy-read screen for reads of 513 .. 16 384 intervals: one read per
// workgroup, the intervals held in registers, ONE pass over memory.
//
// tests/formulation.py::unified_screen_regions is the emulation (fuzzed against the oracle).  One position map
// for starts and ends,
//     idx(x) = min(dx, W) + (dx >> sh) + max(dx - T, 0),    dx = x - pmin,  T = (pmax - pmin) - W,  2^sh >= W,
// gives every position of the first W and of the last W positions of the covered span a bin of its own and the
// positions in between coarse blocks of 2^sh; it is monotone, so the bins are in event order (inside a bin: its
// ends count as before its starts, which errs on the safe side in a coarse block and is the reference's order
// at one position, src/stack.rs:72-81).  With cs / ce the running counts of starts / ends in bin order:
//   * a = the position where the starts counted upwards from pmin reach c + 1, b = the position where the ends
//     counted downwards from pmax reach c + 1: both must lie inside their windows, and no end at or before a
//     (src/stack.rs:83-89 then assigns first_covered at exactly the first c + 1 starts; the tail loop
//     :93-105 stops at b);
//   * every bin that holds a start beyond the first c + 1 must have more than c intervals open even after all
//     of its own ends: cs_before - ce_through > c.
// Zero-length intervals are taken ((0, 0) ones are inert in the reference and left out of every count; any other
// one changes nothing where more than c intervals are open on both sides of it, and the tests put it nowhere
// else: as an end it may not lie at or before a, as a start its bin must be deep).
// Then the read is bad exactly in front of a and behind b.  Unlike the register classes' screen it takes
// intervals shorter than W anywhere (a read of thousands of intervals nearly always has one), because starts
// inside the tail window and ends inside the head window have bins of their own.
//
// Why a kernel of its own: the workgroup classes' trimming filter (sweep_lds.h) makes two latency-bound
// passes over a read's intervals, four loads in flight per thread, and then still sorts and sweeps what it
// kept (configs[3]: 390 us for 455 MB = 1.2 TB/s), and its one-position bins sit at positions 0 / len, so a
// read that is covered only inside a window keeps its whole piles.  Here a thread issues all of its
// 16 loads before it uses the first (one round trip per read), nothing is sorted, and what the screen cannot
// decide (8 % of configs[3]'s reads) goes to a fallback list for those kernels.
#pragma once
#include "device_common.h"
#include "sweep_lds.h"
#include "sweep_wave.h"

namespace yk {

constexpr int kWsT = 512;        // threads per workgroup = bins
constexpr int kWsR = 16;         // intervals per thread and chunk: a read of <= 8192 intervals is loaded once
constexpr int kWsW = 128;        // window positions on either side
constexpr int kWsNB = kWsT - 2 * kWsW; // coarse blocks (a power of two)
constexpr int kWsBins = kWsT;

// The screen of ONE read by the whole workgroup (kWsT threads): true = decided, its regions and count written.
// tab: kWsBins * 4 words, red: NW x 4, sc: NW + 1 words of LDS; ends with a barrier.
// (o, n, len: the read's first interval, its intervals, its length — the persistent kernel has them before the turn starts)
// A thread takes its intervals two at a time (16-byte loads: pair P = tid + kWsT * j holds intervals 2P and 2P + 1, the load
// clamped to the read's last pair — as the register classes' screen does, sweep_wave.h).
__device__ __forceinline__ bool screen_wg_read(const SweepArgs &a, u32 r, u64 o, u32 n, u32 len, u32 *tab, u32 (*red)[4], u32 *sc)
{
    constexpr int T = kWsT, R = kWsR, W = kWsW, NW = T / 64;
    constexpr u32 kEnd = 1u << 16, kField = kEnd - 1u;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    uint4 *bins = reinterpret_cast<uint4 *>(tab);
    char *tb = reinterpret_cast<char *>(tab);
    const uint2 *iv = a.iv + o;
    const u32 chunks = (n + (u32)(T * R) - 1u) / (u32)(T * R);
    bool fallback = n < 2u || len > kMaxKeyPos;

    // ---- the read's smallest start, largest end, largest start and shortest interval (signed)
    uint4 v[R / 2];
    u32 smin = 0xFFFFFFFFu, emax = 0, smax = 0;
    i32 tmin = 0x7FFFFFFF;
    if (!fallback) {
        for (u32 ch = 0; ch < chunks; ch++) {
            const u32 base = ch * (u32)(T * R / 2) + tid; // (pairs)
#pragma unroll
            for (int j = 0; j < R / 2; j++) // (slots beyond the read: copies of its last two intervals)
                v[j] = *reinterpret_cast<const uint4 *>(iv + min(2u * (base + (u32)(j * T)), n - 2u));
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                smin = min(smin, min(v[j].y != 0u ? v[j].x : 0xFFFFFFFFu, v[j].w != 0u ? v[j].z : 0xFFFFFFFFu)); // ((0, 0) intervals are inert: left out)
                smax = max(smax, max(v[j].x, v[j].z));
                emax = max(emax, max(v[j].y, v[j].w));
                tmin = min(tmin, min((i32)(v[j].y - v[j].x), (i32)(v[j].w - v[j].z)));
            }
        }
    }
    smin = wave_min(smin);
    smax = wave_max(smax);
    emax = wave_max(emax);
    const u32 tkey = wave_min((u32)tmin ^ 0x80000000u); // (signed order as unsigned order)
    if (lane == 0) red[wv][0] = smin, red[wv][1] = smax, red[wv][2] = emax, red[wv][3] = tkey;
    // the table starts out zero: 4 * kWsBins words
    for (u32 i = tid; i < (u32)kWsBins; i += T) bins[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    u32 pmin = 0xFFFFFFFFu, pmax = 0, qmax = 0, tn = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        pmin = min(pmin, red[w][0]);
        qmax = max(qmax, red[w][1]);
        pmax = max(pmax, red[w][2]);
        tn = min(tn, red[w][3]);
    }
    const i32 shortest = (i32)(tn ^ 0x80000000u);
    // not plain (a start > its end, a position beyond the read or the key range), or a covered span too
    // short for two windows: the sort's.  (Zero-length intervals are taken: where more than c intervals are
    // open on both sides of one it changes nothing, and the tests below put it nowhere else.)
    fallback = fallback || pmax > len || qmax > kMaxKeyPos || shortest < 0 || pmax - pmin < (u32)(2 * W);

    bool healthy = false;
    u32 ra = 0, rb = 0;
    if (!fallback) { // (uniform)
        const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(kWsNB) + (len != 0 ? 0 : -1);
        const u32 sh = (u32)max(bits, ilog2c(W));
        const u32 span = pmax - pmin, Tt = span - (u32)W;
        const u32 cp = (tid & 3u) * 4u;
        // ---- count: one map for starts and ends
        auto count = [&](u32 s0, u32 e0, bool real) {
            const u32 ds = s0 - pmin, dx = e0 - pmin;
            const u32 is = min(ds, (u32)W) + (ds >> sh) + __builtin_elementwise_sub_sat(ds, Tt);
            const u32 ie = min(dx, (u32)W) + (dx >> sh) + __builtin_elementwise_sub_sat(dx, Tt);
            if (real && e0 != 0u) {
                atomicAdd(reinterpret_cast<u32 *>(tb + ((is << 4) + cp)), 1u);
                atomicAdd(reinterpret_cast<u32 *>(tb + ((ie << 4) + cp)), kEnd);
            }
        };
        for (u32 ch = 0; ch < chunks; ch++) {
            const u32 base = ch * (u32)(T * R / 2) + tid;
            if (chunks > 1u) {
#pragma unroll
                for (int j = 0; j < R / 2; j++)
                    v[j] = *reinterpret_cast<const uint4 *>(iv + min(2u * (base + (u32)(j * T)), n - 2u));
            }
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                const u32 i0 = 2u * (base + (u32)(j * T));
                count(v[j].x, v[j].y, i0 + 1u < n); // (.xy is interval i0 only when i0 + 1 exists too: the clamped last pair)
                count(v[j].z, v[j].w, i0 < n);
            }
        }
        __syncthreads();
        // ---- this thread's bin, in event order: starts | ends << 16
        const uint4 c4 = bins[tid];
        const u32 w = c4.x + c4.y + c4.z + c4.w;
        // ---- windows: thread d < W looks at window position d: the starts at pmin + d (its own bin) and the
        // ends at pmax - d
        u32 f = 0;
        if (tid < (u32)W) {
            const u32 dx = span - tid;
            const u32 it = min((u32)W + (dx >> sh) + (dx - Tt), (u32)(kWsBins - 1));
            const uint4 t4 = bins[it];
            f = (w & kField) | ((t4.x + t4.y + t4.z + t4.w) & (kField << 16));
        }
        // ---- the window scan and the depth scan share their barrier (red[][0..1]; red[] is free again: every wavefront
        // took pmin .. from it in front of the count), the two reductions behind them theirs (red[][2..3]): five barriers a
        // read instead of eleven
        const u32 f_incl = wave_incl_add(f), w_incl = wave_incl_add(w);
        if (lane == 63u) red[wv][0] = f_incl, red[wv][1] = w_incl;
        __syncthreads();
        u32 fbase = 0, ftot = 0, wbase = 0;
#pragma unroll
        for (u32 k = 0; k < (u32)NW; k++) {
            const u32 x = red[k][0], y = red[k][1];
            fbase += k < wv ? x : 0u;
            wbase += k < wv ? y : 0u;
            ftot += x;
        }
        const u32 fex = fbase + f_incl - f, wex = wbase + w_incl - w;
        const i32 F = (i32)(ftot & kField), G = (i32)(ftot >> 16);
        // positions whose running count has not reached c + 1 yet: their number is a - pmin / pmax - b;
        // an end at a head position at or before a spoils the closed form
        const u32 k1 = (u32)min(c + 1, 0x7FFF);
        const u32 run = fex + f;
        u32 notyet = 0;
        bool spoiled = false;
        if (tid < (u32)W) {
            notyet = ((run & kField) < k1 ? 1u : 0u) | ((run >> 16) < k1 ? kEnd : 0u);
            spoiled = (w >> 16) != 0u && (fex & kField) < k1;
        }
        // ---- depth: a bin that holds a start beyond the first c + 1 needs more than c intervals open after
        // all of its own ends
        const i32 cs_ex = (i32)(wex & kField), ce_in = (i32)((wex >> 16) + (w >> 16));
        const bool shallow = (w & kField) != 0u && cs_ex >= (i32)k1 && !(cs_ex - ce_in > c);
        const u32 n_incl = wave_incl_add(notyet), bad_w = wave_or((shallow || spoiled) ? 1u : 0u);
        if (lane == 63u) red[wv][2] = n_incl, red[wv][3] = bad_w;
        __syncthreads();
        u32 ntot = 0, any_bad = 0;
#pragma unroll
        for (u32 k = 0; k < (u32)NW; k++) ntot += red[k][2], any_bad |= red[k][3];
        healthy = any_bad == 0u && F > c && G > c;
        ra = pmin + (ntot & kField);
        rb = pmax - (ntot >> 16);
    } else {
        __syncthreads(); // (red[] / the table are reused by the next read)
    }
    if (tid == 0 && healthy) {
        uint2 *slot = a.stage + (o + 2 * (u64)r);
        u32 g = 0;
        if (ra != 0) slot[g++] = make_uint2(0u, ra);
        if (rb != len) slot[g++] = make_uint2(rb, len);
        a.counts[r] = g;
        if (a.prefilter == 2) atomicAdd(&a.ctr->prefiltered, 1u);
    }
    __syncthreads();
    return healthy;
}

__device__ __forceinline__ bool screen_wg_read(const SweepArgs &a, u32 r, u32 *tab, u32 (*red)[4], u32 *sc)
{
    const u64 o = a.off[r];
    return screen_wg_read(a, r, o, (u32)(a.off[r + 1] - o), a.len[r], tab, red, sc);
}
// SweepArgs.list / list_n: the class list; over_list / over_count: the reads the screen leaves to the sort.
__global__ __launch_bounds__(kWsT) void screen_wg_kernel(SweepArgs a)
{
    constexpr int NW = kWsT / 64;
    __shared__ __attribute__((aligned(16))) u32 tab[kWsBins * 4]; // four copies of every counter (by thread & 3)
    __shared__ u32 red[NW][4];
    __shared__ u32 sc[NW + 1];
    const u32 list_n = *a.list_n;
    for (u32 b = blockIdx.x; b < list_n; b += gridDim.x) { // (uniform)
        const u32 r = a.list[b];
        if (!screen_wg_read(a, r, tab, red, sc) && threadIdx.x == 0) a.over_list[atomicAdd(a.over_count, 1u)] = r;
    }
}

// ---- the screen and its fallback in ONE launch (round 4) ---------------------------------------------------------
// Round 3 ran three kernels one after the other for a workgroup class: the screen, sweep_lds_kernel<256, 8192> over
// the reads it left, sweep_lds_kernel<1024, 32768> over what did not fit there.  On configs[3] the two fallback
// kernels took 61 + 65 us for 4 % of the bytes: each is one latency chain per read (two passes over the intervals,
// the trimming plan, the sort, four sweep passes) with most of the device idle, and the second cannot start before
// the first has ended.  Here a persistent grid does both: a workgroup screens its share of the class list (static
// stride), appends what it cannot decide to a queue in global memory, and when its share is done takes reads off
// that queue — its own and everybody else's — through sweep_lds_read<512, 16384> until every workgroup has finished
// screening and the queue is empty: the fallback reads are sorted WHILE other workgroups still screen, spread over
// every workgroup that has nothing else to do.  What does not fit 16 384 events even after the filter goes to
// over_list for the 1024-thread kernel (launched behind this one; usually nothing).
// Queue: q[] starts out as kQueueEmpty in every slot a read of the class could take (the plan kernel writes the
// marker where it writes the class list), tail = slots handed out, head = slots claimed, done = workgroups that
// finished screening.  A claimed slot beyond tail is waited for until it is filled or `done` says it never will be —
// for a bounded number of looks (below).
constexpr u32 kQueueEmpty = 0xFFFFFFFFu;
constexpr int kWsFbCap = 16384; // events the in-kernel fallback sorts (64 KB of LDS; two workgroups per CU by registers anyway)
#ifndef YK_WG_OCC
#define YK_WG_OCC 4 // wavefronts per SIMD the register budget allows (two workgroups per CU)
#endif
struct ScreenFusedArgs {
    SweepArgs sweep;  // list / list_n: the class; over_list / over_count: beyond kWsFbCap; rej_*: degenerate reads
    u32 *q;           // the queue's slots
    u32 *tail, *head, *done;
};
__global__ __launch_bounds__(kWsT, YK_WG_OCC) void screen_wg_fused_kernel(ScreenFusedArgs f)
{
    constexpr int NW = kWsT / 64;
    __shared__ __attribute__((aligned(16))) u32 tab[kWsBins * 4];
    __shared__ u32 red[NW][4];
    __shared__ u32 sc[NW + 1];
    __shared__ u32 keys[kWsFbCap];
    __shared__ u32 s_next;
    const SweepArgs &a = f.sweep;
    const u32 tid = threadIdx.x;
    const u32 list_n = *a.list_n;
    // ---- the share, then the queue: two loops.  Round 5 measured three other shapes of this kernel on configs[3]
    // (profiles/r05/e_*, f_*; this form: 0.221-0.229 ms):
    //  * the next read's list entry, extent and length asked for a turn ahead: 0.232-0.234 — and, asked through VGPR indices
    //    and kept per lane so that the compiler does not wait for them on the spot (a turn then waits for ONE round trip, its
    //    intervals', instead of three): 0.220-0.225 against 0.220-0.223, profiles/r05/p_* (the round trips it saves are hidden
    //    by the CU's other workgroup already);
    //  * one loop, the queue looked at between the turns so that a fallback read (a ~50 us chain) starts while others still
    //    screen: 0.313-0.316 — the sort's registers and the screen's live side by side (44 bytes of scratch per thread), and
    //    a workgroup that takes a fallback read early delays its own share by as much as it saves the tail;
    //  * the next read's first 64 KB staged in LDS a turn ahead (global_load_lds_dwordx4 into the sort's idle key array,
    //    LDS-only barriers so that the loads stay in flight across the turn): 0.241-0.243 — a second set of registers
    //    for them does not fit 128 VGPRs, and through LDS the copy costs what the overlap gains: two workgroups per CU
    //    already alternate their load and count phases.
    for (u32 b = blockIdx.x; b < list_n; b += gridDim.x) { // (uniform)
        const u32 r = a.list[b];
        if (!screen_wg_read(a, r, tab, red, sc) && tid == 0)
            __hip_atomic_store(&f.q[atomicAdd(f.tail, 1u)], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __threadfence(); // this workgroup's appends before its "done"
        atomicAdd(f.done, 1u);
    }
    LaneConst lc;
#pragma unroll
    for (int i = 0; i < 6; i++) lc.k[i] = (tid & (1u << i)) ? 0xFFFFFFFFu : 0u;
    lc.k[6] = 0;
    lc.addr32 = ((tid & 63u) ^ 32u) << 2;
    // A claimed slot beyond `tail` is waited for until it is filled or `done` says it never will be.  That wait needs
    // the workgroups it waits for to RUN: the grid is sized to be resident as a whole, but a second process on the
    // device, a CU mask or anything else that holds LDS / wave slots can leave some of them undispatched behind the
    // spinning ones.  So the wait is bounded (kFusedPolls looks, ~10 ms — a healthy launch is over in a
    // fraction of one): a workgroup that runs out raises Counters::fused_gave_up and leaves, every other one then
    // leaves at its next look, the kernel ends, and the engine runs the batch again down the three-launch chain
    // (engine.hip: fused_off), which waits for nothing.  (Claims by compare-and-swap — none is lost when a workgroup
    // leaves — were tried first: 512 workgroups retrying on one address took the pass from 0.32 to 1.59 ms.)
    constexpr u32 kFusedPolls = 1u << 14;
    for (;;) {
        if (tid == 0) {
            const u32 idx = atomicAdd(f.head, 1u);
            u32 r = kQueueEmpty, polls = 0;
            for (;;) {
                if (idx < __hip_atomic_load(f.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    // handed out: its writer (a running wavefront, one store behind its tail increment) fills it
                    while ((r = __hip_atomic_load(&f.q[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kQueueEmpty)
                        __builtin_amdgcn_s_sleep(2);
                    break;
                }
                if (__hip_atomic_load(f.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x) {
                    // every workgroup has screened its share: the tail is final
                    if (idx < __hip_atomic_load(f.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                    break;
                }
                if (++polls > kFusedPolls || __hip_atomic_load(&a.ctr->fused_gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(&a.ctr->fused_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break; // (r = kQueueEmpty: this workgroup leaves; the batch is run again)
                }
                __builtin_amdgcn_s_sleep(8);
            }
            s_next = r;
        }
        __syncthreads();
        const u32 r = s_next;
        if (r == kQueueEmpty) break; // (uniform)
        sweep_lds_read<kWsT, kWsFbCap>(a, r, keys, sc, lc);
        __syncthreads(); // keys / sc / s_next reused
    }
}

} // namespace yk
