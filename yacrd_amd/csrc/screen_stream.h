// screen_stream.h — the healthy-read screen for reads of 513 .. 16 384 intervals by ONE WAVEFRONT per
// read: the read streamed through registers twice, no workgroup barrier anywhere (round 6).
//
// The rule, the position map and the emulation are screen_wg.h's (tests/formulation.py::unified_screen_regions; reference
// semantics src/stack.rs:61-139): one map for starts and ends,
//     idx(x) = min(dx, W) + (dx >> sh) + max(dx - T, 0),    dx = x - pmin,  T = (pmax - pmin) - W,  2^sh >= W,
// a bin per position in the first W and the last W positions of the covered span, coarse blocks in between; a = the
// position where the starts counted upwards from pmin reach c + 1 (inside the head window, no end at or before it), b =
// the position where the ends counted downwards from pmax reach c + 1 (inside the tail window), and every bin that holds
// a start beyond the first c + 1 must have more than c intervals open after all of its own ends.  Then the read is bad
// exactly in front of a and behind b.
//
// Why another shape of the same screen.  screen_wg.h gives a read to a 512-thread workgroup that holds it in registers:
// one pass over memory, but a turn is a chain — list entry, extent, intervals, then eleven barriers' worth of table work
// — and two such workgroups fit a CU (128 VGPRs, 72 KB of LDS with the in-kernel fallback's keys): configs[3] (10 000
// reads of 5 000 .. 16 384 intervals) ran at 10 us per read and workgroup, 0.194 ms for the screening alone
// (profiles/r06/a_cfg3_experiments.log: the fused kernel without its fallback), 2.3 TB/s, with the CU's VALU a third
// busy and its LDS pipe less.  Here a read is one wavefront's: pass 1 streams it for the smallest start / largest end,
// pass 2 streams it again (from the Infinity Cache: 46 KB a read, read microseconds before) and counts into a 4 KB table
// of the wavefront's own, one 64-lane scan decides.  24 wavefronts per CU, each a chain of its own, no barrier: the
// memory system sees 24 independent streams per CU instead of 2.  The price is the second pass — `traffic` is up to
// twice the algorithmic bytes — which buys the absence of everything else.
// What the screen cannot decide is appended to over_list; the engine runs screen_wg_fused_kernel over THAT list (a
// fiftieth of the class on the generator's reads), which sorts it on its persistent grid as before.
#pragma once
#include "device_common.h"
#include "sweep_wave.h"

namespace yk {

constexpr int kSsW = 128;                      // window positions on either side
constexpr int kSsNB = 256;                     // coarse blocks (a power of two)
constexpr int kSsBins = 2 * kSsW + kSsNB;      // 512: eight per lane
constexpr int kSsCopies = 2;                   // counters per bin (by lane & 1: a read's hot bins are hit by many lanes at once)
constexpr int kSsTabWords = kSsBins * kSsCopies;
constexpr int kSsPairs = 8;                    // 16-byte pair loads per lane and chunk: 1024 intervals, 8 KB in flight per wavefront
#ifndef YK_SS_OCC
#define YK_SS_OCC 6
#endif

// true = decided: the read's regions and count are written.  tab: kSsTabWords words of LDS, this wavefront's own.
__device__ __forceinline__ bool screen_stream_read(const SweepArgs &a, u32 r, u32 *tab)
{
    constexpr int W = kSsW, P = kSsPairs;
    constexpr u32 kEnd = 1u << 16, kField = kEnd - 1u;
    static_assert(kSsBins == 8 * 64 && kSsCopies == 2, "a lane owns eight consecutive bins = sixteen consecutive words");
    const u32 lane = lane_id();
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    const ulonglong2 oo = load_extent(a.off + r);
    const u64 o = oo.x;
    const u32 n = (u32)(oo.y - oo.x), len = a.len[r];
    if (n < 2u || len > kMaxKeyPos) return false; // (uniform)
    const uint2 *iv = a.iv + o;
    const u32 last2 = n - 2u;
    const u32 chunks = ((n + 1u) / 2u + (u32)(64 * P) - 1u) / (u32)(64 * P);

    // ---- pass 1: smallest start ((0, 0) intervals are inert: left out), largest start, largest end, shortest interval
    u32 smin = 0xFFFFFFFFu, emax = 0, smax = 0;
    i32 tmin = 0x7FFFFFFF;
    for (u32 ch = 0; ch < chunks; ch++) { // (uniform)
        uint4 v[P];
#pragma unroll
        for (int j = 0; j < P; j++) // (pairs beyond the read: copies of its last two intervals)
            v[j] = load_pair(iv + min(2u * (ch * (u32)(64 * P) + (u32)j * 64u + lane), last2));
#pragma unroll
        for (int j = 0; j < P; j++) {
            smin = min(smin, min(v[j].y != 0u ? v[j].x : 0xFFFFFFFFu, v[j].w != 0u ? v[j].z : 0xFFFFFFFFu));
            smax = max(smax, max(v[j].x, v[j].z));
            emax = max(emax, max(v[j].y, v[j].w));
            tmin = min(tmin, min((i32)(v[j].y - v[j].x), (i32)(v[j].w - v[j].z)));
        }
    }
    // the table starts out zero (sixteen words per lane)
    {
        uint4 *t4 = reinterpret_cast<uint4 *>(tab) + lane * 4u;
#pragma unroll
        for (int q = 0; q < 4; q++) t4[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    const u32 pmin = wave_min(smin), qmax = wave_max(smax), pmax = wave_max(emax);
    const i32 shortest = (i32)(wave_min((u32)tmin ^ 0x80000000u) ^ 0x80000000u);
    // not plain (a start > its end, a position beyond the read or the key range), or a covered span too short for two
    // windows: the sort's
    if (pmax > len || qmax > kMaxKeyPos || shortest < 0 || pmin > pmax || pmax - pmin < (u32)(2 * W)) return false; // (uniform)
    wave_lds_sync();

    const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(kSsNB) + (len != 0 ? 0 : -1);
    const u32 sh = (u32)max(bits, ilog2c(W));
    const u32 span = pmax - pmin, Tt = span - (u32)W;
    char *tb = reinterpret_cast<char *>(tab);
    const u32 cp = (lane & 1u) * 4u;
    // ---- pass 2: count, one map for starts and ends
    auto count = [&](u32 s0, u32 e0, bool real) {
        const u32 ds = s0 - pmin, dx = e0 - pmin;
        const u32 is = min(ds, (u32)W) + (ds >> sh) + __builtin_elementwise_sub_sat(ds, Tt);
        const u32 ie = min(dx, (u32)W) + (dx >> sh) + __builtin_elementwise_sub_sat(dx, Tt);
        if (real && e0 != 0u) {
            atomicAdd(reinterpret_cast<u32 *>(tb + ((is << 3) + cp)), 1u);
            atomicAdd(reinterpret_cast<u32 *>(tb + ((ie << 3) + cp)), kEnd);
        }
    };
    for (u32 ch = 0; ch < chunks; ch++) { // (uniform)
        uint4 v[P];
#pragma unroll
        for (int j = 0; j < P; j++)
            v[j] = load_pair(iv + min(2u * (ch * (u32)(64 * P) + (u32)j * 64u + lane), last2));
#pragma unroll
        for (int j = 0; j < P; j++) {
            const u32 i0 = 2u * (ch * (u32)(64 * P) + (u32)j * 64u + lane);
            count(v[j].x, v[j].y, i0 + 1u < n); // (.xy is interval i0 only when i0 + 1 exists too: the clamped last pair)
            count(v[j].z, v[j].w, i0 < n);
        }
    }
    wave_lds_sync();

    // ---- this lane's eight bins, in event order: starts | ends << 16
    u32 w[8];
    {
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tab) + lane * 4u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = t4[q];
            w[2 * q] = x.x + x.y;
            w[2 * q + 1] = x.z + x.w;
        }
    }
    // (the ends of bin 0 are zero-length intervals at pmin: inert when c >= 1 and a regular interval starts there too —
    //  screen_wg.h, tests/formulation.py::drop_inert_at_pmin — and forgotten, starts and ends)
    if (lane == 0 && c >= 1 && (w[0] >> 16) != 0u && (w[0] & kField) > (w[0] >> 16)) w[0] -= (w[0] >> 16) * (kEnd + 1u);
    u32 mine = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) mine += w[j];
    const u32 incl = wave_incl_add(mine);
    u32 wex = incl - mine; // starts | ends << 16 of every bin in front of this lane's
    // the head window is bins 0 .. W - 1 = lanes 0 .. 15: F = its starts
    const i32 F = (i32)((u32)__shfl((int)wex, W / 8, 64) & kField);
    const u32 k1 = (u32)min(c + 1, 0x7FFF);
    u32 notyet = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 starts = w[j] & kField, ends = w[j] >> 16;
        const i32 cs_ex = (i32)(wex & kField), ce_in = (i32)((wex >> 16) + ends);
        // depth: a bin that holds a start beyond the first c + 1 needs more than c intervals open after all of its own ends
        bad |= starts != 0u && cs_ex >= (i32)k1 && !(cs_ex - ce_in > c);
        if (lane < (u32)(W / 8)) { // a head-window position: its running count of starts; an end at or before a spoils the closed form
            notyet += (u32)cs_ex + starts < k1 ? 1u : 0u;
            bad |= ends != 0u && (u32)cs_ex < k1;
        }
        wex += w[j];
    }
    // ---- the tail window, by lanes 16 .. 31: position pmax - d, d = 8 (lane - 16) + j, its ends
    u32 t[8];
    u32 tsum = 0;
    const bool tail_lane = lane >= (u32)(W / 8) && lane < (u32)(W / 4);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 d = (tail_lane ? (lane - (u32)(W / 8)) * 8u : 0u) + (u32)j;
        const u32 dx = span - d;
        const u32 it = min((u32)W + (dx >> sh) + (dx - Tt), (u32)(kSsBins - 1));
        const uint2 x = *reinterpret_cast<const uint2 *>(tab + it * 2u);
        t[j] = tail_lane ? (x.x + x.y) >> 16 : 0u;
        tsum += t[j];
    }
    const u32 tincl = wave_incl_add(tsum);
    const i32 G = (i32)(u32)__shfl((int)tincl, W / 4 - 1, 64);
    u32 run = tincl - tsum;
    u32 notyet_e = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        run += t[j];
        notyet_e += (tail_lane && run < k1) ? 1u : 0u;
    }
    const u32 ntot = (u32)__shfl((int)wave_incl_add(notyet | (notyet_e << 16)), 63, 64);
    const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    const bool healthy = !any_bad && F > c && G > c;
    if (healthy && lane == 0) {
        const u32 ra = pmin + (ntot & kField), rb = pmax - (ntot >> 16);
        uint2 *slot = a.stage + (o + 2 * (u64)r);
        u32 g = 0;
        if (ra != 0) slot[g++] = make_uint2(0u, ra);
        if (rb != len) slot[g++] = make_uint2(rb, len);
        a.counts[r] = g;
        if (a.prefilter == 2) atomicAdd(&a.ctr->prefiltered, 1u);
    }
    return healthy;
}

// SweepArgs.list / list_n: the class list (first: its first entry this launch covers); over_list / over_count: the reads
// the screen leaves to screen_wg_fused_kernel.  One one-wavefront workgroup per list entry where the grid allows: the
// dispatcher deals the reads out as wavefronts retire (their sizes differ by 3 x inside a class).
__global__ __launch_bounds__(64, YK_SS_OCC) void screen_stream_kernel(SweepArgs a)
{
    __shared__ __attribute__((aligned(16))) u32 tab[kSsTabWords];
    const u32 list_n = *a.list_n;
    // (a grid sized from a prediction may be shorter or longer than the class: the stride covers the one, the test the other)
    for (u32 b = a.first + blockIdx.x; b < list_n; b += gridDim.x) { // (uniform)
        const u32 r = a.list[b];
        if (!screen_stream_read(a, r, tab) && lane_id() == 0) a.over_list[atomicAdd(a.over_count, 1u)] = r;
        wave_lds_sync(); // (the next read zeroes the table)
    }
}

} // namespace yk
