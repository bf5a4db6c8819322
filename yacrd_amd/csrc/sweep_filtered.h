// sweep_filtered.h — the follow-on step's FILTERED exact sweep (round 5).
//
// tests/formulation.py::filtered_sweep_regions is the emulation (fuzzed against the oracle, exhaustive over small
// multisets).  The reads the healthy-read screen (sweep_wave.h) defers are the reads yacrd looks for: nearly all of
// them healthy at both ENDS and low somewhere inside.  Sorting such a read whole — 2n keys on 64 lanes, 847 VALU
// instructions per read (profiles/r04) — sorts ~300 events that cannot matter.  Here the screen's own table is built
// once more (healthy_screen: W one-position bins at either end, LANES coarse blocks in between) and read differently:
//   * F > c and G > c: a = the (c+1)-th smallest start, b = the (c+1)-th largest end lie in their windows; the
//     reference (src/stack.rs:83-113) gives (0, a) in front and (b, len) behind, as for a healthy read;
//   * D_i = F + (coarse starts - coarse ends of the blocks before i) is the exact depth on entry to block i and
//     D_i - E_i the least depth any event of the block sees.  D_i - E_i > c: the block is SAFE — none of its starts
//     is low (:83), all of its ends are flagged (:77-79).  A low start therefore lies in an unsafe block, and the
//     flagged end the reference pairs it with (the last one in front of it) lies in an unsafe block too or is the
//     largest end of the nearest block in front that holds an end;
//   * kept: the unsafe blocks and, for each, the nearest block in front that holds an end — <= 128 events or the
//     read is sorted whole after all.  The kept events are compacted through LDS (slots from the blocks' scanned
//     counts), sorted (4 or 8 keys per lane) and swept with their TRUE depths: the block's D_i, carried as a
//     correction per block, plus the kept events in front;
//   * a run of low starts still open when the keys end is closed by the tail (G > c: the (c+1)-th largest end is
//     flagged behind every start).
// Two reads per wavefront on 32-lane groups (129..256 intervals), four on 16-lane groups (<= 128): every scan and
// every sort step is shared.  Plain reads whose intervals are all at least W long only (the screen's own test);
// everything else, and every guard that fails, leaves counts[r] marked and returns false: the caller sorts the read.
#pragma once
#include "sweep_wave.h"

namespace yk {

constexpr int kFilteredCap = 128;             // kept events per read (the tail window's bins hold the keys: W * 16 bytes)
constexpr u32 kFilteredPad = 0xFFFFFFFFu;
static_assert(kScreenWindow * 4 >= kFilteredCap, "the keys live in the tail window's bins");

// One deferred read per group of LANES lanes: v / real0 / real1 / pmin / pmax as for healthy_screen (every slot below n
// real, the others copies), `want`: this group holds a plain read of >= 2 intervals, each at least W long (uniform in
// the group).  r / o / len: the read, its first interval's index, its length.  True (uniform in the group): the
// read's regions and their count are written.
template <int LANES, int WPB, int TABW = kScreenTabWords>
__device__ __forceinline__ bool filtered_group_sweep(const SweepArgs &a, const uint4 (&v)[4], const bool (&real0)[4], const bool (&real1)[4],
                                                     u32 r, u64 o, u32 n, u32 len, i32 c, u32 pmin, u32 pmax, bool want, const LaneConst &lc)
{
    constexpr int NB = LANES, W = kScreenWindow, NBIN = 2 * W + NB, K = kFilteredCap / LANES;
    constexpr u32 kEnd = 1u << 10, kField = kEnd - 1u;
    static_assert(LANES == 16 || LANES == 32, "group masks are 32-bit");
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1), grp = lane / (u32)LANES;
    const int last_addr = (int)((lane | (u32)(LANES - 1)) << 2);
    const u32 gshift = lane & (u32)(64 - LANES);
    constexpr u32 gmask = LANES == 32 ? 0xFFFFFFFFu : 0xFFFFu;
    auto to_group = [&](u32 x) { return (u32)__builtin_amdgcn_ds_bpermute(last_addr, (int)x); }; // the last lane's value
    auto group_bits = [&](bool b) { return (u32)(__builtin_amdgcn_ballot_w64(b) >> gshift) & gmask; };
    static_assert((64 / LANES) * NBIN * 4 <= TABW, "scratch");
    u32 *tab = wave_screen_scratch<WPB, TABW>() + grp * (u32)(NBIN * 4);
    uint4 *bins = reinterpret_cast<uint4 *>(tab);
    u32 *info = tab;                              // [2 * NB] in the head window's bins: slot cursor, depth correction per block
    u32 *keys = tab + (u32)((W + NB) * 4);        // [kFilteredCap] in the tail window's bins
    const bool last = lig == (u32)(LANES - 1);

    // ---- the screen's table, its order statistics and totals (valid in the group's last lane)
    HealthyRead hr;
    bool r0[4], r1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r0[j] = real0[j] && want, r1[j] = real1[j] && want;
    const bool healthy = healthy_screen<LANES, WPB, false, TABW>(v, r0, r1, len, c, pmin, pmax, hr);
    const u32 hF = to_group((u32)min(max(hr.F, 0), 1023) | ((hr.G > c && want && (i32)n > c) ? 0x400u : 0u) | (healthy ? 0x800u : 0u));
    const i32 F = (i32)(hF & 1023u);
    bool ok = (hF & 0x400u) != 0u && F > c;
    const bool is_healthy = ok && (hF & 0x800u) != 0u; // (a read another build deferred: nothing inside)
    if (__builtin_amdgcn_ballot_w64(ok) == 0) return false; // (uniform in the wavefront)
    const u32 ga = to_group(hr.a), gb = to_group(hr.b);

    // ---- the coarse blocks: entry depths, unsafe blocks, the blocks kept
    const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(NB) + (len != 0 ? 0 : -1);
    const u32 sh = (u32)max(bits, ilog2c(W));
    const u32 span = pmax - pmin, T = span - (u32)W, iT = T >> sh;
    const uint4 c4 = bins[(u32)W + lig];
    const u32 w = (c4.x + c4.y + c4.z + c4.w) & ((kField << 10) | kField);
    const i32 S = (i32)(w & kField), E = (i32)(w >> 10);
    const u32 wincl = gscan_add<LANES>(w), wex = wincl - w;
    const i32 D = F + (i32)(wex & kField) - (i32)(wex >> 10);
    const bool inside = lig <= iT; // (the blocks behind hold tail-window ends only: b answers for them)
    const bool unsafe = ok && !is_healthy && inside && S + E > 0 && D - E <= c;
    const u32 mu = group_bits(unsafe), me = group_bits(inside && E > 0);
    const u32 higher = ((mu | me) >> lig) >> 1;
    const bool kept = unsafe || (ok && inside && E > 0 && higher != 0u && (((mu >> lig) >> 1) >> __builtin_ctz(higher | 0x80000000u)) & 1u);
    const u32 km = group_bits(kept);
    const u32 pack = kept ? (u32)(S + E) + ((u32)(S - E) << 16) : 0u; // count | net depth change (two's complement)
    const u32 pincl = gscan_add<LANES>(pack), pex = pincl - pack;
    const u32 m = to_group(pincl) & 0xFFFFu;
    ok = ok && m <= (u32)kFilteredCap;
    if (__builtin_amdgcn_ballot_w64(ok) == 0) return false;
    wave_lds_sync(); // (every lane has read the table)
    *reinterpret_cast<uint2 *>(info + 2u * lig) = make_uint2(pex & 0xFFFFu, (u32)(D - ((i32)pex >> 16)));
    {
        uint4 *kp = reinterpret_cast<uint4 *>(keys + lig * (u32)K);
#pragma unroll
        for (int q = 0; q < K / 4; q++) kp[q] = make_uint4(kFilteredPad, kFilteredPad, kFilteredPad, kFilteredPad);
    }
    wave_lds_sync();

    // ---- the kept events, to the slots of their blocks
    auto put = [&](u32 s, u32 e, bool real) {
        const u32 ds = s - pmin, dx = e - pmin;
        const u32 bs = ds >> sh, be = dx >> sh;
        if (real && ok && ds >= (u32)W && ((km >> bs) & 1u)) keys[atomicAdd(info + 2u * bs, 1u)] = (s << kKeyShift) | 3u;
        if (real && ok && dx <= T && ((km >> be) & 1u)) keys[atomicAdd(info + 2u * be, 1u)] = e << kKeyShift;
    };
#pragma unroll
    for (int j = 0; j < 4; j++) {
        put(v[j].x, v[j].y, r0[j]);
        put(v[j].z, v[j].w, r1[j]);
    }
    wave_lds_sync();
    u32 x[K];
    {
        const uint4 *kp = reinterpret_cast<const uint4 *>(keys + lig * (u32)K);
#pragma unroll
        for (int q = 0; q < K / 4; q++) {
            const uint4 t = kp[q];
            x[4 * q] = t.x, x[4 * q + 1] = t.y, x[4 * q + 2] = t.z, x[4 * q + 3] = t.w;
        }
    }
    bitonic_sort<LANES, K, 2, 0>(x, lc);

    // ---- the sweep: depth in front of a key = the kept keys in front (starts - ends) + its block's correction
    i32 corr[K];
    u32 dl = 0;
#pragma unroll
    for (int q = 0; q < K; q++) {
        const bool real = x[q] != kFilteredPad;
        const u32 blk = real ? (((x[q] >> kKeyShift) - pmin) >> sh) : 0u;
        corr[q] = (i32)info[2u * blk + 1u];
        dl += real ? ((x[q] & 1u) << 1) - 1u : 0u;
    }
    const i32 depth_in = (i32)(gscan_add<LANES>(dl) - dl);
    // last flagged end / last low start of the lane (keys ascend: last = max); 0 = none (no end lies at position 0)
    u32 mf = 0, ml = 0;
    {
        i32 d = depth_in;
#pragma unroll
        for (int q = 0; q < K; q++) {
            const u32 key = x[q];
            const bool real = key != kFilteredPad, is_s = (key & 1u) != 0u, gt = d + corr[q] > c;
            mf = (real && !is_s && gt) ? key : mf;
            ml = (real && is_s && !gt) ? key : ml;
            d += real ? (is_s ? 1 : -1) : 0;
        }
    }
    const u32 mf_incl = gscan_max<LANES>(mf), ml_incl = gscan_max<LANES>(ml);
    const u32 mf_in = gshift_up1<LANES>(mf_incl), ml_in = gshift_up1<LANES>(ml_incl);
    u32 cnt = 0, fb = 0, fe = 0;
    bool orphan = false; // a low start with no flagged end in front of it (cannot happen with F > c: the caller sorts the read)
    {
        u32 tc = mf_in, cml = ml_in;
        i32 d = depth_in;
#pragma unroll
        for (int q = 0; q < K; q++) {
            const u32 key = x[q];
            const bool real = key != kFilteredPad, is_s = (key & 1u) != 0u, gt = d + corr[q] > c;
            const bool fl = real && !is_s && gt, low = real && is_s && !gt;
            const bool close = fl && cml > tc;
            cnt += close ? 1u : 0u;
            fb = close ? tc : fb;
            fe = close ? cml : fe;
            orphan |= low && tc == 0u;
            tc = fl ? key : tc;
            cml = low ? key : cml;
            d += real ? (is_s ? 1 : -1) : 0;
        }
    }
    ok = ok && group_bits(orphan) == 0u;
    if (__builtin_amdgcn_ballot_w64(ok) == 0) return false;
    // ---- regions out: (0, a), the closed runs, the run the tail closes, (b, len)
    const u32 g0 = ga != 0u ? 1u : 0u;
    const u32 cincl = gscan_add<LANES>(cnt);
    if (ok && (cnt != 0u || last)) {
        uint2 *slot = a.stage + (o + 2 * (u64)r);
        u32 pos = g0 + cincl - cnt;
        if (cnt == 1u) {
            slot[pos] = make_uint2(fb >> kKeyShift, fe >> kKeyShift);
        } else if (cnt > 1u) { // several runs close inside one lane: replay it
            u32 tc = mf_in, cml = ml_in;
            i32 d = depth_in;
#pragma unroll
            for (int q = 0; q < K; q++) {
                const u32 key = x[q];
                const bool real = key != kFilteredPad, is_s = (key & 1u) != 0u, gt = d + corr[q] > c;
                const bool fl = real && !is_s && gt, low = real && is_s && !gt;
                if (fl && cml > tc) slot[pos++] = make_uint2(tc >> kKeyShift, cml >> kKeyShift);
                tc = fl ? key : tc;
                cml = low ? key : cml;
                d += real ? (is_s ? 1 : -1) : 0;
            }
        }
        if (last) {
            u32 g = g0 + cincl;
            if (ga != 0u) slot[0] = make_uint2(0u, ga);
            if (ml_incl > mf_incl) slot[g++] = make_uint2(mf_incl >> kKeyShift, ml_incl >> kKeyShift);
            if (gb != len) slot[g++] = make_uint2(gb, len);
            a.counts[r] = g;
        }
    }
    return ok;
}

} // namespace yk
