// sweep_wave.h — small classes (<= 1024 events): one, two or four reads per wavefront, sorted in
// registers; a coverage pre-filter in front of the sort for the 16-keys-per-lane classes.
//
// Same event formulation as sweep_lds.h (reference src/stack.rs:61-139 for regular reads), but
// the dominant cost — sorting the 2n event keys — runs as a bitonic network over VGPRs:
//   * a lane holds K consecutive keys of the sequence (element index = lane*K + r), so strides
//     < K are register-to-register min/max and the post-sort scans are lane-sequential;
//   * strides >= K exchange between lanes: DPP moves for lane xor 1, 2, 8 (VALU only),
//     ds_swizzle for xor 4, 16 and ds_bpermute for xor 32 (the LDS crossbar is otherwise idle,
//     no LDS memory is touched), each followed by ONE v_med3_u32: med3(x, partner, 0) = min,
//     med3(x, partner, ~0) = max, the third operand being a per-lane constant that encodes
//     "upper lane of the pair" xor "descending block".
// The sort touches no LDS memory and no barrier: a 256-thread workgroup is four independent
// wavefronts (the pre-filter keeps a small histogram and its survivors in LDS, per wavefront).
// Pads are end-like keys (0xFFFFFFFE) and depth compares are signed, so nothing after the sort
// needs a validity mask (a read without intervals falls out as [(0,len)] on its own).
// Reads of <= 128 intervals use 16-lane groups: four reads per wavefront (see sweep_group_read).
#pragma once
#include <type_traits>

#include "device_common.h"

namespace yk {

constexpr u32 kPadKey = 0xFFFFFFFEu;

__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c)
{
    return max(min(a, b), min(max(a, b), c)); // -> v_med3_u32
}

// DPP controls (gfx9): quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_ror:8, row_shr:n, wave_shr:1
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR8 = 0x128;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138, DPP_BCAST15 = 0x142,
              DPP_BCAST31 = 0x143;

// value of lane ^ D.  XM (cross-lane mode): 0 = DPP for xor 1/2/8, LDS crossbar for 4/16/32;
// 1 = LDS crossbar (ds_swizzle / ds_bpermute) for every stride (fewest VALU instructions).
template <int D, int XM>
__device__ __forceinline__ u32 lane_xor(u32 x, u32 bperm_addr32)
{
    if constexpr (D == 32) return (u32)__builtin_amdgcn_ds_bpermute((int)bperm_addr32, (int)x);
    else if constexpr (XM == 1 || D == 4 || D == 16)
        return (u32)__builtin_amdgcn_ds_swizzle((int)x, (D << 10) | 0x1F);
    else if constexpr (D == 1) return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_XOR1, 0xF, 0xF, false);
    else if constexpr (D == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_XOR2, 0xF, 0xF, false);
    else return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_ROR8, 0xF, 0xF, false);
}

// wave64 inclusive scans on DPP (row_shr 1/2/4/8, row_bcast 15/31); identity 0
#define YK_DPP0(v, ctrl, rm) (u32) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rm, 0xF, true)
__device__ __forceinline__ u32 wscan_add(u32 v)
{
    v += YK_DPP0(v, DPP_ROW_SHR1, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR2, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR4, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR8, 0xF);
    v += YK_DPP0(v, DPP_BCAST15, 0xA);
    v += YK_DPP0(v, DPP_BCAST31, 0xC);
    return v;
}
__device__ __forceinline__ u32 wscan_max(u32 v)
{
    v = max(v, YK_DPP0(v, DPP_ROW_SHR1, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR2, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR4, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR8, 0xF));
    v = max(v, YK_DPP0(v, DPP_BCAST15, 0xA));
    v = max(v, YK_DPP0(v, DPP_BCAST31, 0xC));
    return v;
}
__device__ __forceinline__ u32 wshift_up1(u32 v) { return YK_DPP0(v, DPP_WAVE_SHR1, 0xF); }

// 16-lane (DPP row) inclusive scans: four reads per wavefront, one per row
__device__ __forceinline__ u32 rscan_add(u32 v)
{
    v += YK_DPP0(v, DPP_ROW_SHR1, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR2, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR4, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR8, 0xF);
    return v;
}
__device__ __forceinline__ u32 rscan_max(u32 v)
{
    v = max(v, YK_DPP0(v, DPP_ROW_SHR1, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR2, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR4, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR8, 0xF));
    return v;
}
__device__ __forceinline__ u32 rscan_min(u32 v) // identity ~0: shifted-in lanes must not win
{
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_ROW_SHR1, 0xF, 0xF, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_ROW_SHR2, 0xF, 0xF, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_ROW_SHR4, 0xF, 0xF, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_ROW_SHR8, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ u32 rshift_up1(u32 v) { return YK_DPP0(v, DPP_ROW_SHR1, 0xF); }

struct LaneConst {
    u32 k[7];   // k[i] = (lane & (1<<i)) ? ~0u : 0u for i < 6; k[6] = 0
    u32 addr32; // byte address of lane ^ 32 for ds_bpermute
};

__device__ __forceinline__ LaneConst make_lane_const(u32 lane)
{
    LaneConst lc;
#pragma unroll
    for (int i = 0; i < 6; i++) lc.k[i] = (lane & (1u << i)) ? 0xFFFFFFFFu : 0u;
    lc.k[6] = 0;
    lc.addr32 = (lane ^ 32u) << 2;
    return lc;
}

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// ---- bitonic sort of LANES*K keys held as x[K] per lane, element index = lane_in_group*K + r --
// LANES = 64: one sequence per wavefront; LANES = 16: four independent sequences, one per DPP row.
template <int LANES, int K, int M, int J, int XM>
__device__ __forceinline__ void bitonic_step(u32 (&x)[K], const LaneConst &lc)
{
    constexpr int P = LANES * K;
    constexpr bool lane_dir = (M >= K) && (M < P); // direction bit lives in the lane id
    const u32 dirm = lane_dir ? lc.k[ilog2c(M / K)] : 0u;
    if constexpr (J >= K) { // partner in another lane
        constexpr int D = J / K;
        const u32 sel = lc.k[ilog2c(D)] ^ dirm; // ~0: this lane keeps the larger key
#pragma unroll
        for (int r = 0; r < K; r++) {
            const u32 t = lane_xor<D, XM>(x[r], lc.addr32);
            x[r] = umed3(x[r], t, sel);
        }
    } else { // partner in another register of the same lane
#pragma unroll
        for (int r = 0; r < K; r++) {
            if ((r & J) == 0) {
                const u32 a = x[r], b = x[r | J];
                if constexpr (M < K) {
                    const bool desc = (r & M) != 0;
                    x[r] = desc ? max(a, b) : min(a, b);
                    x[r | J] = desc ? min(a, b) : max(a, b);
                } else if constexpr (lane_dir) {
                    x[r] = umed3(a, b, dirm);
                    x[r | J] = umed3(a, b, ~dirm);
                } else {
                    x[r] = min(a, b);
                    x[r | J] = max(a, b);
                }
            }
        }
    }
}
template <int LANES, int K, int M, int J, int XM>
__device__ __forceinline__ void bitonic_level(u32 (&x)[K], const LaneConst &lc)
{
    bitonic_step<LANES, K, M, J, XM>(x, lc);
    if constexpr (J > 1) bitonic_level<LANES, K, M, J / 2, XM>(x, lc);
}
template <int LANES, int K, int M, int XM>
__device__ __forceinline__ void bitonic_sort(u32 (&x)[K], const LaneConst &lc)
{
    bitonic_level<LANES, K, M, M / 2, XM>(x, lc);
    if constexpr (M < LANES * K) bitonic_sort<LANES, K, M * 2, XM>(x, lc);
}

// ---- one read per group of LANES lanes, K keys per lane ------------------------------------
// LANES = 64: one read per wavefront.  LANES = 16: four reads per wavefront, one per DPP row — every
// cross-lane step then stays inside a row (10 cross-lane sort stages instead of 21, 4-step scans
// instead of 6) and is shared by four reads.  Arguments are per lane but uniform inside a group.
// The last lane of the group owns the inclusive scan totals and finishes the read.
// 32-lane groups (two reads per wavefront) are row scans plus the row_bcast:15 step.
template <int LANES>
__device__ __forceinline__ u32 gscan_add(u32 v)
{
    if (LANES == 64) return wscan_add(v);
    v = rscan_add(v);
    if (LANES == 32) v += YK_DPP0(v, DPP_BCAST15, 0xA);
    return v;
}
template <int LANES>
__device__ __forceinline__ u32 gscan_max(u32 v)
{
    if (LANES == 64) return wscan_max(v);
    v = rscan_max(v);
    if (LANES == 32) v = max(v, YK_DPP0(v, DPP_BCAST15, 0xA));
    return v;
}
template <int LANES>
__device__ __forceinline__ u32 gshift_up1(u32 v)
{
    if (LANES == 16) return rshift_up1(v);
    const u32 t = wshift_up1(v);
    if (LANES == 32) return (lane_id() == 32u) ? 0u : t; // lane 32 opens the second group
    return t;
}
template <int LANES>
__device__ __forceinline__ u32 gscan_min(u32 v)
{
    v = rscan_min(v);
    if (LANES >= 32)
        v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_BCAST15, 0xA, 0xF, false));
    if (LANES == 64)
        v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, DPP_BCAST31, 0xC, 0xF, false));
    return v;
}

// ---- everything after the event keys are in registers: sort, sweep, regions out ---------------
// m = number of real keys of the group (the rest are pads); zl_check = the wavefront holds >= 2
// zero-length intervals (duplicates must be looked for after the sort).
template <int LANES, int K, int XM>
__device__ __forceinline__ void sweep_group_keys(u32 (&x)[K], u32 m, u32 len, i32 c,
                                                 bool active, u32 r, u64 badmask, u64 zmask,
                                                 bool zl_check, const SweepArgs &a,
                                                 const LaneConst &lc)
{
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1);
    // the read's region slot, looked up again where it is needed (rarely, and by few lanes):
    // kept as a pointer it costs two registers from the loads to the last line
    auto slot_of = [&]() {
        u32 rr = r;
        asm volatile("" : "+v"(rr)); // keeps the address arithmetic here instead of hoisted and spilled
        return a.stage + (a.off[rr] + 2 * (u64)rr);
    };

    bitonic_sort<LANES, K, 2, XM>(x, lc);

    // two zero-length intervals at one position cannot be expressed by the keys: after the sort
    // they are adjacent equal class-1 keys.  Only looked for when the wavefront saw >= 2 of them.
    if (zl_check) {
        bool dup = false;
#pragma unroll
        for (int q = 0; q + 1 < K; q++) dup |= x[q] == x[q + 1] && (x[q] & 3u) == 1u && x[q] != 1u;
        const u32 prev = gshift_up1<LANES>(x[K - 1]);
        dup |= lig != 0 && prev == x[0] && (prev & 3u) == 1u && prev != 1u;
        badmask |= __builtin_amdgcn_ballot_w64(dup);
    }
    const bool group_bad = LANES == 64   ? badmask != 0
                           : LANES == 32 ? (u32)(badmask >> (lane & 32u)) != 0
                                         : ((u32)(badmask >> (lane & 48u)) & 0xFFFFu) != 0;

    // ---- pass 1: depth carried into each lane
    u32 n_starts = 0; // net depth change of the lane = starts - ends = 2 * starts - K
#pragma unroll
    for (int q = 0; q < K; q++) n_starts += x[q] & 1u;
    const u32 delta = 2u * n_starts - (u32)K;
    const u32 dincl = gscan_add<LANES>(delta);
    const i32 depth_in = (i32)(dincl - delta);

    // ---- pass 2: last flagged end / last low start of the lane (keys ascend, so last = max)
    // Passes 2 and 3 (straight-line per variant so the per-key depth / flag values are shared):
    //   pass 2  last flagged end / last low start of the lane (keys ascend, so last = max)
    //   pass 3  regions closed in this lane; tail rule candidates (stack.rs:93-105)
    // Flagged ends are carried between lanes in the flipped domain tk = key ^ 2
    // (device_common.h).  A wavefront without zero-length intervals (all but ~0.1 % of them) only
    // holds classes 0 and 3, where "effective" is plain "flagged" and the loops stay in the true
    // key domain (ZL = false).
    // an end is in the tail when every start precedes it: starts before = (index + depth) / 2
    const u32 tail_base = m - lig * (u32)K;
    const u32 len_key = len > kMaxKeyPos ? 0xFFFFFFFFu : (len << kKeyShift);
    u32 mf_incl, ml_incl, mf_in, ml_in;
    u32 cnt = 0, fb = 0, fe = 0, cand = kNoKey;
    // The loops carry dd = depth + (index inside the lane): one instruction per key (dd += 2 *
    // start bit) instead of a select and an add; "depth > c" becomes dd > c + q with c + q a
    // scalar, and the tail test (starts before == all starts) dd == tail_base.
    auto passes = [&](auto zl_tag) {
        constexpr bool ZL = decltype(zl_tag)::value;
        u32 mf = 0, ml = 0;
        i32 dd = depth_in;
#pragma unroll
        for (int q = 0; q < K; q++) {
            const u32 key = x[q], bit = key & 1u;
            const bool is_s = bit != 0, gt = dd > c + q;
            ml = (is_s && !gt) ? key : ml;
            if (ZL) mf = (!is_s && gt) ? max(mf, key ^ 2u) : mf;
            else mf = (!is_s && gt) ? key : mf;
            dd += (i32)(bit << 1);
        }
        if (!ZL) mf = mf ? (mf ^ 2u) : 0u;
        mf_incl = gscan_max<LANES>(mf);
        ml_incl = gscan_max<LANES>(ml);
        mf_in = max(gshift_up1<LANES>(mf_incl), kNoFlag);
        ml_in = gshift_up1<LANES>(ml_incl);

        u32 tc = ZL ? mf_in : (mf_in ^ 2u); // ZL: flipped domain; else true key, "none" = 3
        u32 cml = ml_in;
        dd = depth_in;
#pragma unroll
        for (int q = 0; q < K; q++) {
            const u32 key = x[q], bit = key & 1u;
            const bool is_s = bit != 0, gt = dd > c + q;
            const bool fl = !is_s && gt, low = is_s && !gt;
            const bool eff = ZL ? (fl && (key ^ 2u) > tc) : fl;
            const u32 begin = ZL ? (tc ^ 2u) : tc;
            const bool close = eff && cml > begin;
            cnt += close ? 1u : 0u;
            fb = close ? begin : fb;
            fe = close ? cml : fe;
            const bool tail = fl && ((u32)dd == tail_base) && key >= len_key;
            cand = min(cand, tail ? (key >> kKeyShift) : kNoKey);
            tc = eff ? (ZL ? (key ^ 2u) : key) : tc;
            cml = low ? key : cml;
            dd += (i32)(bit << 1);
        }
    };
    if (zmask == 0) passes(std::false_type{}); // wave-uniform
    else passes(std::true_type{});
    i32 d;

    const bool live = active && !group_bad;
    u32 g_closed = 0;
    if (__builtin_amdgcn_ballot_w64(cnt != 0) != 0) {
        const u32 cincl = gscan_add<LANES>(cnt);
        g_closed = cincl; // meaningful on the group's last lane
        u32 pos = cincl - cnt;
        if (live && cnt == 1) {
            slot_of()[pos] = make_uint2(fb >> kKeyShift, fe >> kKeyShift);
        } else if (live && cnt > 1) { // several regions close inside one lane: replay it
            uint2 *slot = slot_of();
            u32 tc = mf_in, cml = ml_in;
            d = depth_in;
#pragma unroll
            for (int q = 0; q < K; q++) {
                const u32 key = x[q];
                const bool is_s = (key & 1u) != 0, gt = d > c;
                const bool fl = !is_s && gt, low = is_s && !gt;
                const bool eff = fl && (key ^ 2u) > tc;
                if (eff && cml > (tc ^ 2u))
                    slot[pos++] = make_uint2((tc ^ 2u) >> kKeyShift, cml >> kKeyShift);
                tc = eff ? (key ^ 2u) : tc;
                cml = low ? key : cml;
                d += is_s ? 1 : -1;
            }
        }
    }
    u32 min_ge = kNoKey;
    if (__builtin_amdgcn_ballot_w64(cand != kNoKey) != 0) min_ge = gscan_min<LANES>(cand);
    if (lig == LANES - 1 && active) { // the group's last lane holds every inclusive total
        if (group_bad) {
            a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
            a.counts[r] = 0;
        } else {
            a.counts[r] = finish_read(slot_of(), g_closed, mf_incl ? (mf_incl ^ 2u) : 0u, ml_incl, min_ge, len);
        }
    }
}

// ---- coverage pre-filter (emulated and fuzzed in tests/formulation.py) -----------
// Most of a well-covered read lies deeper than `c`: nothing there can open, close or bound a bad
// region.  The read is cut into NB = LANES bins of 2^sh positions (one bin per lane).  A bin is
// *safe* when more than c intervals span it completely (start in an earlier bin, end in a later
// one): every event inside then has depth_before > c, so its starts are never low and its flagged
// ends are always superseded by a later flagged end outside (the spanning intervals still have
// to end before the depth can reach c).  Every event in a safe bin is dropped; each maximal run
// of safe bins is stood in for by |net| start (net > 0) or end (net < 0) keys at the run's first
// position, net = depth after the run - depth before it, so the depth of every surviving event is
// unchanged.  The per-bin counts come from an LDS histogram (starts | ends << 16, LDS atomics:
// the LDS pipe is otherwise idle here) and one packed row scan.  Survivors are compacted through
// LDS; if every group of the wavefront keeps <= LANES*K/2 keys the caller sorts K/2 keys per lane.
// Exactness does not depend on the bins (any under-estimate of "safe" is fine); reference
// semantics: src/stack.rs:61-139 via the event formulation above.  Only called for wavefronts
// whose intervals all end at or before their read's length (bin index < NB without a clip).
__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wavefront execute in order; this only stops the compiler from moving
    // LDS accesses across the point where lanes exchange data through LDS.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS scratch of one wavefront for the filter: per group LANES coarse bins + the pads' bin + two
// one-position bins, four counters each (one per lane & 3: the reads' hot bins would otherwise
// serialise the LDS atomics of a row), then the compacted keys of every group.
constexpr int kFilterTabWords = 304, kFilterKeyWords = 512;
template <int WPB> // wavefronts per workgroup
__device__ __forceinline__ u32 *wave_filter_scratch()
{
    __shared__ __attribute__((aligned(16))) u32 s_scratch[WPB][kFilterTabWords + kFilterKeyWords];
    return s_scratch[threadIdx.x >> 6];
}

// ---- coverage pre-filter with pile trimming at the two ends of the read (DESIGN.md §3.4 / §3.5;
// tests/formulation.py::trim_keys_minmax is the emulation) ------------------------------------------
// Bins: NB = LANES coarse bins of 2^sh positions, one per lane, plus one bin for the read's SMALLEST
// START position and one for its LARGEST END position — where dovetail overlaps clamp: 0 and `len`
// for a healthy read (on configs[1] 15 % of a read's starts sit at exactly 0 and 15 % of its ends at
// exactly len), the edges of the covered window for a read that is only covered in part.  In key
// order: [starts at pmin][coarse 0 .. NB-1][ends at pmax] (nothing else can lie at those two
// positions: an end is above its own start, a start below its own end).
//   * coarse bin spanned by more than c intervals (depth at its head - its ends > c): every event in
//     it is deep (depth above c on both sides) and dropped; otherwise the bin is kept whole;
//   * starts at pmin: the j-th one has depth j, so only the first c + 1 are kept; ends at pmax: the
//     depth before the j-th of E is E - j, so only the last c + 1 are kept.  Equal keys: which of
//     them does not matter.
// Each maximal run of dropped events is stood in for by |net| keys of one type in front of the next
// bin that keeps something (net = depth after the run - depth before it; 0 for a healthy read), so
// the depth of every kept event is unchanged.  Pass 1 counts (one LDS atomic per key, starts in
// the low half of a counter, ends in the high half), the lanes turn the counters into packed cursors
// quota << 16 | next slot, pass 2 asks them (one LDS atomic per key: kept or not, and where).
// A group that keeps more than LANES * K / 2 keys (a read with low coverage throughout: nothing
// can be dropped) is HEAVY: with DEFER it is reported through `heavy` and goes on empty — the
// caller appends the read to the overflow list, sweep_deferred_kernel sorts it whole — so that one
// such read does not drag the other reads of its wavefront into the full sort (and the full sort
// is not even part of that code path: fewer registers, one more wavefront per SIMD).
// Returns the tier: 0 = sort everything as before, 1 = every group kept <= LANES keys (y1: one
// key per lane), 2 = y: K / 2 keys per lane, 3 = nothing to sort: every group is a healthy read (hr).  Only for wavefronts whose intervals are all plain
// (start < end <= len).
// What trimfilter hands back for a wavefront of healthy reads (tier 3): per group, the number of
// starts kept at the smallest start position and the two positions.
struct HealthyRead {
    u32 kept_starts, pmin, pmax;
};

template <int LANES, int K, bool DEFER, int WPB>
__device__ __forceinline__ int trimfilter(const u32 (&x)[K], u32 n, u32 len, i32 c,
                                          u32 (&y1)[1], u32 (&y)[K / 2], u32 &m_out, bool &heavy,
                                          HealthyRead &hr)
{
    static_assert(K == 16, "a lane reads its K/2 = 8 compacted keys as two 16-byte vectors");
    constexpr int NB = LANES, NBIN = LANES + 3, CAP = LANES * K / 2, GROUPS = 64 / LANES;
    constexpr u32 kTakeOne = 0xFFFF0001u; // quota - 1, slot + 1
    constexpr u32 kPadBin = NB, kHeadBin = NB + 1, kTailBin = NB + 2;
    static_assert(GROUPS * NBIN * 4 <= kFilterTabWords && GROUPS * CAP <= kFilterKeyWords, "scratch");
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1), grp = lane / (u32)LANES;
    u32 *scratch = wave_filter_scratch<WPB>();
    u32 *tab = scratch + grp * (u32)(NBIN * 4);
    u32 *keys = scratch + kFilterTabWords + grp * (u32)CAP;
    uint4 *bins = reinterpret_cast<uint4 *>(tab);
    uint4 *my_keys = reinterpret_cast<uint4 *>(keys) + lig * 2u;
    char *tb = reinterpret_cast<char *>(tab);
    heavy = false;

    // smallest shift with (len >> sh) < NB: the bin holding `len` exists inside the table
    const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(NB) + (len != 0 ? 0 : -1);
    const u32 sh = (u32)max(bits, 0), ksh = sh + kKeyShift;

    bins[lig] = make_uint4(0u, 0u, 0u, 0u);
    if (lig < 3u) bins[NB + lig] = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_sync();

    // ---- pass 1: count.  Byte offset of a key's counter = bin * 16 + (lane & 3) * 4; the pads
    // (0xFFFFFFFE, in the slots of intervals the read does not have) clamp into the pads' bin
    const u32 cp = (lig & 3u) * 4u, head_off = kHeadBin * 16u + cp, tail_off = kTailBin * 16u + cp;
    // the group's smallest start key and largest end key (the pads are end-like and huge: they never
    // win the min, and are masked out of the max); a group without intervals matches nothing
    const int last_addr = (int)((lane | (u32)(LANES - 1)) << 2);
    auto row_total = [&](u32 v) { return (u32)__builtin_amdgcn_ds_bpermute(last_addr, (int)v); };
    u32 smin = x[0], emax = 0;
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
        smin = min(smin, x[2 * j]);
        emax = max(emax, x[2 * j + 1] == kPadKey ? 0u : x[2 * j + 1]);
    }
    const u32 kmin = n ? row_total(gscan_min<LANES>(smin)) : 1u, kmax = row_total(gscan_max<LANES>(emax));
    auto off_start = [&](u32 ks) { return ks == kmin ? head_off : (min(ks >> ksh, (u32)NB) << 4) + cp; }; // the smallest start
    auto off_end = [&](u32 ke) { return ke == kmax ? tail_off : (min(ke >> ksh, (u32)NB) << 4) + cp; };   // the largest end
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
        atomicAdd(reinterpret_cast<u32 *>(tb + off_start(x[2 * j])), 1u);
        atomicAdd(reinterpret_cast<u32 *>(tb + off_end(x[2 * j + 1])), 0x10000u);
    }
    wave_lds_sync();

    // ---- what the bins keep
    // (the two one-position bins are read again where lanes 0 and 1 deal out their quota: holding
    // them until then costs eight registers at the kernel's high-water mark)
    u32 w, n0, n01, n012; // this lane's bin: its total, and the running sizes of its first three copies
    i32 S0, E1;
    {
        const uint4 c4 = bins[lig], h4 = bins[kHeadBin], t4 = bins[kTailBin];
        n0 = (c4.x & 0xFFFFu) + (c4.x >> 16);
        n01 = n0 + (c4.y & 0xFFFFu) + (c4.y >> 16);
        n012 = n01 + (c4.z & 0xFFFFu) + (c4.z >> 16);
        w = c4.x + c4.y + c4.z + c4.w;
        S0 = (i32)((h4.x + h4.y + h4.z + h4.w) & 0xFFFFu);  // starts at the smallest start position
        E1 = (i32)((t4.x + t4.y + t4.z + t4.w) >> 16);      // ends at the largest end position
    }
    const i32 S = (i32)(w & 0xFFFFu), E = (i32)(w >> 16);
    const i32 ks0 = min(S0, c + 1), ke1 = min(E1, c + 1);
    const u32 incl = gscan_add<LANES>(w); // packed: both halves scanned at once
    const u32 ex = incl - w;
    const i32 D = S0 + (i32)(ex & 0xFFFFu) - (i32)(ex >> 16);     // depth at the head of this lane's bin
    const bool deep = D - E > c;                                   // spanned by more than c intervals
    const u32 keep = deep ? 0u : (u32)(S + E);
    // The healthy read: every coarse bin that holds anything is deep and the two piles balance.  What
    // is kept is then min(S0, c + 1) copies of the smallest start key and as many of the largest end
    // key, and what the sweep makes of those is known in closed form (sweep_group_read; DESIGN.md
    // §3.5; tests/formulation.py::healthy_read_regions): when every group of the wavefront is like
    // that — nine wavefronts in ten on configs[1] and [2] — the rest of the plan, pass 2, the sort and
    // the sweep are skipped.
    if (__builtin_amdgcn_ballot_w64(keep != 0u || ks0 != ke1) == 0) {
        hr.kept_starts = (u32)ks0;
        hr.pmin = kmin >> kKeyShift;
        hr.pmax = kmax >> kKeyShift;
        return 3;
    }
    // (sequence index + 1) << 16 | depth after the kept block, for bins that keep something (an empty
    // bin keeps nothing: the bins in front of the smallest start, where D is not the depth, are empty)
    const u32 tag = (deep || w == 0u) ? 0u : (((lig + 2u) << 16) | (u32)(D - E + S));
    const u32 tag0 = S0 > 0 ? ((1u << 16) | (u32)ks0) : 0u;
    const u32 mi = gscan_max<LANES>(tag);
    u32 exm = gshift_up1<LANES>(mi);
    if (LANES == 16 && lig == 0) exm = 0; // (row_shr pulls nothing in, bound_ctrl zero: explicit for clarity)
    const u32 prev = max(exm, tag0);
    const i32 net = tag ? D - (i32)(prev & 0xFFFFu) : 0;
    const u32 nsyn = (u32)(net < 0 ? -net : net);
    const i32 a_last = (i32)(max(row_total(mi), tag0) & 0xFFFFu);
    const i32 net1 = E1 > 0 ? ke1 - a_last : 0;
    const u32 nsyn1 = (u32)(net1 < 0 ? -net1 : net1);
    const u32 mine = keep + nsyn;
    const u32 ri = gscan_add<LANES>(mine);
    const u32 coarse_total = row_total(ri);
    const u32 base = (u32)ks0 + ri - mine;
    const u32 tail_base = (u32)ks0 + coarse_total;
    u32 m = tail_base + (u32)ke1 + nsyn1;
    heavy = m > (u32)CAP;
    if constexpr (DEFER) m = heavy ? 0u : m; // its read goes to the overflow list; the group goes on empty
    else if (__builtin_amdgcn_ballot_w64(heavy) != 0) return 0; // the wavefront sorts everything
    my_keys[0] = make_uint4(kPadKey, kPadKey, kPadKey, kPadKey);
    my_keys[1] = make_uint4(kPadKey, kPadKey, kPadKey, kPadKey);
    {
        // cursors: a coarse bin's copies keep everything (quota 0x7FFF) or nothing
        const u32 q = (deep || heavy) ? 0u : 0x7FFF0000u;
        const uint4 s4 = bins[kHeadBin + min(lig, 1u)];
        bins[lig] = make_uint4(q | base, q | (base + n0), q | (base + n01), q | (base + n012));
        // the two one-position bins hold one type each: their quota is dealt out to the four copies
        if (lig < 2u) {
            const u32 sh2 = lig ? 16u : 0u;
            const u32 keep2 = heavy ? 0u : (u32)(lig ? ke1 : ks0), b2 = lig ? tail_base : 0u;
            const u32 k0 = (s4.x >> sh2) & 0xFFFFu, k1 = (s4.y >> sh2) & 0xFFFFu, k2 = (s4.z >> sh2) & 0xFFFFu;
            const u32 q0 = min(k0, keep2), q1 = min(k1, keep2 - q0), q2 = min(k2, keep2 - q0 - q1),
                      q3 = keep2 - q0 - q1 - q2;
            bins[kHeadBin + lig] = make_uint4((q0 << 16) | b2, (q1 << 16) | (b2 + q0), (q2 << 16) | (b2 + q0 + q1),
                                              (q3 << 16) | (b2 + q0 + q1 + q2));
        }
        if (lig == 2u) bins[kPadBin] = make_uint4(0u, 0u, 0u, 0u); // the pads: quota 0
    }
    if (__builtin_amdgcn_ballot_w64(((nsyn | nsyn1) != 0) && !heavy) != 0) { // rare: net != 0 somewhere
        if (!heavy) {
            // starts go right in front of the bin (position - 1, start class; never in front of the
            // kept starts at the smallest start position), ends to its head
            const u32 pk = (lig << sh) << kKeyShift;
            const u32 synkey = net > 0 ? max(pk, kmin + 1u) - 1u : pk;
#pragma unroll 1
            for (u32 t = 0; t < nsyn; t++) keys[base + keep + t] = synkey;
            if (lig == 0) {
                const u32 synkey1 = net1 > 0 ? kmax - 1u : kmax;
#pragma unroll 1
                for (u32 t = 0; t < nsyn1; t++) keys[tail_base + (u32)ke1 + t] = synkey1;
            }
        }
    }
    wave_lds_sync();

    // ---- pass 2: every key asks the cursor it counted on
#pragma unroll
    for (int q0 = 0; q0 < K; q0 += 8) {
        u32 got[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            got[q] = atomicAdd(reinterpret_cast<u32 *>(tb + ((q & 1) ? off_end(x[q0 + q]) : off_start(x[q0 + q]))), kTakeOne);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if ((i32)got[q] >= 0x10000) keys[got[q] & 0xFFFFu] = x[q0 + q];
        }
    }
    wave_lds_sync();
    m_out = m;
    if (__builtin_amdgcn_ballot_w64(m > (u32)LANES) == 0) { // every group fits one key per lane
        y1[0] = keys[lig];
        return 1;
    }
    const uint4 lo = my_keys[0], hi = my_keys[1];
    y[0] = lo.x, y[1] = lo.y, y[2] = lo.z, y[3] = lo.w;
    y[4] = hi.x, y[5] = hi.y, y[6] = hi.z, y[7] = hi.w;
    return 2;
}

// ---- the bin filter without trimming (round 1): used by the builds that do not
// defer (sweep_small_fused_kernel for short launches, the one-read-per-wavefront class), where the
// 16-keys-per-lane fallback is part of the code path and this leaner filter fits 96 registers ------
template <int LANES, int K, int WPB>
__device__ __forceinline__ bool prefilter(const u32 (&x)[K], u32 n, u32 len, i32 c, u32 (&y)[K / 2],
                                          u32 &m_out)
{
    static_assert(K == 16, "a lane reads its K/2 = 8 compacted keys as two 16-byte vectors");
    constexpr int NB = LANES, CAP = LANES * K / 2, GROUPS = 64 / LANES;
    static_assert(GROUPS * (NB + 1) * 4 <= kFilterTabWords && GROUPS * CAP <= kFilterKeyWords, "scratch");
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1), grp = lane / (u32)LANES;
    u32 *scratch = wave_filter_scratch<WPB>();
    u32 *tab = scratch + grp * (u32)((NB + 1) * 4);        // (NB bins + one for the pads) x 4 copies
    u32 *keys = scratch + kFilterTabWords + grp * (u32)CAP;
    uint4 *my_bin = reinterpret_cast<uint4 *>(tab) + lig;
    uint4 *my_keys = reinterpret_cast<uint4 *>(keys) + lig * 2u;

    // smallest shift with (len >> sh) < NB: the bin holding `len` and every later one stay unsafe
    const i32 bits = 32 - (i32)__builtin_clz(len | 1u) - ilog2c(NB) + (len != 0 ? 0 : -1);
    const u32 sh = (u32)max(bits, 0), ksh = sh + kKeyShift;

    // ---- histogram: starts in the low half of a counter, ends in the high half (LDS atomics).
    // The compacted-key area starts out as pads.
    *my_bin = make_uint4(0u, 0u, 0u, 0u);
    my_keys[0] = make_uint4(kPadKey, kPadKey, kPadKey, kPadKey);
    my_keys[1] = make_uint4(kPadKey, kPadKey, kPadKey, kPadKey);
    wave_lds_sync();
    u32 *cell0 = tab + (lig & 3u);        // this lane's copy of bin 0
    u32 *pad_cell = cell0 + NB * 4;
    u32 *cell[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
        const bool real = lig + (u32)LANES * (q / 2) < n;
        u32 *p = cell0 + (x[q] >> ksh) * 4u; // positions <= len: the bin is inside the table
        cell[q] = real ? p : pad_cell;
        atomicAdd(cell[q], (q & 1) ? 0x10000u : 1u);
    }
    wave_lds_sync();
    const uint4 w4 = *my_bin;
    const u32 w = w4.x + w4.y + w4.z + w4.w;
    const u32 incl = gscan_add<LANES>(w); // packed: both halves scanned at once
    const i32 S = (i32)(w & 0xFFFFu), E = (i32)(w >> 16);
    const i32 cs = (i32)(incl & 0xFFFFu), ce = (i32)(incl >> 16);
    const i32 depth_after = cs - ce, depth_at = depth_after - (S - E);
    const bool safe = (cs - S) - ce > c && lig < (len >> sh);
    const u64 sball = __builtin_amdgcn_ballot_w64(safe);
    if (sball == 0) return false; // wave-uniform: nothing to drop
    // the group's safe bits, bit b = bin b (zeros beyond the group)
    const u64 smask = LANES == 64 ? sball
                                  : (sball >> (lane & (u32)(64 - LANES))) & ((1ull << (LANES & 63)) - 1ull);
    const bool head = safe && (((smask << 1) >> lig) & 1ull) == 0;
    const bool tail = safe && (((smask >> 1) >> lig) & 1ull) == 0;
    const u32 hv = gscan_max<LANES>(head ? (((lig + 1u) << 16) | (u32)depth_at) : 0u);
    const i32 net = tail ? depth_after - (i32)(hv & 0xFFFFu) : 0;
    const u32 nsyn = min((u32)(net < 0 ? -net : net), (u32)CAP + 1u);
    const u32 synkey = (((hv >> 16) - 1u) << ksh) | (net > 0 ? 3u : 0u);

    // ---- slots: a bin's survivors go to [base, base + S + E), a run's stand-ins to the tail
    // lane's range; counters of safe bins (and of the pads) start at CAP = "nowhere"
    const u32 mine = safe ? nsyn : (u32)(S + E);
    const u32 rincl = gscan_add<LANES>(mine);
    const u32 m = (u32)__builtin_amdgcn_ds_bpermute((int)((lane | (u32)(LANES - 1)) << 2), (int)rincl);
    if (__builtin_amdgcn_ballot_w64(m > (u32)CAP) != 0) return false; // some group keeps too much
    const u32 base = rincl - mine;
    {
        uint4 b4;
        b4.x = base;
        b4.y = b4.x + (w4.x & 0xFFFFu) + (w4.x >> 16);
        b4.z = b4.y + (w4.y & 0xFFFFu) + (w4.y >> 16);
        b4.w = b4.z + (w4.z & 0xFFFFu) + (w4.z >> 16);
        *my_bin = safe ? make_uint4(CAP, CAP, CAP, CAP) : b4;
        if (lig < 4u) tab[NB * 4 + lig] = (u32)CAP;
    }
#pragma unroll 1
    for (u32 t = 0; t < nsyn; t++) keys[base + t] = synkey;
    wave_lds_sync();
    // slot requests go out in batches of 8, all in flight before the first store needs its answer
    // (16 at once cost eight more registers than the rest of the kernel needs)
#pragma unroll
    for (int q0 = 0; q0 < K; q0 += 8) {
        u32 pos[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pos[q] = atomicAdd(cell[q0 + q], 1u);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (pos[q] < (u32)CAP) keys[pos[q]] = x[q0 + q];
        }
    }
    wave_lds_sync();
    const uint4 lo = my_keys[0], hi = my_keys[1];
    y[0] = lo.x, y[1] = lo.y, y[2] = lo.z, y[3] = lo.w;
    y[4] = hi.x, y[5] = hi.y, y[6] = hi.z, y[7] = hi.w;
    m_out = m;
    return true;
}

// ---- one read per group of LANES lanes: loads, keys, (pre-filter,) sweep ------------------------
template <int LANES, int K, int XM, bool DEFER = false, int WPB = 4>
__device__ __forceinline__ void sweep_group_read(const uint2 *__restrict__ iv, u32 n, u32 len,
                                                 u32 cov, bool active, u32 r,
                                                 const SweepArgs &a, const LaneConst &lc)
{
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1);
    const i32 c = (i32)min(cov, 0x3FFFFFFFu); // depths are <= 1024: every larger c behaves the same, and c + q cannot overflow

    // ---- coalesced interval loads (8 B/lane), keys straight into registers
    u32 x[K];
    u32 bad = 0, nz = 0;
    bool plain;
    {
        // Every load is issued before the first use (one memory latency per read, not K/2), from
        // one base pointer with the index clamped to the read's last interval: no per-load
        // branches or address arithmetic; the duplicates are turned into pads below.  A group
        // without intervals reads the first offsets instead (always mapped).
        const uint2 *src = n ? iv : reinterpret_cast<const uint2 *>(a.off);
        const u32 last = n ? n - 1u : 0u;
        uint2 v[K / 2];
#pragma unroll
        for (int j = 0; j < K / 2; j++) v[j] = src[min(lig + (u32)LANES * j, last)];
        // Plain intervals (start < end <= min(len, kMaxKeyPos)) need two instructions per key;
        // a wavefront that holds anything else (zero-length, start > end, an end beyond the read,
        // huge positions: ~0.1 % of them) re-derives its keys with the class and rejection logic
        // of make_event_keys and sorts everything (the pre-filter relies on positions <= len).
        const u32 len_c = min(len, kMaxKeyPos);
        u32 irregular = 0;
#pragma unroll
        for (int j = 0; j < K / 2; j++) {
            const bool real = lig + (u32)LANES * j < n;
            irregular |= (real && (v[j].x >= v[j].y || v[j].y > len_c)) ? 1u : 0u;
            x[2 * j] = real ? ((v[j].x << kKeyShift) | 3u) : kPadKey;
            x[2 * j + 1] = real ? (v[j].y << kKeyShift) : kPadKey;
        }
        plain = __builtin_amdgcn_ballot_w64(irregular != 0) == 0; // wave-uniform
        if constexpr (K == 16 && DEFER) {
            // the deferring build holds neither the class / rejection logic nor the 16-keys-per-lane
            // sort: the reads of such a wavefront are finished by sweep_deferred_kernel
            if (!plain) {
                if (lig == LANES - 1 && active) a.over_list[atomicAdd(a.over_count, 1u)] = r;
                return;
            }
        } else if (!plain) {
#pragma unroll
            for (int j = 0; j < K / 2; j++) {
                u32 ks, ke, b = 0, z = 0;
                make_event_keys(v[j], ks, ke, b, z);
                const bool real = lig + (u32)LANES * j < n;
                x[2 * j] = real ? ks : kPadKey;
                x[2 * j + 1] = real ? ke : kPadKey;
                bad |= real ? b : 0u;
                nz += real ? z : 0u;
            }
        }
    }
    const u64 badmask = __builtin_amdgcn_ballot_w64(bad != 0);
    const u64 zmask = __builtin_amdgcn_ballot_w64(nz != 0);
    if (LANES == 64 && badmask != 0) { // wave-uniform: skip the work, queue for the exact path
        if (lig == LANES - 1 && active) {
            a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
            a.counts[r] = 0;
        }
        return;
    }
    // two zero-length intervals at one position: only looked for when the wavefront saw >= 2
    const bool zl_check = (zmask & (zmask - 1)) != 0 || __builtin_amdgcn_ballot_w64(nz > 2) != 0;

    if constexpr (K == 16 && DEFER) {
        { // (the engine only launches this build with the filter on; every wavefront here is plain)
            u32 y1[1], y[K / 2], mf;
            bool heavy;
            HealthyRead hr;
            const int tier = trimfilter<LANES, K, true, WPB>(x, n, len, c, y1, y, mf, heavy, hr);
            if (a.prefilter == 2 && lig == 0 && active && !heavy) atomicAdd(&a.ctr->prefiltered, 1u);
            if (tier == 3) {
                // k starts at pmin then k ends at pmax, k = min(S0, c + 1): the depth passes c only
                // when k = c + 1, and then exactly between the two positions — the read is bad in
                // front of pmin and behind pmax (the sweep's first closed region and finish_read's
                // last one); with k <= c (or no interval at all) nothing ever exceeds c and the whole
                // read is one bad region (finish_read's "mf_t == 0" branch).  len >= 1 whenever n >= 1.
                if (lig == 0 && active) {
                    uint2 *slot = a.stage + (a.off[r] + 2 * (u64)r);
                    u32 g = 0;
                    if ((i32)hr.kept_starts <= c) {
                        if (len != 0) slot[g++] = make_uint2(0u, len);
                    } else {
                        if (hr.pmin != 0) slot[g++] = make_uint2(0u, hr.pmin);
                        if (hr.pmax != len) slot[g++] = make_uint2(hr.pmax, len);
                    }
                    a.counts[r] = g;
                }
                return;
            }
            const bool act = active && !heavy; // a heavy read is finished by sweep_deferred_kernel
            // the sort's per-lane constants are derived here, not carried through the filter (8 registers)
            u32 lane2 = lane;
            asm volatile("" : "+v"(lane2));
            const LaneConst lc2 = make_lane_const(lane2);
            if (tier == 1) sweep_group_keys<LANES, 1, XM>(y1, mf, len, c, act, r, badmask, zmask, zl_check, a, lc2);
            else sweep_group_keys<LANES, K / 2, XM>(y, mf, len, c, act, r, badmask, zmask, zl_check, a, lc2);
            // a read the filter could not thin goes to the overflow list: sweep_deferred_kernel sorts it
            // whole, one read per wavefront
            if (heavy && lig == LANES - 1 && active) a.over_list[atomicAdd(a.over_count, 1u)] = r;
        }
        return;
    } else if constexpr (K == 16) {
        if (a.prefilter && plain) { // uniform
            u32 y[K / 2], mf;
            if (prefilter<LANES, K, WPB>(x, n, len, c, y, mf)) {
                if (a.prefilter == 2 && lig == 0 && active) atomicAdd(&a.ctr->prefiltered, 1u);
                sweep_group_keys<LANES, K / 2, XM>(y, mf, len, c, active, r, badmask, zmask,
                                                   zl_check, a, lc);
                return;
            }
        }
    }
    if constexpr (!(K == 16 && DEFER))
        sweep_group_keys<LANES, K, XM>(x, 2 * n, len, c, active, r, badmask, zmask, zl_check, a, lc);
}

// Body of one workgroup (four wavefronts, 4 * 64/LANES reads) of class (LANES, K).
template <int LANES, int K, int XM, bool DEFER = false, int WPB = 4>
__device__ __forceinline__ void sweep_group_block(const SweepArgs &a, u32 block)
{
    const u32 lane = lane_id();
    const LaneConst lc = make_lane_const(lane);

    constexpr u32 GROUPS = 64 / LANES; // reads per wavefront
    const u32 list_n = *a.list_n;
    const u32 wave = block * (u32)WPB + (threadIdx.x >> 6);
    if (a.first + wave * GROUPS >= list_n) return; // grids may be sized for more reads than the class holds
    const u32 idx = a.first + wave * GROUPS + lane / (u32)LANES;
    const bool active = idx < list_n;
    u32 r = 0, n = 0, len = 0;
    u64 o = 0;
    if (active) {
        r = a.list[idx];
        o = a.off[r];
        n = (u32)(a.off[r + 1] - o);
        len = a.len[r];
    }
    sweep_group_read<LANES, K, XM, DEFER, WPB>(a.iv + o, n, len, a.cov, active, r, a, lc);
}

// One kernel per (LANES, K): small K keep small register footprints.
template <int LANES, int K, int XM>
__global__ __launch_bounds__(256) void sweep_group_kernel(SweepArgs a)
{
    sweep_group_block<LANES, K, XM>(a, blockIdx.x);
}

template <int LANES, int K>
inline void launch_sweep_group(const SweepArgs &sa, u32 n_reads, hipStream_t stream, int xlane_mode)
{
    constexpr u32 per_block = 4u * (64 / LANES); // reads per 256-thread workgroup
    const u32 grid = (n_reads + per_block - 1) / per_block;
    if (xlane_mode == 1)
        hipLaunchKernelGGL((sweep_group_kernel<LANES, K, 1>), dim3(grid ? grid : 1), dim3(256), 0,
                           stream, sa);
    else
        hipLaunchKernelGGL((sweep_group_kernel<LANES, K, 0>), dim3(grid ? grid : 1), dim3(256), 0,
                           stream, sa);
}

// The reads the filter deferred (low coverage throughout: every event kept), sorted whole: one read
// per wavefront on all 64 lanes (4 keys per lane up to 128 intervals, 8 up to 256: a short serial
// chain per wavefront instead of the 16-keys-per-lane sort their class would run).  A persistent
// grid; the count is only known on the device.
__global__ __launch_bounds__(256) void sweep_deferred_kernel(SweepArgs a)
{
    const u32 lane = lane_id();
    const LaneConst lc = make_lane_const(lane);
    const u32 n_list = *a.list_n;
    for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_list; w += gridDim.x * 4u) { // wave-uniform
        const u32 r = a.list[w];
        const u64 o = a.off[r];
        const u32 n = (u32)(a.off[r + 1] - o), len = a.len[r];
        if (n <= 128u) sweep_group_read<64, 4, 0>(a.iv + o, n, len, a.cov, true, r, a, lc);
        else sweep_group_read<64, 8, 0>(a.iv + o, n, len, a.cov, true, r, a, lc);
    }
}

// ---- every register-sort class in ONE launch ------------------------------------------------
// The classes R2..H16 are independent; launched one after the other each pays its own ramp-up
// and drain (~4-7 us for the minor ones on configs[1]).  Here the grid is the concatenation of
// the per-class grids and a workgroup looks up its class (<= 5 uniform compares).
#ifndef YK_DEFER_WAVES
#define YK_DEFER_WAVES 1
#endif
#ifndef YK_DEFER_OCC
#define YK_DEFER_OCC 6
#endif
// wavefronts per workgroup of the fused launch: the deferring build runs one-wavefront workgroups (a
// slot is free again as soon as its wavefront ends, not when the slowest of four does)
constexpr int kFusedWaves = 4, kDeferWaves = YK_DEFER_WAVES, kDeferOcc = YK_DEFER_OCC;
struct FusedArgs {
    SweepArgs base;           // list / list_n filled per class from the table below
    u32 n_entries;
    u32 cls[5];               // CLS_R2 .. CLS_H16
    u32 block_end[5];         // running end of the per-class grids
    u32 first[5];             // SweepArgs.first per class
    const u32 *list[5];
    const u32 *list_n[5];
};

template <bool DEFER, int WPB>
__device__ __forceinline__ void sweep_small_fused_body(const FusedArgs &f)
{
    u32 e = 0, first = 0;
    while (e + 1 < f.n_entries && blockIdx.x >= f.block_end[e]) {
        first = f.block_end[e];
        e++;
    }
    SweepArgs a = f.base;
    a.list = f.list[e];
    a.list_n = f.list_n[e];
    a.first = f.first[e];
    const u32 b = blockIdx.x - first;
    switch (f.cls[e]) { // the one-read-per-wavefront classes stay separate kernels (registers)
    case CLS_R2: sweep_group_block<16, 2, 0, false, WPB>(a, b); break;
    case CLS_R4: sweep_group_block<16, 4, 0, false, WPB>(a, b); break;
    case CLS_R8: sweep_group_block<16, 8, 0, false, WPB>(a, b); break;
    case CLS_R16: sweep_group_block<16, 16, 0, DEFER, WPB>(a, b); break;
    default: sweep_group_block<32, 16, 0, DEFER, WPB>(a, b); break;
    }
}
// Two builds.  DEFER: a read the filter cannot thin goes to f.base.over_list (sweep_deferred_kernel
// finishes it) instead of dragging its wavefront into the 16-keys-per-lane sort; without that sort
// in the filtered path the kernel fits 80 registers, six workgroups per CU, and it is bound by how
// many wavefronts overlap their latencies (configs[2]: 1.90 -> 1.43 ms; configs[1]: 47.9 -> 38.5 +
// 4.4 us).  The extra launch costs ~4 us however little it has to do, so small batches use the
// other build (the engine decides by the classes' interval count).
// (__launch_bounds__' second argument: wavefronts per SIMD)
__global__ __launch_bounds__(64 * kDeferWaves, kDeferOcc) void sweep_small_fused_defer_kernel(FusedArgs f)
{
    sweep_small_fused_body<true, kDeferWaves>(f);
}
__global__ __launch_bounds__(64 * kFusedWaves, 5) void sweep_small_fused_kernel(FusedArgs f)
{
    sweep_small_fused_body<false, kFusedWaves>(f);
}

inline u32 sweep_group_reads_per_block(int cls, int waves = 4)
{
    return ((cls <= CLS_R16) ? 4u : (cls == CLS_H16) ? 2u : 1u) * (u32)waves;
}

} // namespace yk
