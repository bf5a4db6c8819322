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
#ifndef YK_EXP_NO_COUNT
#define YK_EXP_NO_COUNT 0
#endif
#ifndef YK_EXP_NO_FALLBACK
#define YK_EXP_NO_FALLBACK 0
#endif
#ifndef YK_WG_NT
#define YK_WG_NT 0 // (1: the workgroup screen's interval loads non-temporal: A/B, profiles/r06/y_*)
#endif
#ifndef YK_WS_SIZED
#define YK_WS_SIZED 1
#endif
#ifndef YK_EXP_FILT_STAGE
#define YK_EXP_FILT_STAGE 0 // (timing experiments only: wg_filtered_read leaves, "done", behind stage 1 .. 4)
#endif
#ifndef YK_EXP_SKIP_UNDECIDED
#define YK_EXP_SKIP_UNDECIDED 0
#endif
#ifndef YK_WS_SKIP_DEAD
#define YK_WS_SKIP_DEAD 0 // (1: groups of T pairs beyond the read are neither loaded nor counted — measured SLOWER, 0.213 -> 0.248 ms on configs[3], profiles/r06/j_*: the uniform branches keep the eight loads from being issued together)
#endif
#ifndef YK_EXP_NO_FILTERED
#define YK_EXP_NO_FILTERED 0
#endif
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
// what the screen learned about a read it could NOT decide, for wg_filtered_read below (the table stays in LDS)
struct WgVerdict {
    bool plain;   // the table is the read's: every position inside the read and the key range, a span of two windows or more
    bool ends_ok; // F > c, G > c, no end at or before a: (0, a) and (b, len) are the read's first and last regions whatever lies between
    u32 pmin, pmax, sh, ra, rb;
};
// R: intervals per thread and chunk (a multiple of four: pair loads) — the body is straight-line code over R slots, real or not,
// so a read of n <= T * R' < T * R intervals is cheaper through the build for R' (screen_wg_kernel picks one per read).
template <int R = kWsR>
__device__ __forceinline__ bool screen_wg_read(const SweepArgs &a, u32 r, u64 o, u32 n, u32 len, u32 *tab, u32 (*red)[4], u32 *sc, WgVerdict &vd)
{
    static_assert(R % 4 == 0 && R >= 4 && R <= kWsR, "pairs of pairs");
    constexpr int T = kWsT, W = kWsW, NW = T / 64;
    constexpr u32 kEnd = 1u << 16, kField = kEnd - 1u;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    uint4 *bins = reinterpret_cast<uint4 *>(tab);
    char *tb = reinterpret_cast<char *>(tab);
    const uint2 *iv = a.iv + o;
    const u32 chunks = (n + (u32)(T * R) - 1u) / (u32)(T * R);
    bool fallback = n < 2u; // (len is looked at behind the intervals' loads: it may still be on its way)

    // ---- the read's smallest start, largest end, largest start and shortest interval (signed)
#if YK_WS_SKIP_DEAD
    auto live = [&](u32 ch, int j) { return 2u * (ch * (u32)(T * R / 2) + (u32)(j * T)) < n; }; // (uniform: group j of chunk ch holds an interval)
#else
    auto live = [&](u32, int) { return true; };
#endif
    uint4 v[R / 2];
    u32 smin = 0xFFFFFFFFu, emax = 0, smax = 0;
    i32 tmin = 0x7FFFFFFF;
    if (!fallback) {
        for (u32 ch = 0; ch < chunks; ch++) {
            const u32 base = ch * (u32)(T * R / 2) + tid; // (pairs)
            // Round 6: a group of T pairs that lies beyond the read as a whole — uniform: its first interval's index against n —
            // is neither loaded nor looked at (a read of 5 700 intervals fills six of a chunk's eight groups, a read of 600 one:
            // the loads, the minima and the count's sixteen instructions + two LDS atomics per slot were spent on all eight).
#pragma unroll
            for (int j = 0; j < R / 2; j++) { // (slots beyond the read inside a live group: copies of its last two intervals)
                if (live(ch, j)) v[j] = load_pair<(YK_WG_NT != 0)>(iv + min(2u * (base + (u32)(j * T)), n - 2u));
            }
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                if (!live(ch, j)) continue;
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
    fallback = fallback || len > kMaxKeyPos || pmax > len || qmax > kMaxKeyPos || shortest < 0 || pmax - pmin < (u32)(2 * W);

    bool healthy = false;
    u32 ra = 0, rb = 0;
    vd.plain = !fallback, vd.ends_ok = false, vd.pmin = pmin, vd.pmax = pmax, vd.sh = 0, vd.ra = 0, vd.rb = 0;
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
#if !YK_EXP_NO_COUNT // (timing experiments only, tools/build_variant.sh: wrong results)
            if (real && e0 != 0u) {
                atomicAdd(reinterpret_cast<u32 *>(tb + ((is << 4) + cp)), 1u);
                atomicAdd(reinterpret_cast<u32 *>(tb + ((ie << 4) + cp)), kEnd);
            }
#else
            if (real && e0 != 0u && (is ^ ie) == 0xFFFFFFFFu) tb[0] = 1;
#endif
        };
        for (u32 ch = 0; ch < chunks; ch++) {
            const u32 base = ch * (u32)(T * R / 2) + tid;
            if (chunks > 1u) {
#pragma unroll
                for (int j = 0; j < R / 2; j++)
                    if (live(ch, j)) v[j] = load_pair<(YK_WG_NT != 0)>(iv + min(2u * (base + (u32)(j * T)), n - 2u));
            }
#pragma unroll
            for (int j = 0; j < R / 2; j++) {
                if (!live(ch, j)) continue;
                const u32 i0 = 2u * (base + (u32)(j * T));
                count(v[j].x, v[j].y, i0 + 1u < n); // (.xy is interval i0 only when i0 + 1 exists too: the clamped last pair)
                count(v[j].z, v[j].w, i0 < n);
            }
        }
        __syncthreads();
        // ---- this thread's bin, in event order: starts | ends << 16
        const uint4 c4 = bins[tid];
        u32 w = c4.x + c4.y + c4.z + c4.w;
        // (round 6) A regular interval cannot END at pmin: the ends of bin 0 are zero-length intervals at pmin, and those are
        // inert when c >= 1 and a regular interval starts there too — pushed at depth 0, popped alone in the heap in front of
        // the next push (src/stack.rs:72-89; tests/formulation.py::drop_inert_at_pmin) — so bin 0 forgets them, starts and
        // ends.  (SURVEY.md 8d's generator clamps a degenerate interval of a read covered only inside a window onto the
        // window's first position; as "an end at or before a" it sent the read to the sort.  Tested per slot in the count
        // this cost a tenth of the launch, profiles/r06/j_*; here it is one thread's three instructions.)
        if (tid == 0 && c >= 1 && (w >> 16) != 0u && (w & kField) > (w >> 16)) w -= (w >> 16) * (kEnd + 1u);
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
        const u32 n_incl = wave_incl_add(notyet), bad_w = wave_or((shallow ? 1u : 0u) | (spoiled ? 2u : 0u));
        if (lane == 63u) red[wv][2] = n_incl, red[wv][3] = bad_w;
        __syncthreads();
        u32 ntot = 0, any_bad = 0;
#pragma unroll
        for (u32 k = 0; k < (u32)NW; k++) ntot += red[k][2], any_bad |= red[k][3];
        healthy = any_bad == 0u && F > c && G > c;
        ra = pmin + (ntot & kField);
        rb = pmax - (ntot >> 16);
        vd.ends_ok = (any_bad & 2u) == 0u && F > c && G > c;
        vd.sh = sh, vd.ra = ra, vd.rb = rb;
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

__device__ __forceinline__ bool screen_wg_read(const SweepArgs &a, u32 r, u32 *tab, u32 (*red)[4], u32 *sc, WgVerdict &vd)
{
    const u64 o = a.off[r];
    return screen_wg_read<kWsR>(a, r, o, (u32)(a.off[r + 1] - o), a.len[r], tab, red, sc, vd);
}
// ... by the read's size: 4 / 8 / 12 / 16 slots per thread (round 6: configs[3]'s reads of 5 000-6 000 intervals fill six of the
// sixteen-slot build's eight pair groups; skipping the dead groups behind uniform branches was slower, see YK_WS_SKIP_DEAD)
// (o, n: the read's extent when the caller has it — plan_kernel's record —, n = 0xFFFFFFFF: loaded here)
__device__ __forceinline__ bool screen_wg_read_sized(const SweepArgs &a, u32 r, u64 o, u32 n, u32 *tab, u32 (*red)[4], u32 *sc)
{
    WgVerdict vd;
    if (n == 0xFFFFFFFFu) { // (uniform)
        o = a.off[r];
        n = (u32)(a.off[r + 1] - o);
    }
    const u32 len = a.len[r]; // (asked for in front of the intervals, needed behind them: the same round trip)
#if YK_WS_SIZED
    if (n <= (u32)(kWsT * 4)) return screen_wg_read<4>(a, r, o, n, len, tab, red, sc, vd);   // (uniform)
    if (n <= (u32)(kWsT * 8)) return screen_wg_read<8>(a, r, o, n, len, tab, red, sc, vd);
    if (n <= (u32)(kWsT * 12)) return screen_wg_read<12>(a, r, o, n, len, tab, red, sc, vd);
#endif
    return screen_wg_read<kWsR>(a, r, o, n, len, tab, red, sc, vd);
}
__device__ __forceinline__ bool screen_wg_read(const SweepArgs &a, u32 r, u32 *tab, u32 (*red)[4], u32 *sc)
{
    WgVerdict vd;
    return screen_wg_read(a, r, tab, red, sc, vd);
}

// ---- the FILTERED exact sweep of a read the screen could not decide (round 6) -----------------------------------------
// tests/formulation.py::unified_filtered_regions is the emulation (fuzzed against the oracle, exhaustive over small
// multisets: tests/test_formulation.py).  The reads the screen leaves are the reads yacrd looks for — on the generator's
// reads: chimeras, healthy at both ENDS and low at a junction somewhere inside — and until round 6 every one of them went
// through sweep_lds_read: two more passes over its intervals, the pile-trimming plan, a sort of what that keeps — for a
// chimera of 6 000 intervals the ~2 100 events at the junction PLUS as many stand-in keys as the depth in front of it, 8 192
// or 16 384 keys — and four sweep passes: 50-90 us by one workgroup, and the launch's tail is one such read.  Here the
// screen's own table, still in LDS, is read once more:
//   * D_i = starts - ends of the bins in front of bin i is the EXACT depth on entry to it (the map is monotone: the bins
//     are in event order), D_i - E_i the least depth any of its events sees.  D_i - E_i > c: the bin is SAFE — none of
//     its starts is low (src/stack.rs:83), all of its ends are flagged (:77-79).  A low start lies in an unsafe bin, and
//     the flagged end the reference pairs it with — the last one in front of it — lies in an unsafe bin too or is the
//     largest end of the nearest bin in front that holds an end;
//   * kept: the unsafe bins behind a's and in front of b's and, for each, the nearest bin in front that holds an end.
//     Their events (<= kWfCap, or the read is sweep_lds_read's after all) are collected in one more pass over the
//     intervals (from the L2: the screen has just read them), sorted in LDS and swept with their TRUE depths: the kept
//     events in front + a correction per bin (D_i minus the kept bins' net in front of it) — no stand-in keys;
//   * the ends are the screen's: (0, a) in front, (b, len) behind (WgVerdict::ends_ok); a run of low starts still open when
//     the keys end is closed by the tail.
// A shallow start at or behind b's bin, a zero-length interval in a kept bin whose start is low, a low start with no flagged
// end in front of it: false (nothing written) — the caller sorts the read.  True: the read's regions and their count are written.
constexpr int kWfCap = 4096; // kept events: eight keys per thread
__device__ __forceinline__ bool wg_filtered_read(const SweepArgs &a, u32 r, const WgVerdict &vd, u32 *tab, u32 *sc, u32 *keys,
                                                 unsigned long long *s_masks, const LaneConst &lc)
{
    constexpr int T = kWsT, W = kWsW, NW = T / 64;
    constexpr u32 kEnd = 1u << 16, kField = kEnd - 1u;
    static_assert(kWsBins == T, "one bin per thread");
    if (!vd.plain || !vd.ends_ok) return false; // (uniform)
    if (YK_EXP_FILT_STAGE == 1) return true;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    const u64 o = a.off[r];
    const u32 n = (u32)(a.off[r + 1] - o), len = a.len[r];
    const u32 pmin = vd.pmin, sh = vd.sh, span = vd.pmax - pmin, Tt = span - (u32)W;
    auto idx = [&](u32 x) {
        const u32 dx = x - pmin;
        return min(dx, (u32)W) + (dx >> sh) + __builtin_elementwise_sub_sat(dx, Tt);
    };
    uint4 *bins = reinterpret_cast<uint4 *>(tab);
    // ---- this thread's bin: counts, the depth on entry
    const uint4 c4 = bins[tid];
    const u32 w = c4.x + c4.y + c4.z + c4.w;
    const i32 S = (i32)(w & kField), E = (i32)(w >> 16);
    u32 tot;
    const u32 wex = block_excl_add<T>(w, sc, tot);
    const i32 D = (i32)(wex & kField) - (i32)(wex >> 16);
    const u32 ia = vd.ra - pmin, ib = idx(vd.rb);
    const bool unsafe = tid > ia && S + E > 0 && D - E <= c;
    const bool inside = tid > ia && tid < ib;
    const unsigned long long mu = __builtin_amdgcn_ballot_w64(unsafe && inside), me = __builtin_amdgcn_ballot_w64(inside && E > 0);
    if (lane == 0) s_masks[wv] = mu, s_masks[NW + wv] = me;
    // a start that may be low at or behind b: the closed form for the tail does not hold
    if (block_or<T>((unsafe && S > 0 && tid >= ib) ? 1u : 0u, sc)) return false; // (its barriers: the masks are written)
    // ---- kept: unsafe, or the nearest bin in front of an unsafe one that holds an end
    bool kept = unsafe && inside;
    if (inside && !unsafe && E > 0) {
        unsigned long long m = (mu | me) & ~((2ull << lane) - 1ull); // the bins behind this one, this wavefront's first
        u32 wq = wv;
        while (m == 0 && ++wq < (u32)NW) m = s_masks[wq] | s_masks[NW + wq];
        if (m != 0 && wq < (u32)NW) kept = ((s_masks[wq] >> __builtin_ctzll(m)) & 1ull) != 0;
    }
    u32 m_tot, net_tot;
    const u32 cex = block_excl_add<T>(kept ? (u32)(S + E) : 0u, sc, m_tot);
    if (m_tot == 0 || m_tot > (u32)kWfCap) return false; // (uniform)
    const u32 nex = block_excl_add<T>(kept ? (u32)(S - E) : 0u, sc, net_tot); // (two's complement)
    // per bin, where its counters were (every thread has read its own: the scans' barriers lie in between):
    // slot cursor, depth correction, kept
    bins[tid] = make_uint4(cex, (u32)(D - (i32)nex), kept ? 1u : 0u, 0u);
    if (YK_EXP_FILT_STAGE == 2) return true;
    u32 P = 2;
    while (P < m_tot) P <<= 1;
    for (u32 i = m_tot + tid; i < P; i += T) keys[i] = kNoKey;
    __syncthreads();
    // ---- the kept events, to the slots of their bins (eight loads in flight per thread)
    const uint2 *iv = a.iv + o;
    for (u32 i0 = tid; i0 < n; i0 += 8 * T) {
        uint2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = iv[min(i0 + (u32)(j * T), n - 1u)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (i0 + (u32)(j * T) >= n || v[j].y == 0u) continue; // ((0, 0) intervals are inert: nowhere in the table)
            const u32 is = idx(v[j].x), ie = idx(v[j].y);
            const bool ks = tab[4u * is + 2u] != 0u, ke = tab[4u * ie + 2u] != 0u;
            if (v[j].x == v[j].y) { // a zero-length interval: its two keys sort between the position's ends and its regular starts
                if (ks) {
                    const u32 at = atomicAdd(&tab[4u * is], 2u);
                    keys[at] = (v[j].x << kKeyShift) | 1u, keys[at + 1u] = (v[j].x << kKeyShift) | 2u;
                }
                continue;
            }
            if (ks) keys[atomicAdd(&tab[4u * is], 1u)] = (v[j].x << kKeyShift) | 3u;
            if (ke) keys[atomicAdd(&tab[4u * ie], 1u)] = v[j].y << kKeyShift;
        }
    }
    __syncthreads(); // (the keys are written)
    if (YK_EXP_FILT_STAGE == 3) return true;
    if (P >= 1024) hybrid_sort_lds<T>(keys, P, lc);
    else bitonic_sort_lds<T>(keys, P);
    if (YK_EXP_FILT_STAGE == 4) return true;
    // ---- the sweep: thread t owns the sorted keys [t K, t K + K); depth in front of a key = the kept keys in front
    // (starts - ends) + its bin's correction
    constexpr int KMAX = kWfCap / T;
    const u32 K = P >= (u32)T ? P / (u32)T : 1u;
    const u32 q0 = min(tid * K, m_tot), q1 = min(q0 + K, m_tot);
    // (the chunk's keys and their bins' corrections, once: the four passes below run on registers)
    u32 kx[KMAX];
    i32 cr[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; q++) {
        const bool real = q0 + (u32)q < q1;
        kx[q] = real ? keys[q0 + (u32)q] : kNoKey;
        cr[q] = real ? (i32)tab[4u * idx(kx[q] >> kKeyShift) + 1u] : 0;
    }
    u32 delta = 0;
#pragma unroll
    for (int q = 0; q < KMAX; q++) delta += kx[q] == kNoKey ? 0u : (kx[q] & 1u) ? 1u : 0xFFFFFFFFu;
    u32 dtot;
    const i32 depth_in = (i32)block_excl_add<T>(delta, sc, dtot);
    // last flagged end / last low start of the chunk (keys ascend: last = max); 0 = none (no end lies at position 0)
    u32 mf = 0, ml = 0;
    {
        i32 d = depth_in;
#pragma unroll
        for (int q = 0; q < KMAX; q++) {
            const u32 key = kx[q];
            const bool real = key != kNoKey, is_s = (key & 1u) != 0u, gt = d + cr[q] > c;
            mf = (real && !is_s && gt) ? key : mf;
            ml = (real && is_s && !gt) ? key : ml;
            d += real ? (is_s ? 1 : -1) : 0;
        }
    }
    u32 mf_t, ml_t;
    const u32 mf_in = block_excl_max<T>(mf, sc, mf_t), ml_in = block_excl_max<T>(ml, sc, ml_t);
    u32 cnt = 0, orphan = 0;
    {
        u32 tc = mf_in, cml = ml_in;
        i32 d = depth_in;
#pragma unroll
        for (int q = 0; q < KMAX; q++) {
            const u32 key = kx[q];
            const bool real = key != kNoKey, is_s = (key & 1u) != 0u, gt = d + cr[q] > c;
            const bool fl = real && !is_s && gt, low = real && is_s && !gt;
            cnt += (fl && cml > tc) ? 1u : 0u;
            // a low start with no flagged end in front of it (cannot happen behind a), or a zero-length interval whose start is
            // low (one with more than c open around it is an ordinary pair of keys): the caller sorts the read
            orphan |= (low && (tc == 0u || (key & 3u) == 1u)) ? 1u : 0u;
            tc = fl ? key : tc;
            cml = low ? key : cml;
            d += real ? (is_s ? 1 : -1) : 0;
        }
    }
    if (block_or<T>(orphan, sc)) return false; // (nothing is written yet)
    u32 g_closed;
    u32 pos = block_excl_add<T>(cnt, sc, g_closed);
    // ---- regions out: (0, a), the closed runs, the run the tail closes, (b, len)
    uint2 *slot = a.stage + (o + 2 * (u64)r);
    const u32 g0 = vd.ra != 0u ? 1u : 0u;
    if (cnt) {
        u32 tc = mf_in, cml = ml_in;
        i32 d = depth_in;
        pos += g0;
#pragma unroll
        for (int q = 0; q < KMAX; q++) {
            const u32 key = kx[q];
            const bool real = key != kNoKey, is_s = (key & 1u) != 0u, gt = d + cr[q] > c;
            const bool fl = real && !is_s && gt, low = real && is_s && !gt;
            if (fl && cml > tc) slot[pos++] = make_uint2(tc >> kKeyShift, cml >> kKeyShift);
            tc = fl ? key : tc;
            cml = low ? key : cml;
            d += real ? (is_s ? 1 : -1) : 0;
        }
    }
    if (tid == 0) {
        u32 g = g0 + g_closed;
        if (vd.ra != 0u) slot[0] = make_uint2(0u, vd.ra);
        if (ml_t > mf_t) slot[g++] = make_uint2(mf_t >> kKeyShift, ml_t >> kKeyShift);
        if (vd.rb != len) slot[g++] = make_uint2(vd.rb, len);
        a.counts[r] = g;
        if (a.prefilter == 2) atomicAdd(&a.ctr->prefiltered, 1u);
    }
    return true;
}
// SweepArgs.list / list_n: the class list; over_list / over_count: the reads the screen leaves to the sort.
#ifndef YK_WGK_OCC
#define YK_WGK_OCC 1
#endif
__global__ __launch_bounds__(kWsT, YK_WGK_OCC) void screen_wg_kernel(SweepArgs a)
{
    constexpr int NW = kWsT / 64;
    __shared__ __attribute__((aligned(16))) u32 tab[kWsBins * 4]; // four copies of every counter (by thread & 3)
    __shared__ u32 red[NW][4];
    __shared__ u32 sc[NW + 1];
    const u32 list_n = *a.list_n;
    for (u32 b = blockIdx.x; b < list_n; b += gridDim.x) { // (uniform)
        u32 r, n = 0xFFFFFFFFu;
        u64 o = 0;
        if (a.rec != nullptr) { // the entry's record: read, extent — one round trip instead of two
            const uint4 q = a.rec[b];
            r = q.w, n = q.z, o = ((u64)q.y << 32) | q.x;
        } else {
            r = a.list[b];
        }
        if (!screen_wg_read_sized(a, r, o, n, tab, red, sc) && threadIdx.x == 0) a.over_list[atomicAdd(a.over_count, 1u)] = r;
    }
}

// ---- what the screen leaves: table again, filtered exact sweep, whole-read sort (rounds 4-6) ----------------------------
// History, because every shape below was built, verified bit-exact and measured on configs[3] (10 000 reads of 5 000 ..
// 16 384 intervals, 2 % chimeras; profiles/r04 .. r06):
//   round 3  three kernels one after the other: the screen (screen_wg_kernel), sweep_lds_kernel<256, 8192> over what it left,
//            sweep_lds_kernel<1024, 32768> over what did not fit there: 0.128 + 0.061 + 0.065 ms — the two fallback kernels
//            one latency chain per read (two passes over the intervals, the trimming plan, a sort of 8 192 - 16 384 keys, four
//            sweep passes) with most of the device idle;
//   rounds 4-5  ONE persistent launch: a workgroup screened a static share of the class list, appended what it could not
//            decide to a QUEUE in global memory, then took reads off it — anybody's — through sweep_lds_read, and WAITED
//            for further appends until every workgroup had screened its share: 0.22 ms.  The wait needed the whole grid
//            resident (engines took turns, a bounded wait gave up, the engine ran the batch again), and up to 512 pollers
//            reading the same two words every 0.2 us are served one after the other at the memory side like the atomics they are;
//   round 6  (a) the share claimed from a counter: slower (5 000 more same-address atomics); (b) the queue drained by
//            compare-and-swap, nobody waiting: 1.0 ms (512 workgroups that finish together retry on one word); (c) no queue:
//            a workgroup sorts its own leftovers before it retires, the dispatcher balances: 0.24 — the launch's tail is
//            one fallback read, 90 us by one workgroup, whenever such a read sits in the last three quarters of the list;
//            (d) + the FILTERED exact sweep above for what the screen leaves: 0.21 — and the same launch with all that code
//            in it but never run took 0.163 where the screen alone takes 0.120 (128 VGPRs, 88 bytes of scratch written by
//            every workgroup, 74 KB of LDS); (e) TWO launches — the screen alone, one read per dispatcher-fed workgroup
//            (screen_wg_kernel: 0.120), then THIS kernel over the list of what it left (a fiftieth of the reads, one per
//            workgroup: the table built again 10 us, kept bins + keys 10, sort 11, sweep 8: 0.039): 0.16 ms for the class,
//            configs[3]'s step 0.31 -> 0.245 ms (with the device-wide class's launches on a side stream).  Kept: (e).
// The kernel still screens (its list may be a class list: YACRD_TEST / A/B paths hand it one), keeps what neither the
// screen nor the filtered sweep decides in a list of its own and takes THAT through sweep_lds_read<512, 16384> before it
// retires; what does not fit 16 384 events even after the trimming filter goes to over_list for the 1024-thread kernel
// (launched behind this one; usually nothing).
#ifndef YK_WS_FB_CAP
#define YK_WS_FB_CAP 16384
#endif
constexpr int kWsFbCap = YK_WS_FB_CAP; // events the in-kernel fallback sorts (64 KB of LDS; two workgroups per CU by registers anyway)
constexpr u32 kFusedShareMax = 32;
#ifndef YK_WG_OCC
#define YK_WG_OCC 4 // wavefronts per SIMD the register budget allows (two workgroups per CU)
#endif
struct ScreenFusedArgs {
    SweepArgs sweep;  // list / list_n: the class; over_list / over_count: beyond kWsFbCap; rej_*: degenerate reads
    u32 *n_fallback;  // reads the screen left to the in-kernel sort (a count for the host: Counters::fb_med)
    u32 share;        // consecutive list entries a workgroup screens per turn of its stride loop (1 .. kFusedShareMax)
};
__global__ __launch_bounds__(kWsT, YK_WG_OCC) void screen_wg_fused_kernel(ScreenFusedArgs f)
{
    constexpr int NW = kWsT / 64;
    __shared__ __attribute__((aligned(16))) u32 tab[kWsBins * 4];
    __shared__ u32 red[NW][4];
    __shared__ u32 sc[NW + 1];
    __shared__ u32 keys[kWsFbCap];
    __shared__ u32 s_mine[kFusedShareMax];
    __shared__ unsigned long long s_masks[2 * NW];
    const SweepArgs &a = f.sweep;
    const u32 tid = threadIdx.x;
    const u32 list_n = *a.list_n;
    // (the sorts' lane constants are made where a sort is — a read in fifty — and not kept alive across the screens: held
    //  from the kernel's start they were spilled there, 100 bytes of scratch per thread)
    auto lane_const = [&]() {
        LaneConst lc;
#pragma unroll
        for (int i = 0; i < 6; i++) lc.k[i] = (threadIdx.x & (1u << i)) ? 0xFFFFFFFFu : 0u;
        lc.k[6] = 0;
        lc.addr32 = ((threadIdx.x & 63u) ^ 32u) << 2;
        return lc;
    };
    // Two loops per share — the screens, then the sorts — and not the sort where the screen fails: the sort's registers and
    // the screen's live side by side otherwise (44 bytes of scratch per thread, profiles/r05/e_*).
    for (u32 base = blockIdx.x * f.share; base < list_n; base += gridDim.x * f.share) { // (uniform)
        u32 mine = 0;
        for (u32 b = base; b < min(base + f.share, list_n); b++) {
            const u32 r = a.list[b];
            WgVerdict vd;
            if (screen_wg_read(a, r, tab, red, sc, vd)) continue; // (uniform; ends with a barrier)
#if YK_EXP_SKIP_UNDECIDED // (timing experiments only: the code below is in the kernel and never runs)
            if (r != 0xFFFFFFFFu) continue;
#endif
#if !YK_EXP_NO_FILTERED
            const bool done = wg_filtered_read(a, r, vd, tab, sc, keys, s_masks, lane_const()); // (uniform)
            __syncthreads(); // (the table, sc, the masks and the keys are the next read's)
            if (done) continue;
#endif
            if (tid == 0) s_mine[mine] = r;
            mine++;
        }
        if (mine == 0) continue;
        if (tid == 0) atomicAdd(f.n_fallback, mine);
        __syncthreads();
        for (u32 k = 0; k < mine; k++) {
#if !YK_EXP_NO_FALLBACK // (timing experiments only: the reads are dropped)
            sweep_lds_read<kWsT, kWsFbCap>(a, s_mine[k], keys, sc, lane_const());
#endif
            __syncthreads(); // keys / sc reused
        }
    }
}

} // namespace yk
