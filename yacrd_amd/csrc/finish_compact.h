// finish_compact.h — the follow-on kernel: finish what the screen deferred, then scan + compact + classify.
//
// One workgroup of 1024 threads per 1024 consecutive reads, one thread per read.
//   Phase A  the reads of this slab that the healthy-read screen (sweep_wave.h) marked in counts[] are
//            sorted here: listed in LDS, one read per wavefront and turn.  (Rounds 1-2 gave them a launch
//            of their own —
//            sweep_deferred_kernel: one read per wavefront on 64 lanes, 11 us of a 30 us batch on
//            configs[1] — or, for large batches, a compaction launch plus a class launch whose grid the
//            host had to size.)  No lists in global memory, no host-sized grid, one dispatch less.
//   Phase B  bad_offsets = exclusive scan of the per-read region counts (single pass, decoupled look-back
//            over workgroup aggregates), regions copied into the CSR, type_of_read
//            (reference src/editor/mod.rs:85-100).
#pragma once
#include "plan_compact.h"
#include "sweep_lds.h"
#include "sweep_wave.h"
#include "sweep_filtered.h"

namespace yk {

struct CompactArgs2 {
    SweepArgs sweep;      // off / iv / len / cov / prefilter / stage / counts / rej_list / rej_count / ctr
    u64 *scan_state;      // one word per workgroup: flag << 62 | value (1 = aggregate, 2 = inclusive prefix)
    u32 n_reads;
    double not_cov;
    u64 *bad_offsets;     // [R+1]
    uint2 *bad_regions;
    u64 region_cap;
    uint8_t *read_type;   // [R]
    Counters *host_ctr;   // pinned host copy of the counter block, written by the slab that ends the batch (or null)
};

constexpr int kFinishWaves = kScanBlock / 64;
#ifndef YK_FINISH_ORDER_REL
#define YK_FINISH_ORDER_REL __ATOMIC_RELAXED // (A/B: __ATOMIC_RELEASE / __ATOMIC_ACQUIRE, see the look-back below)
#define YK_FINISH_ORDER_ACQ __ATOMIC_RELAXED
#endif
#ifndef YK_FINISH_OCC
#define YK_FINISH_OCC 8 // register budget of the kernel as wavefronts per SIMD (512 / 8 = 64 VGPRs: two workgroups per CU)
#endif

// The marked reads of one workgroup's slab, listed in LDS by their index inside the slab, are sorted whole,
// one read per wavefront on all 64 lanes: 4 keys per lane up to 128 intervals, 8 up to 256 (56 registers: two
// workgroups per CU; the 16 / 32-lane layouts with the bin filter in front need 96 and a filter table in LDS,
// which left one workgroup per CU and made the scan + compaction of EVERY slab twice as slow).
// One item = one read.  (As a function of its own — not inlined, so that the loop over the items cannot hoist
// per-lane constants out and hold them next to the sort's keys — it measured the same or slower: YK_FINISH_INLINE.)
#ifndef YK_FINISH_INLINE
#define YK_FINISH_INLINE __forceinline__
#endif
template <int K>
__device__ YK_FINISH_INLINE void finish_item(const u64 *off, const uint2 *iv, const u32 *len, uint2 *stage, u32 *counts,
                                             u32 *rej_list, u32 *rej_count, Counters *ctr, u32 cov, u32 rr, u64 o, u32 n,
                                             u32 length)
{
    SweepArgs fa;
    fa.off = off, fa.iv = iv, fa.len = len, fa.list = nullptr, fa.list_n = nullptr, fa.first = 0, fa.cov = cov;
    fa.prefilter = 0, fa.stage = stage, fa.counts = counts, fa.rej_list = rej_list, fa.rej_count = rej_count;
    fa.over_list = nullptr, fa.over_count = nullptr, fa.ctr = ctr;
    const LaneConst lc = make_lane_const(lane_id());
    sweep_group_read<64, K, 0, kFinishWaves>(iv + o, n, length, cov, true, rr, fa, lc);
}
struct MarkedRead { // what the thread that found the mark already knows about the read (saves the sort one round trip)
    u64 o;
    u32 n, len, idx, pad;
};
#ifndef YK_FINISH_FILTERED
// Phase A through the filtered exact sweep (sweep_filtered.h), two marked reads per wavefront and turn, what the filter leaves sorted
// whole behind a barrier.  Built, bit-exact (104 parity tests, 1.6 M fuzzed reads) and NOT faster: configs[1] one batch
// at a time 65.2 -> 65.8 us with 128 registers (109 used, no scratch, one workgroup per CU), 70.6 at the kernel's 64 (68 bytes of scratch);
// pipelined 24.7 -> 26.0 / 25.2; at sigma = 300 53.5 -> 55.7 / 62.1 (profiles/r06/P_finish.log).  A slab's two dozen marked reads are two
// turns of sixteen wavefronts either way, and the filter's table + the barrier cost what the shorter sweep saves.  The plain sorts with
// 128 registers (-DYK_FINISH_OCC=4: 99 used, no scratch) measured 25.3 us pipelined / 64.1 one at a time (24.7 / 65.2), 51.5 / 102.7
// at sigma = 300 (53.5 / 105.2), 390 000 reads 87.4 / 134.9 (87.6 / 129.9): a draw, left at 8.
#define YK_FINISH_FILTERED 0
#endif
template <int LANES, int WPB, int TABW>
__device__ __forceinline__ bool filtered_turn(const SweepArgs &a, bool active, u32 r, u64 o, u32 n, u32 len, const LaneConst &lc);
// s_fb / s_nfb (YK_FINISH_FILTERED): the slab's reads the filter did not take, sorted whole behind a barrier (s_nfb starts at 0)
__device__ __forceinline__ void finish_marked(const SweepArgs &a, u32 bid, u32 n_marked, const MarkedRead *list, unsigned short *s_fb, u32 *s_nfb)
{
#if YK_FINISH_FILTERED
    constexpr int kTabWords = 2 * (2 * kScreenWindow + 32) * 4; // two 32-lane groups
    const u32 wv = threadIdx.x >> 6, lane = lane_id();
    const LaneConst lcf = make_lane_const(lane);
    for (u32 p0 = wv * 2u; p0 < n_marked; p0 += (u32)kFinishWaves * 2u) { // (uniform in the wavefront)
        const u32 p = p0 + (lane >> 5);
        const bool have = p < n_marked;
        const MarkedRead m = list[have ? p : p0];
        const bool done = filtered_turn<32, kFinishWaves, kTabWords>(a, have, bid * kScanBlock + m.idx, m.o, m.n, m.len, lcf);
        if (have && !done && (lane & 31u) == 31u) s_fb[atomicAdd(s_nfb, 1u)] = (unsigned short)p;
    }
    __syncthreads();
    const u32 nfb = *s_nfb;
    for (u32 i = wv; i < nfb; i += (u32)kFinishWaves) { // (uniform in the wavefront)
        const MarkedRead m = list[s_fb[i]];
#else
    for (u32 i = threadIdx.x >> 6; i < n_marked; i += (u32)kFinishWaves) { // (uniform in the wavefront)
        const MarkedRead m = list[i];
#endif
        const u32 rr = bid * kScanBlock + m.idx;
        if (m.n > 128u)
            finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, m.o, m.n, m.len);
        else
            finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, m.o, m.n, m.len);
    }
}

__global__ __launch_bounds__(kScanBlock, YK_FINISH_OCC) void finish_compact_kernel(CompactArgs2 c)
{
    __shared__ u32 sc[kScanBlock / 64];
    __shared__ u32 s_bid;
    __shared__ u64 s_base;
    __shared__ u32 s_n;
    __shared__ unsigned long long s_iv;
    __shared__ MarkedRead s_list[kScanBlock]; // the marked reads of the slab
#if YK_FINISH_FILTERED
    __shared__ unsigned short s_fb[kScanBlock];
#else
    unsigned short *s_fb = nullptr;
#endif
    __shared__ u32 s_nfb;
    const SweepArgs &a = c.sweep;
    Counters *ctr = a.ctr;
    if (threadIdx.x == 0) {
        s_bid = atomicAdd(&ctr->scan_ticket, 1u);
        s_n = 0;
        s_iv = 0;
        s_nfb = 0;
    }
    __syncthreads();
    const u32 bid = s_bid, lane = lane_id();

    // ---- phase A: the reads the screen left marked.  (Nothing of phase B is kept in registers across
    // it: the sort needs them all.)
    {
        const u32 r0 = bid * kScanBlock + threadIdx.x;
        const bool marked = r0 < c.n_reads && a.counts[r0] == kDeferredMark;
        if (__builtin_amdgcn_ballot_w64(marked) != 0) { // (uniform in the wavefront)
            const u64 o0 = marked ? a.off[r0] : 0;
            const u32 n = marked ? (u32)(a.off[r0 + 1] - o0) : 0u;
            const u32 len0 = marked ? a.len[r0] : 0u;
            const u64 m = __builtin_amdgcn_ballot_w64(marked);
            u32 base = 0;
            if (lane == (u32)__builtin_ctzll(m)) base = atomicAdd(&s_n, (u32)__builtin_popcountll(m));
            base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
            if (marked) {
                const u32 pos = base + (u32)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                MarkedRead mr;
                mr.o = o0, mr.n = n, mr.len = len0, mr.idx = threadIdx.x, mr.pad = 0;
                s_list[pos] = mr;
            }
            u64 iv = n; // intervals of the marked reads, for the roofline's exact byte count
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) iv += __shfl_xor(iv, d, 64);
            if (lane == 0) atomicAdd(&s_iv, (unsigned long long)iv);
        }
    }
    __syncthreads();
    if (s_n) { // uniform in the workgroup
#ifndef YK_FINISH_SKIP
        finish_marked(a, bid, s_n, s_list, s_fb, &s_nfb);
#endif
        __syncthreads(); // (global stores of this workgroup's wavefronts are visible to each other after it)
        if (threadIdx.x == 0) {
            // returning atomics, waited for: they are performed before this slab publishes its aggregate below, so the
            // slab that ends the batch — whose look-back has seen every aggregate — reads final counters.  (Issued
            // before the sorts and waited for after them they cost three registers across the sort: spills, +10 us
            // on 2 M reads.)
            const u32 t0 = atomicAdd(&ctr->deferred, s_n);
            const unsigned long long t1 = atomicAdd((unsigned long long *)&ctr->deferred_iv, s_iv);
            asm volatile("" ::"v"(t0), "v"(t1));
        }
    }
    const u32 r = bid * kScanBlock + threadIdx.x;
    const bool in = r < c.n_reads;
    u32 g = in ? a.counts[r] : 0u;
    if (g == kDeferredMark) g = 0u; // (a marked read of more than 256 intervals cannot exist)
    const u64 off_r = in ? a.off[r] : 0;
    const u32 L = in ? a.len[r] : 0u;
    // the screen's closed form (device_common.h: kClosedForm): regions (0, a) and (b, len), whichever is not empty
    const bool closed = g == kClosedForm;
    uint2 ab = make_uint2(0u, L);
    if (closed) {
        ab = a.closed[r];
        g = (ab.x != 0u ? 1u : 0u) + (ab.y != L ? 1u : 0u);
    }

    // ---- phase B: scan, compaction, classification
    u32 tot;
    const u32 local = block_excl_add<kScanBlock>(g, sc, tot);
    if (threadIdx.x < 64) { // decoupled look-back, 64 predecessors per round trip
        constexpr u64 kAgg = 1ull << 62, kPre = 2ull << 62, kVal = (1ull << 62) - 1;
        u64 base = 0;
        if (bid > 0) {
            if (lane == 0)
                // (Relaxed on purpose.  The hand-over of the counters below relies on this slab's two counter atomics being
                // performed before this store: they are returning atomics issued by this very lane and waited for, i.e.
                // done at the memory side.  Spelled as release here + acquire in the look-back — ADVICE r3 — the
                // compiler emits an L2 write-back / invalidate per store / poll: the kernel went from 0.150 to 0.344 ms
                // on configs[2], profiles/r04/b_ab_split_follow_on.log; on the short batches this kernel still serves:
                // 20.6 -> 23.0 us for 100 000 reads, 33.7 -> 54.5 us for 390 000, the pipelined batch 26.6 -> 30.0 us,
                // profiles/r04/m_ab_release_acquire_short_batches.log; -DYK_FINISH_ORDER_REL=__ATOMIC_RELEASE
                // -DYK_FINISH_ORDER_ACQ=__ATOMIC_ACQUIRE builds it.)
                __hip_atomic_store(&c.scan_state[bid], kAgg | tot, YK_FINISH_ORDER_REL, __HIP_MEMORY_SCOPE_AGENT);
            for (i32 hi = (i32)bid - 1;; hi -= 64) {
                const i32 idx = hi - (i32)lane; // lane 0 looks at the nearest predecessor
                u64 v, pre;
                for (;;) { // until the window holds no empty entry before its nearest prefix
                    v = idx >= 0 ? __hip_atomic_load(&c.scan_state[idx], YK_FINISH_ORDER_ACQ, __HIP_MEMORY_SCOPE_AGENT)
                                 : kPre; // before the first workgroup: prefix 0
                    pre = __builtin_amdgcn_ballot_w64((v >> 62) == 2);
                    const u64 before = pre ? ((pre & (0 - pre)) - 1ull) : ~0ull; // lanes nearer than it
                    if ((__builtin_amdgcn_ballot_w64((v >> 62) == 0) & before) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const u32 first_pre = pre ? (u32)__builtin_ctzll(pre) : 64u;
                u64 part = lane <= first_pre ? (v & kVal) : 0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                base += part;
                if (pre) break;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&c.scan_state[bid], kPre | (base + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
            if ((u64)(bid + 1) * kScanBlock >= c.n_reads) ctr->total_regions = base + tot;
        }
        // The counters go home from here: the slab that ends the batch has seen every other slab's aggregate, hence
        // every slab is past its phase A (the only writer of counters besides the totals set right here), and the
        // overflow flag follows from the total.  One wavefront, 512 bytes, no copy command behind the kernel.
        if (c.host_ctr && (u64)(bid + 1) * kScanBlock >= c.n_reads) {
            const u64 total = (u64)__shfl((long long)(base + tot), 0, 64);
            const u32 *src = reinterpret_cast<const u32 *>(ctr);
            u32 *dst = reinterpret_cast<u32 *>(c.host_ctr);
            constexpr u32 kWords = (u32)(sizeof(Counters) / 4);
            for (u32 i = lane; i < kWords; i += 64u) {
                u32 w = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i == (u32)(offsetof(Counters, total_regions) / 4)) w = (u32)total;
                if (i == (u32)(offsetof(Counters, total_regions) / 4) + 1u) w = (u32)(total >> 32);
                if (i == (u32)(offsetof(Counters, region_overflow) / 4)) w = total > c.region_cap ? 1u : 0u;
                dst[i] = w;
            }
        }
    }
    __syncthreads();
    if (in) {
        const u64 dst = s_base + local;
        c.bad_offsets[r] = dst;
        if (r == c.n_reads - 1) c.bad_offsets[c.n_reads] = dst + g;

        u32 bad = 0;
        bool middle = false;
        const bool fits = dst + g <= c.region_cap;
        if (closed) { // (neither region lies in the middle: the first begins at 0, the second ends at len)
            u32 k = 0;
            if (ab.x != 0u && fits) c.bad_regions[dst + k++] = make_uint2(0u, ab.x);
            if (ab.y != L && fits) c.bad_regions[dst + k] = make_uint2(ab.y, L);
            bad = ab.x + (L - ab.y);
        } else {
            const uint2 *slot = a.stage + (off_r + 2 * (u64)r);
            for (u32 k = 0; k < g; k++) {
                const uint2 v = slot[k];
                if (fits) c.bad_regions[dst + k] = v;
                bad += v.y - v.x;
                middle |= (v.x != 0u) & (v.y != L);
            }
        }
        if (!fits) atomicOr(&ctr->region_overflow, 1u);
        c.read_type[r] = (uint8_t)classify(bad, middle, L, c.not_cov);
    }
}


// ==== the follow-on step as TWO kernels (round 4): long batches ========================================
// finish_compact_kernel above does both phases in 1024-thread workgroups at a 64-register budget.  On long batches
// (configs[2]: 0.150 ms, configs[4]: 0.36 ms = a fifth of the step) that costs twice: its sorts spill (40 bytes of
// scratch per thread — all 2 M threads of configs[2] write 12 of them: the 97 MB WRITE_SIZE / 199 MB FETCH_SIZE
// of round 3's PMC against ~20 / 124 MB of data), and a slab's ~24 marked reads keep 16 wavefronts busy for one
// turn and 8 for a second while the slab's scan waits (the sorts are ~600 dependent instructions per read: 29 M of
// them for configs[2]'s 47 600 reads = 47 us of VALU time when every SIMD issues; the kernel needed ~100).
// For batches of kPlanSmallReads reads and more the engine launches instead:
//   deferred_sweep_kernel   512 threads per slab of 2048 reads: the slab's marked reads are listed in LDS, the long
//                           ones first, and sorted one per wavefront and turn — the same 64-lane sorts, but four
//                           wavefronts take six turns each instead of sixteen taking one and a half, eight
//                           workgroups share a CU, and nothing else waits for them;
//   scan_compact_kernel     1024 threads x 4 consecutive reads: 16-byte loads of counts / lengths / closed forms,
//                           decoupled look-back over 4096-read aggregates (a quarter of the tickets and of the
//                           look-back chain), 16-byte stores of bad_offsets, one 4-byte store of four read types;
//                           no scratch.
// Short batches keep the one-dispatch form: a dispatch more costs more than the phases gain there (configs[1], one
// batch at a time: 69.5 us against 93 us, profiles/r04/b_ab_split_follow_on.log).
// (First attempt, same log: the marked reads through the sorting build's code — sweep_group_read<32 | 16, 16>, two or
// four reads per wavefront behind the bin filter — 0.190 ms for configs[2]'s 47 600 reads: per read that code is
// no cheaper than the 64-lane sort, and pairs / quadruples fill badly from a list of two dozen.)
// (Round 4 also sent the marked reads through the pile-trimming filter (§3.5) and a SHORT register sort first: bit-exact
// and no faster — the filter's two counting passes, its plan and the scatter cost what the shorter sort saved, 43.6 M VALU
// instructions against 40.3 M for configs[2]'s 47 608 reads, profiles/r04/g_ab_trimmed_deferred_sweep.log; the code went
// when round 5's filtered exact sweep, sweep_filtered.h, took its place.)
#ifndef YK_DEFER_SWEEP_OCC
#define YK_DEFER_SWEEP_OCC 6 // wavefronts per SIMD the sweep kernel's register budget allows (80 VGPRs)
#endif

// One turn of the filtered sweep: this lane group's read (active: it has one) — loads and tests as screen_reads', then
// filtered_group_sweep.  True: done (uniform in the group).
template <int LANES, int WPB, int TABW = kScreenTabWords>
__device__ __forceinline__ bool filtered_turn(const SweepArgs &a, bool active, u32 r, u64 o, u32 n, u32 len, const LaneConst &lc)
{
    constexpr int K = 16;
    const u32 lane = lane_id(), lig = lane & (u32)(LANES - 1);
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    const bool two = active && n >= 2u;
    const uint2 *src = two ? a.iv + o : reinterpret_cast<const uint2 *>(a.off);
    const u32 last2 = two ? n - 2u : 0u;
    uint4 v[K / 4];
#pragma unroll
    for (int j = 0; j < K / 4; j++) v[j] = load_pair(src + min(2u * (lig + (u32)LANES * j), last2));
    u32 smin = v[0].x, emax = v[0].y, smax = v[0].x;
    i32 tmin = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < K / 4; j++) {
        smin = min(smin, min(v[j].x, v[j].z));
        smax = max(smax, max(v[j].x, v[j].z));
        emax = max(emax, max(v[j].y, v[j].w));
        tmin = min(tmin, min((i32)(v[j].y - v[j].x), (i32)(v[j].w - v[j].z)));
    }
    const int last_addr = (int)((lane | (u32)(LANES - 1)) << 2);
    const u32 pmin = (u32)__builtin_amdgcn_ds_bpermute(last_addr, (int)gscan_min<LANES>(smin));
    const u32 pmax = (u32)__builtin_amdgcn_ds_bpermute(last_addr, (int)gscan_max<LANES>(emax));
    const bool irregular = !two || pmax > min(len, kMaxKeyPos) || smax > kMaxKeyPos || tmin < (i32)kScreenWindow;
    const bool girr = group_any<LANES>(__builtin_amdgcn_ballot_w64(irregular));
    const u32 n_eff = girr ? 0u : n;
    bool real0[K / 4], real1[K / 4];
#pragma unroll
    for (int j = 0; j < K / 4; j++) {
        const u32 i0 = 2u * (lig + (u32)LANES * j);
        real0[j] = i0 + 1u < n_eff; // (.xy is interval i0 only when i0 + 1 exists too: see screen_reads)
        real1[j] = i0 < n_eff;
    }
    return filtered_group_sweep<LANES, WPB, TABW>(a, v, real0, real1, r, o, n, len, c, pmin, pmax, !girr, lc);
}
// ==== the follow-on step's sweep over a LIST of the marked reads (round 5) ===============================================
// Round 4's deferred_sweep_kernel (and this round's first filtered form of it) gave every slab of 2048 reads to one
// workgroup: a slab's ~49 marked reads are 25 pairs for eight wavefronts — four turns of which the last keeps one wavefront
// busy — then a barrier, then the ~7 reads the filter did not take; 2 442 such workgroups ran in 3.2 rounds on the 768 that
// fit.  PMC: 46 % of the wave slots occupied on average (profiles/r05/b_*).  Here the marks are LISTED first
// (mark_list_kernel: 4096 reads per workgroup, one returning atomic per workgroup on one of kDeferShards counters: shard =
// workgroup & 255, a cache line each; 9 us for 5 M reads) and the list is dealt out evenly: workgroup w of a grid that is
// resident as a whole takes entries [w, w + 1) * ceil(total / grid) of the shards laid end to end, 512 at a time, two per
// wavefront and turn through the filtered sweep, what the filter leaves one per wavefront at the end of every 512.
// configs[2]: follow-on step 0.126-0.130 -> 0.107-0.109 ms; configs[4]: 0.250-0.262 -> 0.237-0.258 (profiles/r05/q_*;
// 512 / 256 threads and occupancy 6 / 5: the same).  The marks stay the truth: a redo of the follow-on step lists what is
// still marked.
constexpr u32 kDeferShards = 256, kDeferShardStride = 16; // (words between two shards' counters)
struct DeferList {
    u32 *list;      // [kDeferShards][shard_cap] read ids
    u32 *count;     // [kDeferShards * kDeferShardStride]
    u32 shard_cap;
};
constexpr int kMarkThreads = 1024, kMarkReads = 4 * kMarkThreads;
__global__ __launch_bounds__(kMarkThreads) void mark_list_kernel(const u32 *__restrict__ counts, u32 n_reads, DeferList dl)
{
    __shared__ u32 s_ids[kMarkReads];
    __shared__ u32 s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * kMarkReads + threadIdx.x * 4u;
    u32 g[4] = {0, 0, 0, 0};
    if (r0 + 4 <= n_reads) {
        const uint4 g4 = *reinterpret_cast<const uint4 *>(counts + r0); // (engine-owned: 256-byte aligned)
        g[0] = g4.x, g[1] = g4.y, g[2] = g4.z, g[3] = g4.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = r0 + k < n_reads ? counts[r0 + k] : 0u;
    }
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) mine += g[k] == kDeferredMark ? 1u : 0u;
    if (__builtin_amdgcn_ballot_w64(mine != 0) != 0) { // (uniform in the wavefront; 2-3 % of the reads are marked)
        u32 incl = mine; // wavefront-inclusive count, then one LDS atomic per wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = (u32)__shfl_up((int)incl, d, 64);
            if ((int)lane_id() >= d) incl += t;
        }
        u32 base = 0;
        if (lane_id() == 63u) base = atomicAdd(&s_n, incl);
        base = (u32)__shfl((int)base, 63, 64);
        u32 at = base + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (g[k] == kDeferredMark) s_ids[at++] = (u32)(r0 + k);
    }
    __syncthreads();
    const u32 n = s_n;
    if (n == 0) return; // (uniform)
    const u32 shard = blockIdx.x & (kDeferShards - 1u);
    if (threadIdx.x == 0) s_base = atomicAdd(dl.count + shard * kDeferShardStride, n);
    __syncthreads();
    const u32 b0 = s_base;
    for (u32 i = threadIdx.x; i < n; i += kMarkThreads)
        if (b0 + i < dl.shard_cap) dl.list[(size_t)shard * dl.shard_cap + b0 + i] = s_ids[i];
}

#ifndef YK_LIST_THREADS
#define YK_LIST_THREADS 512
#endif
#ifndef YK_LIST_OCC
#define YK_LIST_OCC YK_DEFER_SWEEP_OCC
#endif
constexpr int kListThreads = YK_LIST_THREADS;
__global__ __launch_bounds__(kListThreads, YK_LIST_OCC) void deferred_list_kernel(SweepArgs a, DeferList dl)
{
    constexpr u32 kWaves = kListThreads / 64;
    constexpr int kTabWords = 2 * (2 * kScreenWindow + 32) * 4; // two 32-lane groups
    static_assert(kDeferShards <= (u32)kListThreads, "one thread per shard counter");
    __shared__ u32 s_pre[kDeferShards + 1];
    __shared__ u32 sc[kWaves + 1];
    __shared__ uint4 s_list[kListThreads]; // read, intervals, length
    __shared__ unsigned short s_fb[kListThreads];
    __shared__ u32 s_nfb;
    __shared__ unsigned long long s_iv;
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    u32 total;
    {
        const u32 cnt = tid < kDeferShards ? min(dl.count[tid * kDeferShardStride], dl.shard_cap) : 0u;
        const u32 ex = block_excl_add<kListThreads>(cnt, sc, total);
        if (tid < kDeferShards) s_pre[tid] = ex;
        if (tid == 0) s_pre[kDeferShards] = total;
    }
    const u32 per = (total + gridDim.x - 1u) / gridDim.x;
    const u32 lo = (u32)min((u64)total, (u64)blockIdx.x * per), hi = (u32)min((u64)total, (u64)lo + per);
    if (lo >= hi) return; // (uniform)
    const LaneConst lcf = make_lane_const(lane);
    if (tid == 0) s_iv = 0;
    for (u32 c0 = lo; c0 < hi; c0 += (u32)kListThreads) { // (uniform)
        __syncthreads(); // (s_pre is written; the last round's lists are done with — every wavefront has read its s_nfb)
        if (tid == 0) s_nfb = 0; // (behind the barrier: ADVICE r5 — in front of it, a wavefront still on its way to the last round's
                                 //  `nfb = s_nfb` could read 0 and skip its fallback reads; the adds start behind the next barrier)
        const u32 e = c0 + tid, n_here = min((u32)kListThreads, hi - c0);
        if (e < hi) {
            u32 s = 0; // the shard that holds entry e: the last one whose first entry is <= e
#pragma unroll
            for (u32 step = kDeferShards / 2; step > 0; step >>= 1)
                if (s_pre[s + step] <= e) s += step;
            const u32 r = dl.list[(size_t)s * dl.shard_cap + (e - s_pre[s])];
            const u64 o = a.off[r];
            const u32 n = (u32)(a.off[r + 1] - o);
            s_list[tid] = make_uint4(r, n, a.len[r], 0u);
            atomicAdd(&s_iv, (unsigned long long)n); // (intervals of the marked reads, for the roofline's exact byte count)
        }
        __syncthreads();
        for (u32 p0 = wv * 2u; p0 < n_here; p0 += kWaves * 2u) { // (uniform in the wavefront)
            const u32 p = p0 + (lane >> 5);
            const bool have = p < n_here;
            const uint4 q = s_list[have ? p : p0];
            const u64 o = a.off[q.x]; // (in the L2: the listing has just read it)
            const bool done = filtered_turn<32, (int)kWaves, kTabWords>(a, have, q.x, o, q.y, q.z, lcf);
            if (have && !done && (lane & 31u) == 31u) s_fb[atomicAdd(&s_nfb, 1u)] = (unsigned short)p;
        }
        __syncthreads();
        const u32 nfb = s_nfb;
        for (u32 i = wv; i < nfb; i += kWaves) { // (uniform in the wavefront): sorted whole, one read per wavefront
            const uint4 q = s_list[s_fb[i]];
            const u64 o = a.off[q.x];
            if (q.y > 128u) finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, q.x, o, q.y, q.z);
            else finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, q.x, o, q.y, q.z);
        }
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd((unsigned long long *)&a.ctr->deferred_iv, s_iv);
        atomicAdd(&a.ctr->deferred, hi - lo);
    }
}

constexpr int kScanThreads = 1024;
// PER consecutive reads per thread.  4; 8 (half the tickets and half the look-back chain again: configs[4] 1 221 -> 611
// workgroups) measured 65.1 us against 55.6 on configs[4], profiles/r05/m_*: the engine instantiates <4> only.
template <int PER>
__global__ __launch_bounds__(kScanThreads) void scan_compact_kernel(CompactArgs2 c)
{
    constexpr int kScanPer = PER, kScanReads = kScanThreads * PER;
    static_assert(PER % 4 == 0 && kScanReads % kScanBlock == 0, "16-byte loads; the control block holds one scan word per 1024 reads: more than this kernel's workgroups use");
    __shared__ u32 sc[kScanThreads / 64];
    __shared__ u32 s_bid;
    __shared__ u64 s_base;
    const SweepArgs &a = c.sweep;
    Counters *ctr = a.ctr;
    if (threadIdx.x == 0) s_bid = atomicAdd(&ctr->scan_ticket, 1u);
    __syncthreads();
    const u32 bid = s_bid, lane = lane_id();
    const u64 r0 = (u64)bid * kScanReads + threadIdx.x * (u32)kScanPer; // (u64: the last slab of a batch of nearly 2^32 reads)
    u32 g[kScanPer], L[kScanPer];
    uint2 ab[kScanPer];
    bool closed[kScanPer];
    // 16-byte loads where the four reads exist (counts / closed are the engine's own buffers; the lengths are the
    // caller's: their alignment is looked at)
    const bool vec = r0 + kScanPer <= c.n_reads && (reinterpret_cast<uintptr_t>(a.len) & 15u) == 0;
    if (vec) {
#pragma unroll
        for (int q = 0; q < kScanPer / 4; q++) {
            const uint4 g4 = *reinterpret_cast<const uint4 *>(a.counts + r0 + 4 * q);
            const uint4 l4 = *reinterpret_cast<const uint4 *>(a.len + r0 + 4 * q);
            g[4 * q] = g4.x, g[4 * q + 1] = g4.y, g[4 * q + 2] = g4.z, g[4 * q + 3] = g4.w;
            L[4 * q] = l4.x, L[4 * q + 1] = l4.y, L[4 * q + 2] = l4.z, L[4 * q + 3] = l4.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPer; k++) {
            const bool in = r0 + k < c.n_reads;
            g[k] = in ? a.counts[r0 + k] : 0u;
            L[k] = in ? a.len[r0 + k] : 0u;
        }
    }
    bool any_closed = false;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (g[k] == kDeferredMark) g[k] = 0u; // (cannot be left: the deferred sweep ran before)
        closed[k] = g[k] == kClosedForm;
        any_closed |= closed[k];
        ab[k] = make_uint2(0u, L[k]);
    }
    if (any_closed) { // the screen's closed form (device_common.h: kClosedForm)
        if (vec) {
#pragma unroll
            for (int q = 0; q < kScanPer / 2; q++) {
                const uint4 c0 = *reinterpret_cast<const uint4 *>(a.closed + r0 + 2 * q);
                if (closed[2 * q]) ab[2 * q] = make_uint2(c0.x, c0.y);
                if (closed[2 * q + 1]) ab[2 * q + 1] = make_uint2(c0.z, c0.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kScanPer; k++)
                if (closed[k]) ab[k] = a.closed[r0 + k];
        }
    }
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (closed[k]) g[k] = (ab[k].x != 0u ? 1u : 0u) + (ab[k].y != L[k] ? 1u : 0u);
        mine += g[k];
    }
    u32 tot;
    u32 local = block_excl_add<kScanThreads>(mine, sc, tot);
    if (threadIdx.x < 64) { // decoupled look-back, 64 predecessors per round trip (as in finish_compact_kernel)
        constexpr u64 kAgg = 1ull << 62, kPre = 2ull << 62, kVal = (1ull << 62) - 1;
        u64 base = 0;
        if (bid > 0) {
            if (lane == 0) __hip_atomic_store(&c.scan_state[bid], kAgg | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (i32 hi = (i32)bid - 1;; hi -= 64) {
                const i32 idx = hi - (i32)lane;
                u64 v, pre;
                for (;;) {
                    v = idx >= 0 ? __hip_atomic_load(&c.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kPre;
                    pre = __builtin_amdgcn_ballot_w64((v >> 62) == 2);
                    const u64 before = pre ? ((pre & (0 - pre)) - 1ull) : ~0ull;
                    if ((__builtin_amdgcn_ballot_w64((v >> 62) == 0) & before) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const u32 first_pre = pre ? (u32)__builtin_ctzll(pre) : 64u;
                u64 part = lane <= first_pre ? (v & kVal) : 0;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                base += part;
                if (pre) break;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&c.scan_state[bid], kPre | (base + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
            if ((u64)(bid + 1) * kScanReads >= c.n_reads) ctr->total_regions = base + tot;
        }
        // the counters go home from the slab that ends the batch (every other writer of counters is a kernel that
        // finished before this one started)
        if (c.host_ctr && (u64)(bid + 1) * kScanReads >= c.n_reads) {
            const u64 total = (u64)__shfl((long long)(base + tot), 0, 64);
            const u32 *src = reinterpret_cast<const u32 *>(ctr);
            u32 *dst = reinterpret_cast<u32 *>(c.host_ctr);
            constexpr u32 kWords = (u32)(sizeof(Counters) / 4);
            for (u32 i = lane; i < kWords; i += 64u) {
                u32 w = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i == (u32)(offsetof(Counters, total_regions) / 4)) w = (u32)total;
                if (i == (u32)(offsetof(Counters, total_regions) / 4) + 1u) w = (u32)(total >> 32);
                if (i == (u32)(offsetof(Counters, region_overflow) / 4)) w = total > c.region_cap ? 1u : 0u;
                dst[i] = w;
            }
        }
    }
    __syncthreads();
    if (r0 >= c.n_reads) return;
    u64 dst = s_base + local;
    u64 offs[kScanPer];
    u64 types = 0;
    bool overflow = false;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        offs[k] = dst;
        const u64 r = r0 + k;
        if (r < c.n_reads) {
            u32 bad = 0;
            bool middle = false;
            const bool fits = dst + g[k] <= c.region_cap;
            if (closed[k]) { // (neither region lies in the middle: the first begins at 0, the second ends at len)
                u32 j = 0;
                if (ab[k].x != 0u && fits) c.bad_regions[dst + j++] = make_uint2(0u, ab[k].x);
                if (ab[k].y != L[k] && fits) c.bad_regions[dst + j] = make_uint2(ab[k].y, L[k]);
                bad = ab[k].x + (L[k] - ab[k].y);
            } else if (g[k]) {
                const uint2 *slot = a.stage + (a.off[r] + 2 * r);
                for (u32 j = 0; j < g[k]; j++) {
                    const uint2 v = slot[j];
                    if (fits) c.bad_regions[dst + j] = v;
                    bad += v.y - v.x;
                    middle |= (v.x != 0u) & (v.y != L[k]);
                }
            }
            overflow |= !fits;
            types |= (u64)classify(bad, middle, L[k], c.not_cov) << (8 * k);
            if (r == c.n_reads - 1) c.bad_offsets[c.n_reads] = dst + g[k];
        }
        dst += g[k];
    }
    if (overflow) atomicOr(&ctr->region_overflow, 1u);
    if (r0 + kScanPer <= c.n_reads) {
        ulonglong2 *bo = reinterpret_cast<ulonglong2 *>(c.bad_offsets + r0); // (engine-owned: 256-byte aligned, r0 % 4 == 0)
#pragma unroll
        for (int q = 0; q < kScanPer / 2; q++) bo[q] = make_ulonglong2(offs[2 * q], offs[2 * q + 1]);
        if constexpr (kScanPer == 4) *reinterpret_cast<u32 *>(c.read_type + r0) = (u32)types;
        else *reinterpret_cast<u64 *>(c.read_type + r0) = types;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPer; k++)
            if (r0 + k < c.n_reads) {
                c.bad_offsets[r0 + k] = offs[k];
                c.read_type[r0 + k] = (uint8_t)(types >> (8 * k));
            }
    }
}

} // namespace yk
