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
__device__ __forceinline__ void finish_marked(const SweepArgs &a, u32 bid, u32 n_marked, const MarkedRead *list)
{
    for (u32 i = threadIdx.x >> 6; i < n_marked; i += (u32)kFinishWaves) { // (uniform in the wavefront)
        const MarkedRead m = list[i];
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
    const SweepArgs &a = c.sweep;
    Counters *ctr = a.ctr;
    if (threadIdx.x == 0) {
        s_bid = atomicAdd(&ctr->scan_ticket, 1u);
        s_n = 0;
        s_iv = 0;
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
        finish_marked(a, bid, s_n, s_list);
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
// ---- a marked read through the pile-trimming filter (§3.5) and a SHORT register sort (round 4) -------------------
// The 64-lane sort of a marked read is 847 instructions, most of them spent ordering events that cannot matter: of a
// chimera's ~400 events the filter keeps the c + 1 outermost of the piles at 0 / len and what lies around the junction —
// 60-100 keys: one or two per lane instead of eight.  LdsTrim<64, kTrimWords>: 32 one-position bins at either end of the
// read, up to 64 coarse bins in between, two bins per lane; the plan (lds_trim_plan, sweep_lds.h) is the workgroup classes'
// with one wavefront as the "workgroup"; the survivors go from LDS into registers — K' = 1, 2, 4 or 8 keys per lane by their
// number — and through sweep_group_keys, the register classes' sort + sweep.  Plain reads only (start < end <= len):
// anything else takes the full sort (finish_item).  false = not taken.
constexpr int kTrimWords = 1280; // per wavefront: 640 survivors at most | 128 zero-length counters | 128 bins x 4 counters
template <int KP>
__device__ __forceinline__ void trimmed_keys_sweep(const u32 *keys, u32 m_sort, u32 len, i32 c, u32 rr, const SweepArgs &a, const LaneConst &lc)
{
    u32 x[KP];
    const u32 lane = lane_id();
#pragma unroll
    for (int q = 0; q < KP; q++) {
        const u32 i = lane * (u32)KP + (u32)q;
        x[q] = i < m_sort ? keys[i] : kPadKey;
    }
    sweep_group_keys<64, KP, 0>(x, m_sort, len, c, true, rr, 0ull, 0ull, false, a, lc);
}
__device__ __forceinline__ bool trim_item(const SweepArgs &a, u32 rr, u64 o, u32 n, u32 len, u32 *keys, const LaneConst &lc)
{
    using F = LdsTrim<64, kTrimWords>;
    if (len > kMaxKeyPos || n < 2u || !a.prefilter) return false; // (uniform)
    const u32 lane = lane_id();
    const typename F::Geo geo = F::geo(len);
    u32 *tab = F::tab(keys);
    reinterpret_cast<uint4 *>(tab)[2 * lane] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4 *>(tab)[2 * lane + 1] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint2 *>(F::ztab(keys))[lane] = make_uint2(0u, 0u);
    const uint2 *iv = a.iv + o;
    uint2 v[4];
    bool real[4];
    u32 irregular = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = iv[min(lane + 64u * (u32)j, n - 1u)];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        real[j] = lane + 64u * (u32)j < n;
        irregular |= (real[j] && (v[j].x >= v[j].y || v[j].y > len)) ? 1u : 0u;
    }
    if (__builtin_amdgcn_ballot_w64(irregular != 0) != 0) return false; // (uniform: the full sort knows every kind of interval)
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (real[j]) {
            const u32 ks = (v[j].x << kKeyShift) | 3u, ke = v[j].y << kKeyShift;
            atomicAdd(tab + F::idx(geo, ks) * 4u + (lane & 3u), 1u);
            atomicAdd(tab + F::idx(geo, ke) * 4u + (lane & 3u), 0x10000u);
        }
    }
    wave_lds_sync();
    u32 syn_start = 0;
    const u32 m_sort = lds_trim_plan<64, kTrimWords>(keys, len, a.cov, nullptr, syn_start);
    if (m_sort == 0 || m_sort > 512u) return false; // (uniform; counts[rr] is still the mark: the caller sorts the read whole)
    auto take = [&](u32 *cur) -> u32 {
        return (i32)*reinterpret_cast<volatile u32 *>(cur) >= 0x10000 ? atomicAdd(cur, F::kTakeOne) : 0u;
    };
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (real[j]) {
            const u32 ks = (v[j].x << kKeyShift) | 3u, ke = v[j].y << kKeyShift;
            const u32 is = F::idx(geo, ks), ie = F::idx(geo, ke);
            const u32 ps = take(tab + is * 4u + (F::uniform(geo, is) ? 1u : (lane & 3u)));
            const u32 pe = take(tab + ie * 4u + (F::uniform(geo, ie) ? 0u : (lane & 3u)));
            if ((i32)ps >= 0x10000) keys[ps & 0xFFFFu] = ks; // quota left: kept, at the slot in the low half
            if ((i32)pe >= 0x10000) keys[pe & 0xFFFFu] = ke;
        }
    }
    wave_lds_sync();
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    if (m_sort <= 64u) trimmed_keys_sweep<1>(keys, m_sort, len, c, rr, a, lc);
    else if (m_sort <= 128u) trimmed_keys_sweep<2>(keys, m_sort, len, c, rr, a, lc);
    else if (m_sort <= 256u) trimmed_keys_sweep<4>(keys, m_sort, len, c, rr, a, lc);
    else trimmed_keys_sweep<8>(keys, m_sort, len, c, rr, a, lc);
    return true;
}

// reads per slab / threads per workgroup (follow-on step on configs[2] / configs[4], profiles/r04/i_ab_deferred_slab_size.log):
// 256 / 256: 0.231 / 0.541 ms; 512 / 256: 0.158 / 0.347; 1024 / 256: 0.152 / 0.297; 2048 / 512: 0.134 / 0.283 — longer lists
// fill the wavefronts' turns evenly and fewer workgroups post the two counter atomics
#ifndef YK_DEFER_SLAB
#define YK_DEFER_SLAB 2048
#endif
#ifndef YK_DEFER_THREADS
#define YK_DEFER_THREADS 512
#endif
constexpr int kDeferSlab = YK_DEFER_SLAB, kDeferThreads = YK_DEFER_THREADS; // (A/B: profiles/r04)
static_assert(kDeferSlab % kDeferThreads == 0 && kDeferSlab <= 65536, "a thread looks at whole reads; list entries hold a 16-bit index");

#ifndef YK_DEFER_TRIM
// The marked reads through the pile-trimming filter and a short sort first (trim_item): bit-exact (GPU tests and fuzz with
// the two-kernel follow-on forced), and no faster — the filter's two counting passes, its plan and the scatter cost what the
// shorter sort saves: 43.6 M VALU instructions against 40.3 M for configs[2]'s 47 608 reads, the follow-on step 0.173 against
// 0.155 ms (profiles/r04/g_ab_trimmed_deferred_sweep.log; round 2 found the same inside the register-sort kernel).  Off.
#define YK_DEFER_TRIM 0
#endif
#ifndef YK_DEFER_FILTER
// The marked reads through the FILTERED exact sweep first (sweep_filtered.h, round 5): two (129..256 intervals) or four
// (<= 128) reads per wavefront and turn; what it does not take — not plain, an interval shorter than the screen's window,
// a window short of c + 1, more than 128 kept events: ~15 % of the generator's deferred reads — is listed again and
// sorted whole, one read per wavefront, as before.  0 builds round 4's kernel.
#define YK_DEFER_FILTER 1
#endif
#ifndef YK_DEFER_SWEEP_OCC
// wavefronts per SIMD the register budget allows.  Without the trimming path: 8 / 6 / 5 gave 0.164 / 0.158 / 0.159 ms of
// follow-on time on configs[2] (profiles/r04/c_ab_follow_on.log); with it the kernel wants 128 registers.  The filtered
// sweep (32-lane groups only, 3 KB of tables per wavefront, 8-byte list entries: 44 KB per workgroup) keeps three workgroups per CU.
#define YK_DEFER_SWEEP_OCC (YK_DEFER_TRIM ? 4 : 6)
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
    for (int j = 0; j < K / 4; j++) v[j] = *reinterpret_cast<const uint4 *>(src + min(2u * (lig + (u32)LANES * j), last2));
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
#if YK_DEFER_FILTER
// (round 5) The marked reads of a slab — listed in LDS with their intervals and length, eight bytes each — go through the
// filtered exact sweep two per wavefront and turn, on 32-lane groups whatever their size (a read of <= 128 intervals on 16
// lanes would share its turn with three others, but its table — 5 KB per wavefront instead of 3 — costs the kernel a
// workgroup per CU: deferred_sweep_kernel serves long batches, and with lists of 16-byte entries and 5 KB tables it was
// no faster than round 4's on configs[4], 0.291 against 0.282 ms of follow-on step, for all its 33 % fewer VALU
// instructions: profiles/r05/c_ab_deferred_filtered_2048_1024_none.log); what the filter does not take is listed again and
// sorted whole, one read per wavefront.
__global__ __launch_bounds__(kDeferThreads, YK_DEFER_SWEEP_OCC) void deferred_sweep_kernel(SweepArgs a, u32 n_reads)
{
    constexpr u32 kWaves = kDeferThreads / 64;
    constexpr int kTabWords = 2 * (2 * kScreenWindow + 32) * 4; // two 32-lane groups
    __shared__ uint2 s_list[kDeferSlab]; // index inside the slab | intervals << 16, length
    __shared__ unsigned short s_fb[kDeferSlab];
    __shared__ u32 s_n, s_nfb;
    __shared__ unsigned long long s_iv;
    if (threadIdx.x == 0) s_n = 0, s_nfb = 0, s_iv = 0;
    __syncthreads();
    const u32 lane = lane_id();
    const u32 slab0 = blockIdx.x * (u32)kDeferSlab;
    constexpr int PER = kDeferSlab / kDeferThreads;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)k * kDeferThreads + threadIdx.x, r = slab0 + i;
        const bool marked = r < n_reads && a.counts[r] == kDeferredMark;
        const u64 mm = __builtin_amdgcn_ballot_w64(marked);
        if (mm == 0) continue; // (uniform in the wavefront)
        u32 n = 0, len0 = 0;
        if (marked) {
            n = (u32)(a.off[r + 1] - a.off[r]);
            len0 = a.len[r];
        }
        u32 base = 0;
        if (lane == (u32)__builtin_ctzll(mm)) base = atomicAdd(&s_n, (u32)__builtin_popcountll(mm));
        base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(mm));
        if (marked) s_list[base + (u32)__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = make_uint2(i | (n << 16), len0);
        u64 iv = n; // intervals of the marked reads, for the roofline's exact byte count
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) iv += __shfl_xor(iv, d, 64);
        if (lane == 0) atomicAdd(&s_iv, (unsigned long long)iv);
    }
    __syncthreads();
    const u32 n_marked = s_n;
    if (n_marked == 0) return; // uniform in the workgroup
    if (threadIdx.x == 0) {
        atomicAdd(&a.ctr->deferred, n_marked);
        atomicAdd((unsigned long long *)&a.ctr->deferred_iv, s_iv);
    }
    const LaneConst lcf = make_lane_const(lane);
    const u32 wv = threadIdx.x >> 6;
    for (u32 p0 = wv * 2u; p0 < n_marked; p0 += kWaves * 2u) { // (uniform in the wavefront)
        const u32 p = p0 + (lane >> 5);
        const bool have = p < n_marked;
        const uint2 e = s_list[have ? p : p0];
        const u32 rr = slab0 + (e.x & 0xFFFFu);
        const u64 o = a.off[rr]; // (in the L2: the listing has just read it)
        const bool done = filtered_turn<32, (int)kWaves, kTabWords>(a, have, rr, o, e.x >> 16, e.y, lcf);
        if (have && !done && (lane & 31u) == 31u) s_fb[atomicAdd(&s_nfb, 1u)] = (unsigned short)p;
    }
    __syncthreads();
    const u32 nfb = s_nfb;
    for (u32 i = wv; i < nfb; i += kWaves) { // (uniform in the wavefront): sorted whole, one read per wavefront
        const uint2 e = s_list[s_fb[i]];
        const u32 rr = slab0 + (e.x & 0xFFFFu), n = e.x >> 16, len = e.y;
        const u64 o = a.off[rr];
        if (n > 128u) finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
        else finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
    }
}
#else
__global__ __launch_bounds__(kDeferThreads, YK_DEFER_SWEEP_OCC) void deferred_sweep_kernel(SweepArgs a, u32 n_reads)
{
    // what the thread that found the mark already knows about the read — offset, index inside the slab | intervals << 16,
    // length — saves every turn a round trip: the long reads from the front, the others from the back
    __shared__ u64 s_off[kDeferSlab];
    __shared__ uint2 s_list[kDeferSlab];
    __shared__ u32 s_n8, s_n4;
    __shared__ unsigned long long s_iv;
    if (threadIdx.x == 0) s_n8 = 0, s_n4 = 0, s_iv = 0;
    __syncthreads();
    const u32 lane = lane_id();
    const u32 slab0 = blockIdx.x * (u32)kDeferSlab;
    constexpr int PER = kDeferSlab / kDeferThreads;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)k * kDeferThreads + threadIdx.x, r = slab0 + i;
        const bool marked = r < n_reads && a.counts[r] == kDeferredMark;
        const u64 mm = __builtin_amdgcn_ballot_w64(marked);
        if (mm == 0) continue; // (uniform in the wavefront)
        u32 n = 0, len0 = 0;
        u64 o0 = 0;
        if (marked) {
            o0 = a.off[r];
            n = (u32)(a.off[r + 1] - o0);
            len0 = a.len[r];
        }
        const bool big = marked && n > 128u; // (a marked read has at most 256 intervals)
        const u64 mb = __builtin_amdgcn_ballot_w64(big), ms = mm & ~mb;
        u32 bb = 0, bs = 0;
        if (lane == (u32)__builtin_ctzll(mm)) {
            if (mb) bb = atomicAdd(&s_n8, (u32)__builtin_popcountll(mb));
            if (ms) bs = atomicAdd(&s_n4, (u32)__builtin_popcountll(ms));
        }
        bb = (u32)__builtin_amdgcn_readlane((int)bb, (int)__builtin_ctzll(mm));
        bs = (u32)__builtin_amdgcn_readlane((int)bs, (int)__builtin_ctzll(mm));
        const u64 below = (1ull << lane) - 1ull;
        if (marked) {
            const u32 at = big ? bb + (u32)__builtin_popcountll(mb & below)
                               : (u32)kDeferSlab - 1u - (bs + (u32)__builtin_popcountll(ms & below));
            s_list[at] = make_uint2(i | (n << 16), len0);
            s_off[at] = o0;
        }
        u64 iv = n; // intervals of the marked reads, for the roofline's exact byte count
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) iv += __shfl_xor(iv, d, 64);
        if (lane == 0) atomicAdd(&s_iv, (unsigned long long)iv);
    }
    __syncthreads();
    const u32 n8 = s_n8, n4 = s_n4, n_marked = n8 + n4;
    if (n_marked == 0) return; // uniform in the workgroup
    if (threadIdx.x == 0) {
        atomicAdd(&a.ctr->deferred, n_marked);
        atomicAdd((unsigned long long *)&a.ctr->deferred_iv, s_iv);
    }
    constexpr u32 kWaves = kDeferThreads / 64;
#if YK_DEFER_TRIM
    __shared__ __attribute__((aligned(16))) u32 s_trim[kWaves][kTrimWords];
    const LaneConst lc = make_lane_const(lane);
#endif
#if 0
    // ---- first the filtered sweep, several reads per wavefront and turn; what it leaves is listed in s_fb
    __shared__ unsigned short s_fb[kDeferSlab];
    __shared__ u32 s_nfb;
    {
        if (threadIdx.x == 0) s_nfb = 0;
        __syncthreads();
        const LaneConst lcf = make_lane_const(lane);
        const u32 wv = threadIdx.x >> 6;
        auto turn = [&](auto lanes_tag, u32 p, bool have, u32 at) {
            constexpr int LANES = decltype(lanes_tag)::value;
            const uint2 e = s_list[have ? at : 0u];
            const u64 o = s_off[have ? at : 0u];
            const bool done = filtered_turn<LANES, (int)kWaves>(a, have, slab0 + (e.x & 0xFFFFu), o, e.x >> 16, e.y, lcf);
            if (have && !done && (lane & (u32)(LANES - 1)) == (u32)(LANES - 1)) s_fb[atomicAdd(&s_nfb, 1u)] = (unsigned short)at;
            (void)p;
        };
        for (u32 p0 = wv * 2u; p0 < n8; p0 += kWaves * 2u) { // (uniform in the wavefront)
            const u32 p = p0 + (lane >> 5);
            turn(std::integral_constant<int, 32>{}, p, p < n8, p);
        }
        for (u32 p0 = wv * 4u; p0 < n4; p0 += kWaves * 4u) {
            const u32 p = p0 + (lane >> 4);
            turn(std::integral_constant<int, 16>{}, p, p < n4, (u32)kDeferSlab - 1u - p);
        }
        __syncthreads();
        const u32 nfb = s_nfb;
        for (u32 i = wv; i < nfb; i += kWaves) { // (uniform in the wavefront): sorted whole, one read per wavefront
            const u32 at = s_fb[i];
            const uint2 e = s_list[at];
            const u32 rr = slab0 + (e.x & 0xFFFFu), n = e.x >> 16, len = e.y;
            const u64 o = s_off[at];
            if (n > 128u) finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
            else finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
        }
        return;
    }
#endif
    // one read per wavefront and turn, the long ones (8 keys per lane) first: list position p < n8 is s_list[p],
    // p >= n8 is the (p - n8)-th entry from the back
    for (u32 p = threadIdx.x >> 6; p < n_marked; p += kWaves) { // (uniform in the wavefront)
        const u32 at = p < n8 ? p : (u32)kDeferSlab - 1u - (p - n8);
        const uint2 e = s_list[at];
        const u32 rr = slab0 + (e.x & 0xFFFFu), n = e.x >> 16, len = e.y;
        const u64 o = s_off[at];
#if YK_DEFER_TRIM
        if (trim_item(a, rr, o, n, len, s_trim[threadIdx.x >> 6], lc)) continue;
#endif
        if (n > 128u) finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
        else finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, n, len);
    }
}

#endif // YK_DEFER_FILTER

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
