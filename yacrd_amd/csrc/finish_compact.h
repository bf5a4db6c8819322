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
#include "sweep_wave.h"

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
                __hip_atomic_store(&c.scan_state[bid], kAgg | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (i32 hi = (i32)bid - 1;; hi -= 64) {
                const i32 idx = hi - (i32)lane; // lane 0 looks at the nearest predecessor
                u64 v, pre;
                for (;;) { // until the window holds no empty entry before its nearest prefix
                    v = idx >= 0 ? __hip_atomic_load(&c.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
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

} // namespace yk
