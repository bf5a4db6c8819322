// one_batch.h — a short batch in ONE launch (round 4 prototype).
//
// What a batch of 10 000 - 400 000 reads pays for on the default path is not bytes but dispatches: plan_kernel
// (bins the reads by size), a host sync on its class counts (or a prediction of them), the fused screen launch over
// the class lists, finish_compact_kernel — three dependent kernels, each with its ramp-up, drain and launch gap, 66-70 us
// for 100 000 reads of which the memory system is busy for ~15.  Here a workgroup OWNS a slab of 128 consecutive reads
// from the first byte to the last:
//   S  every wavefront takes 16 consecutive reads: their offsets (one load), their classes (a ballot: <= 128 intervals
//      -> 16-lane groups, <= 256 -> 32-lane halves; nothing is binned across the batch, so there is nothing to plan),
//      then the healthy-read screen of sweep_wave.h (screen_reads: the fused launch's code) over them, two reads per
//      group and turn; verdicts (closed form / deferred) land in the workgroup's LDS, not in counts[] / closed[];
//   A  the reads the screen left are sorted right there, one per wavefront on 64 lanes (finish_item, finish_compact.h);
//   B  region counts -> exclusive scan (decoupled look-back over the slabs' aggregates, every wavefront of the
//      workgroup looking at 64 predecessors at once), regions into the CSR, type_of_read; the slab that ends the batch
//      sends the counter block home.
// No class lists, no class counts, no prediction; what the host waits for is one kernel.
// A read of more than 256 intervals (the workgroup / device-wide classes) is not handled here: the kernel raises
// Counters::ob_unsupported and the engine runs the batch through the default path (engine.hip: run_one_launch).
#pragma once
#include "finish_compact.h"

namespace yk {

constexpr int kObWaves = 8, kObThreads = 64 * kObWaves;
constexpr int kObReadsPerWave = 16, kObSlab = kObWaves * kObReadsPerWave; // reads per workgroup
#ifndef YK_OB_ITEMS
#define YK_OB_ITEMS 2 // reads per lane group and turn of the screen (loads of both in flight together)
#endif
#ifndef YK_OB_OCC
#define YK_OB_OCC 6 // wavefronts per SIMD the register budget allows: three workgroups per CU (LDS: 3 x 44 KB)
#endif
constexpr int kObItems = YK_OB_ITEMS;

struct OneBatchArgs {
    CompactArgs2 c;   // sweep (off / iv / len / cov / prefilter / stage / counts / rej_list / rej_count / ctr), scan_state, outputs
    u32 *zero;        // the engine's other control block, zeroed here for the next run (as plan_kernel does)
    u32 zero_words;
};

struct VerdictsToLds { // (see VerdictsToGlobal)
    u32 *g;       // [kObSlab] region count | kClosedForm | kDeferredMark, by index inside the slab
    uint2 *ab;    // [kObSlab]
    u32 r_base;   // first read of the slab
    u32 count;    // a.prefilter == 2
    Counters *ctr;
    __device__ __forceinline__ void closed(u32 r, u32 ra, u32 rb, u32 len) const
    {
        const u32 i = r - r_base;
        if (ra != 0 || rb != len) {
            ab[i] = make_uint2(ra, rb);
            g[i] = kClosedForm;
        } else {
            g[i] = 0;
        }
        if (count) atomicAdd(&ctr->prefiltered, 1u);
    }
    __device__ __forceinline__ void deferred(u32 r) const { g[r - r_base] = kDeferredMark; }
};

__global__ __launch_bounds__(kObThreads, YK_OB_OCC) void one_batch_kernel(OneBatchArgs ob)
{
    static_assert(!YK_HOLE_FORM, "hole_form_call uses a one-wavefront table");
    const CompactArgs2 &c = ob.c;
    const SweepArgs &a = c.sweep;
    Counters *ctr = a.ctr;
    for (u32 i = blockIdx.x * kObThreads + threadIdx.x; i < ob.zero_words; i += gridDim.x * kObThreads) ob.zero[i] = 0;

    __shared__ u32 s_g[kObSlab];
    __shared__ uint2 s_ab[kObSlab];
    __shared__ uint8_t s_l16[kObWaves][kObReadsPerWave], s_l32[kObWaves][kObReadsPerWave];
    __shared__ u32 sc[kObWaves];
    __shared__ u32 s_bid, s_n_def, s_unsup;
    __shared__ unsigned long long s_iv_def;
    __shared__ u64 s_part[kObWaves]; // look-back: per wavefront, the sum of its window up to its nearest prefix
    __shared__ u32 s_flag[kObWaves]; //            1 = holds a prefix, 2 = an empty entry in front of it
    if (threadIdx.x == 0) {
        s_bid = atomicAdd(&ctr->scan_ticket, 1u);
        s_n_def = 0, s_unsup = 0, s_iv_def = 0;
    }
    __syncthreads();
    const u32 bid = s_bid, lane = lane_id(), wave = threadIdx.x >> 6;
    const u32 slab0 = bid * (u32)kObSlab;

    // ---- S: this wavefront's 16 reads through the screen
    {
        const u32 r0 = slab0 + wave * (u32)kObReadsPerWave;
        const bool in = lane < (u32)kObReadsPerWave && r0 + lane < c.n_reads;
        u32 n = 0;
        bool huge = false;
        if (in) {
            const ulonglong2 oo = *reinterpret_cast<const ulonglong2 *>(a.off + (r0 + lane)); // off[r], off[r + 1] (8-byte aligned 16-byte load)
            const u64 nn = oo.y - oo.x;
            huge = nn > 256;
            n = huge ? 0u : (u32)nn;
        }
        const bool k16 = in && !huge && n <= 128u, k32 = in && !huge && n > 128u;
        const u64 m16 = __builtin_amdgcn_ballot_w64(k16), m32 = __builtin_amdgcn_ballot_w64(k32);
        const u64 lt = (1ull << lane) - 1ull;
        if (k16) s_l16[wave][__builtin_popcountll(m16 & lt)] = (uint8_t)lane;
        if (k32) s_l32[wave][__builtin_popcountll(m32 & lt)] = (uint8_t)lane;
        if (lane < (u32)kObReadsPerWave && !k16 && !k32) s_g[wave * kObReadsPerWave + lane] = 0u; // beyond the batch / not handled here
        if (__builtin_amdgcn_ballot_w64(huge) != 0 && lane == 0) s_unsup = 1u;
        wave_lds_sync();
        const u32 cnt16 = (u32)__builtin_popcountll(m16), cnt32 = (u32)__builtin_popcountll(m32);
        const VerdictsToLds sink{s_g, s_ab, slab0, a.prefilter == 2 ? 1u : 0u, ctr};
        for (u32 i0 = 0; i0 < cnt16; i0 += 4u * (u32)kObItems) { // (uniform in the wavefront)
            u32 r[kObItems];
            bool act[kObItems];
#pragma unroll
            for (int t = 0; t < kObItems; t++) {
                const u32 idx = i0 + (u32)t * 4u + (lane >> 4);
                act[t] = idx < cnt16;
                r[t] = act[t] ? r0 + s_l16[wave][idx] : 0u;
            }
            screen_reads<16, kObItems, false, kObWaves>(a, r, act, sink);
            wave_lds_sync(); // (the next turn zeroes the table)
        }
        for (u32 i0 = 0; i0 < cnt32; i0 += 2u * (u32)kObItems) {
            u32 r[kObItems];
            bool act[kObItems];
#pragma unroll
            for (int t = 0; t < kObItems; t++) {
                const u32 idx = i0 + (u32)t * 2u + (lane >> 5);
                act[t] = idx < cnt32;
                r[t] = act[t] ? r0 + s_l32[wave][idx] : 0u;
            }
            screen_reads<32, kObItems, false, kObWaves>(a, r, act, sink);
            wave_lds_sync();
        }

        // ---- A: what the screen left, one read per turn on all 64 lanes
        const bool marked = lane < (u32)kObReadsPerWave && s_g[wave * kObReadsPerWave + lane] == kDeferredMark;
        u64 todo = __builtin_amdgcn_ballot_w64(marked);
        if (todo) {
            u32 n_def = 0;
            u64 iv_def = 0;
            while (todo) { // (uniform)
                const u32 i = (u32)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const u32 rr = r0 + i;
                const u64 o = a.off[rr];
                const u32 nr = (u32)(a.off[rr + 1] - o);
                const u32 length = a.len[rr];
                if (nr > 128u)
                    finish_item<8>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, nr, length);
                else
                    finish_item<4>(a.off, a.iv, a.len, a.stage, a.counts, a.rej_list, a.rej_count, a.ctr, a.cov, rr, o, nr, length);
                n_def++;
                iv_def += nr;
            }
            if (lane == 0) {
                atomicAdd(&s_n_def, n_def);
                atomicAdd(&s_iv_def, (unsigned long long)iv_def);
            }
        }
    }
    __syncthreads(); // (global stores of this workgroup's wavefronts — counts[], the stage slots — are visible to each other after it)
    if (threadIdx.x == 0 && (s_n_def || s_unsup)) {
        // returning atomics, waited for: performed before this slab publishes its aggregate (see finish_compact_kernel)
        const u32 t0 = s_n_def ? atomicAdd(&ctr->deferred, s_n_def) : 0u;
        const unsigned long long t1 = s_n_def ? atomicAdd((unsigned long long *)&ctr->deferred_iv, s_iv_def) : 0ull;
        const u32 t2 = s_unsup ? atomicOr(&ctr->ob_unsupported, 1u) : 0u;
        asm volatile("" ::"v"(t0), "v"(t1), "v"(t2));
    }

    // ---- B: scan, compaction, classification (one thread per read)
    const u32 r = slab0 + threadIdx.x;
    const bool in = threadIdx.x < (u32)kObSlab && r < c.n_reads;
    u32 g = in ? s_g[threadIdx.x] : 0u;
    if (g == kDeferredMark) g = a.counts[r];
    const u64 off_r = in ? a.off[r] : 0;
    const u32 L = in ? a.len[r] : 0u;
    const bool closed = g == kClosedForm;
    uint2 ab = make_uint2(0u, L);
    if (closed) {
        ab = s_ab[threadIdx.x];
        g = (ab.x != 0u ? 1u : 0u) + (ab.y != L ? 1u : 0u);
    }
    u32 tot;
    const u32 local = block_excl_add<kObThreads>(g, sc, tot);

    // decoupled look-back, kObThreads predecessors per round trip: wavefront w looks at bid - 1 - 64 w - lane
    constexpr u64 kAgg = 1ull << 62, kPre = 2ull << 62, kVal = (1ull << 62) - 1;
    if (bid > 0 && threadIdx.x == 0)
        __hip_atomic_store(&c.scan_state[bid], kAgg | tot, YK_FINISH_ORDER_REL, __HIP_MEMORY_SCOPE_AGENT);
    u64 base = 0;
    u32 polls = 0;
    for (i32 hi = (i32)bid - 1; bid > 0;) { // (uniform in the workgroup)
        const i32 idx = hi - (i32)threadIdx.x;
        const u64 v = idx >= 0 ? __hip_atomic_load(&c.scan_state[idx], YK_FINISH_ORDER_ACQ, __HIP_MEMORY_SCOPE_AGENT)
                               : kPre; // before the first slab: prefix 0
        const u64 pre = __builtin_amdgcn_ballot_w64((v >> 62) == 2);
        const u64 before = pre ? ((pre & (0 - pre)) - 1ull) : ~0ull; // lanes nearer than this window's nearest prefix
        const bool hole = (__builtin_amdgcn_ballot_w64((v >> 62) == 0) & before) != 0;
        const u32 first_pre = pre ? (u32)__builtin_ctzll(pre) : 64u;
        u64 part = lane <= first_pre ? (v & kVal) : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) {
            s_part[wave] = part;
            s_flag[wave] = (pre ? 1u : 0u) | (hole ? 2u : 0u);
        }
        __syncthreads();
        // nearest window first: sums up to the first window that holds a prefix; an empty entry on the way = look again
        u64 sum = 0;
        bool found = false, again = false;
#pragma unroll
        for (int w = 0; w < kObWaves; w++) {
            if (!found && !again) {
                const u32 f = s_flag[w];
                if (f & 2u) again = true;
                else {
                    sum += s_part[w];
                    found = (f & 1u) != 0;
                }
            }
        }
        __syncthreads(); // (s_part / s_flag are written again)
        if (again) {
            // (tickets are handed out in slab order, so every predecessor is running or done and this wait ends; the bound
            // is there so that a broken invariant shows up as a batch sent down the default path, not as a hung device)
            if (++polls > (1u << 20)) {
                if (threadIdx.x == 0) {
                    atomicOr(&ctr->ob_unsupported, 2u);
                    if (c.host_ctr) __hip_atomic_store(&c.host_ctr->ob_unsupported, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        base += sum;
        if (found) break;
        hi -= (i32)kObThreads;
    }
    const bool last_slab = (u64)(bid + 1) * kObSlab >= c.n_reads;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&c.scan_state[bid], kPre | (base + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last_slab) ctr->total_regions = base + tot;
    }
    // the counters go home from the slab that ends the batch (finish_compact_kernel: it has seen everyone's aggregate)
    if (c.host_ctr && last_slab && threadIdx.x < 64) {
        const u64 total = base + tot;
        const u32 *src = reinterpret_cast<const u32 *>(ctr);
        u32 *dst = reinterpret_cast<u32 *>(c.host_ctr);
        constexpr u32 kWords = (u32)(sizeof(Counters) / 4);
        for (u32 i = lane; i < kWords; i += 64u) {
            u32 w = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i == (u32)(offsetof(Counters, total_regions) / 4)) w = (u32)total;
            if (i == (u32)(offsetof(Counters, total_regions) / 4) + 1u) w = (u32)(total >> 32);
            if (i == (u32)(offsetof(Counters, region_overflow) / 4)) w = total > c.region_cap ? 1u : 0u;
            if (i == (u32)(offsetof(Counters, ob_unsupported) / 4)) continue; // (see below)
            dst[i] = w;
        }
        // (its own word: a slab whose look-back gave up writes 2 there directly, whenever that happens)
        if (lane == 0) {
            const u32 w = __hip_atomic_load(&ctr->ob_unsupported, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w) __hip_atomic_store(&c.host_ctr->ob_unsupported, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (in) {
        const u64 dst = base + local;
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
