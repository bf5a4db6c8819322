// plan_compact.h — size-class binning before the sweeps; scan + compaction + classification after.
#pragma once
#include "device_common.h"

namespace yk {

// ---- plan: bin reads by event count (reads offsets only: 8 B/read) --------------------------
// Ranks: one LDS atomic per (wavefront, class present in it) — lanes of a class are counted
// with a ballot, their rank inside the wavefront is a popcount — then one global atomic per class
// per workgroup.  (One LDS atomic per read serialised 1024 lanes on one address: 8 us -> 3 us.)
constexpr int kPlanBlock = 1024; // threads; a workgroup takes PER slabs of 1024 consecutive reads (PER reads per thread)
constexpr u32 kPlanSmallReads = 400000; // batches below this: PER = 1
#ifndef YK_PLAN_SMALL_BLOCK
#define YK_PLAN_SMALL_BLOCK 1024
#endif
constexpr int kPlanSmallBlock = YK_PLAN_SMALL_BLOCK; // threads per workgroup of the short batches' form

// `zero` / `zero_words`: the control block of the NEXT run (the engine alternates between two), left
// zeroed here so that no run starts with a fill on its critical path.
// A workgroup takes PER slabs of 1024 consecutive reads: its two global atomics (counts and interval totals of
// the classes it met) hit the same two cache lines as every other workgroup's and are performed at the memory
// side one after the other — 1 953 workgroups for configs[2]'s 2 M reads made a 24 MB pass take 35 us; 489 take
// 19: PER = 4 for long batches.  A batch of 100 000 reads is 98 workgroups at PER = 1 and 25 at 4 — a quarter of
// the device's CUs busy for 9-11 us instead of 7: with the three-dispatch chain of round 3 PER = 1 gives 24.4 us
// per pipelined batch against 26.6 and 64.5 against 68 us for one batch at a time (round 2, five dispatches,
// measured it the other way round: 27.9 against 26.0).
template <int PER, int BLK = kPlanBlock>
__global__ __launch_bounds__(BLK) void plan_kernel(const u64 *off, u32 n_reads, u32 *lists,
                                                   Counters *ctr, u32 mode, u32 *zero,
                                                   u32 zero_words, u32 *counts, uint4 *mrec, u32 mrec_cap1, u32 mrec_cap2)
{
    constexpr int kPlanBlock = BLK; // (shadows the long batches' constant)
    for (u32 i = blockIdx.x * kPlanBlock + threadIdx.x; i < zero_words; i += gridDim.x * kPlanBlock)
        zero[i] = 0;
    __shared__ u32 s_cnt[CLS_COUNT];
    __shared__ u32 s_base[CLS_COUNT];
    __shared__ unsigned long long s_iv[CLS_COUNT];
    if (threadIdx.x < CLS_COUNT) {
        s_cnt[threadIdx.x] = 0;
        s_iv[threadIdx.x] = 0;
    }
    __syncthreads();
    constexpr int kPlanPer = PER, kPlanReads = kPlanBlock * PER;
    u32 cls[kPlanPer], local[kPlanPer];
    u64 first[kPlanPer]; // (off[r]: the workgroup classes' records)
    u32 niv[kPlanPer];
    const u64 lt = (1ull << lane_id()) - 1ull;
#pragma unroll
    for (int k = 0; k < kPlanPer; k++) {
        const u32 r = blockIdx.x * (u32)kPlanReads + (u32)k * kPlanBlock + threadIdx.x;
        cls[k] = CLS_COUNT; // CLS_COUNT = no read in this lane
        local[k] = 0;
        u64 n = 0;
        first[k] = 0, niv[k] = 0;
        if (r < n_reads) {
            first[k] = off[r];
            n = off[r + 1] - first[k];
            niv[k] = (u32)n;
            const u64 m = 2 * n;
            // mode: 0 = default, 1 = every read to the general path, 2 = one read per wavefront only,
            // 3 = rows (<= 128 intervals) but no 32-lane halves
            u32 c;
            if (mode == 1) c = CLS_GENERAL;
            else if (mode != 2 && m <= 32) c = CLS_R2;
            else if (mode != 2 && m <= 64) c = CLS_R4;
            else if (mode != 2 && m <= 128) c = CLS_R8;
            else if (mode != 2 && m <= 256) c = CLS_R16;
            else if (mode == 0 && m <= 512) c = CLS_H16;
            else if (m <= 128) c = CLS_W2;
            else if (m <= 256) c = CLS_W4;
            else if (m <= 512) c = CLS_W8;
            else if (m <= kSmallEvents) c = CLS_W16;
            else if (m <= kMedium1Events) c = CLS_MED1;
            else if (m <= kMedium2Events) c = CLS_MED2;
            else c = CLS_GENERAL;
            cls[k] = c;
        }
        // wave-aggregated ranks: peel off one class at a time
        u64 todo = __builtin_amdgcn_ballot_w64(cls[k] != CLS_COUNT);
        while (todo) {
            const u32 c = (u32)__builtin_amdgcn_readlane((int)cls[k], (int)__builtin_ctzll(todo));
            const u64 mask = __builtin_amdgcn_ballot_w64(cls[k] == c);
            // interval total of the class inside the wavefront (lanes outside contribute 0)
            u64 iv = (cls[k] == c) ? n : 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) iv += __shfl_xor(iv, d, 64);
            u32 base = 0;
            if (lane_id() == (u32)__builtin_ctzll(mask)) {
                base = atomicAdd(&s_cnt[c], (u32)__builtin_popcountll(mask));
                atomicAdd(&s_iv[c], (unsigned long long)iv);
            }
            base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(mask));
            if (cls[k] == c) local[k] = base + (u32)__builtin_popcountll(mask & lt);
            todo &= ~mask;
        }
    }
    __syncthreads();
    if (threadIdx.x < CLS_COUNT && s_cnt[threadIdx.x]) {
        s_base[threadIdx.x] = atomicAdd(&ctr->n[threadIdx.x], s_cnt[threadIdx.x]);
        atomicAdd((unsigned long long *)&ctr->iv[threadIdx.x], s_iv[threadIdx.x]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPlanPer; k++) {
        const u32 r = blockIdx.x * (u32)kPlanReads + (u32)k * kPlanBlock + threadIdx.x;
        if (cls[k] != CLS_COUNT) {
            const u64 pos = s_base[cls[k]] + local[k];
            lists[(u64)cls[k] * n_reads + pos] = r;
            // A read's region count starts out as "closed form: see closed[r]" (round 5): the screen then answers a read
            // it decides with ONE store, its (a, b) — not a count word besides — and everything else that finishes a read
            // writes its count over this (coalesced here: one lane per read and 4 bytes there)
            counts[r] = kClosedForm;
            // the workgroup classes' records (round 6): screen_wg_kernel gives a read to a one-read workgroup whose life is
            // mostly dependent round trips — list entry, then extent, then intervals —; with the extent beside the list entry it
            // is two.  M1's records first (a read of the class has more than 512 intervals: at most mrec_cap1), M2's behind.
            if (mrec != nullptr && (cls[k] == CLS_MED1 || cls[k] == CLS_MED2)) {
                const u64 at = cls[k] == CLS_MED1 ? pos : (u64)mrec_cap1 + pos;
                if (pos < (cls[k] == CLS_MED1 ? mrec_cap1 : mrec_cap2))
                    mrec[at] = make_uint4((u32)first[k], (u32)(first[k] >> 32), niv[k], r);
            }
        }
    }
}

#ifndef YK_SCAN_BLOCK
#define YK_SCAN_BLOCK 1024
#endif
constexpr int kScanBlock = YK_SCAN_BLOCK; // reads per workgroup of the follow-on kernel (one thread each)

// ---- follow-on kernel: compact each read's regions into the CSR and tag the read ------------
// Classification is reference src/editor/mod.rs:85-100 (type_of_read): u32 wrapping sum of
// (end - begin), IEEE f64 divide (no fast-math), strict >, NotCovered before Chimeric.
__device__ __forceinline__ u32 classify(u32 bad, bool middle_gap, u32 len, double not_cov)
{
    if ((double)bad / (double)len > not_cov) return 2u; // YACRD_NOT_COVERED (NaN -> false)
    return middle_gap ? 1u : 0u;                        // YACRD_CHIMERIC : YACRD_NOT_BAD
}

// (the scan + compaction + classification kernel lives in finish_compact.h: it first finishes the reads the
// screen deferred)

// Standalone classification over an existing region CSR (editors re-classify per record,
// e.g. reference src/editor/scrubbing.rs:181).
__global__ __launch_bounds__(256) void classify_csr_kernel(const u64 *bad_offsets,
                                                           const uint2 *bad_regions, const u32 *len,
                                                           u32 n_reads, double not_cov,
                                                           uint8_t *read_type)
{
    const u32 r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    const u64 a = bad_offsets[r], b = bad_offsets[r + 1];
    const u32 L = len[r];
    u32 bad = 0;
    bool middle = false;
    for (u64 k = a; k < b; k++) {
        const uint2 v = bad_regions[k];
        bad += v.y - v.x;
        middle |= (v.x != 0u) & (v.y != L);
    }
    read_type[r] = (uint8_t)classify(bad, middle, L, not_cov);
}

} // namespace yk
