// plan_compact.h — size-class binning before the sweeps; scan + compaction + classification after.
#pragma once
#include "device_common.h"

namespace yk {

// ---- plan: bin reads by event count (reads offsets only: 8 B/read) --------------------------
// Ranks are taken with LDS atomics inside the workgroup and one global atomic per class per
// workgroup, so the global counters see ~4 atomics per 1024 reads.
constexpr int kPlanBlock = 1024;

__global__ __launch_bounds__(kPlanBlock) void plan_kernel(const u64 *off, u32 n_reads, u32 *lists,
                                                          Counters *ctr, u32 force_general)
{
    __shared__ u32 s_cnt[CLS_COUNT];
    __shared__ u32 s_base[CLS_COUNT];
    __shared__ unsigned long long s_iv[CLS_COUNT];
    if (threadIdx.x < CLS_COUNT) {
        s_cnt[threadIdx.x] = 0;
        s_iv[threadIdx.x] = 0;
    }
    __syncthreads();
    const u32 r = blockIdx.x * kPlanBlock + threadIdx.x;
    u32 cls = 0, local = 0;
    if (r < n_reads) {
        const u64 n = off[r + 1] - off[r];
        const u64 m = 2 * n;
        if (force_general) cls = CLS_GENERAL;
        else if (m <= 128) cls = CLS_W2;
        else if (m <= 256) cls = CLS_W4;
        else if (m <= 512) cls = CLS_W8;
        else if (m <= kSmallEvents) cls = CLS_W16;
        else if (m <= kMedium1Events) cls = CLS_MED1;
        else if (m <= kMedium2Events) cls = CLS_MED2;
        else cls = CLS_GENERAL;
        local = atomicAdd(&s_cnt[cls], 1u);
        atomicAdd(&s_iv[cls], (unsigned long long)n);
    }
    __syncthreads();
    if (threadIdx.x < CLS_COUNT && s_cnt[threadIdx.x]) {
        s_base[threadIdx.x] = atomicAdd(&ctr->n[threadIdx.x], s_cnt[threadIdx.x]);
        atomicAdd((unsigned long long *)&ctr->iv[threadIdx.x], s_iv[threadIdx.x]);
    }
    __syncthreads();
    if (r < n_reads) lists[(u64)cls * n_reads + s_base[cls] + local] = r;
}

// ---- bad_offsets = exclusive scan of per-read region counts (three small kernels) -----------
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(kScanBlock) void count_block_sums_kernel(const u32 *counts,
                                                                      u32 n_reads, u64 *block_sums)
{
    __shared__ u32 sc[kScanBlock / 64];
    const u32 r = blockIdx.x * kScanBlock + threadIdx.x;
    u32 v = (r < n_reads) ? counts[r] : 0u;
    u32 tot;
    block_excl_add<kScanBlock>(v, sc, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single workgroup: in-place exclusive scan of block_sums[nb]; total -> block_sums[nb]
__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(u64 *block_sums, u32 nb)
{
    __shared__ u64 sh[kScanBlock];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < nb; base += kScanBlock) {
        const u32 i = base + threadIdx.x;
        const u64 v = (i < nb) ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (u32 d = 1; d < kScanBlock; d <<= 1) { // Hillis-Steele, inclusive
            u64 t = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const u64 carry = carry_s;
        if (i < nb) block_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry_s = carry + sh[kScanBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry_s;
}

// ---- follow-on kernel: compact each read's regions into the CSR and tag the read ------------
// Classification is reference src/editor/mod.rs:85-100 (type_of_read): u32 wrapping sum of
// (end - begin), IEEE f64 divide (no fast-math), strict >, NotCovered before Chimeric.
__device__ __forceinline__ u32 classify(u32 bad, bool middle_gap, u32 len, double not_cov)
{
    if ((double)bad / (double)len > not_cov) return 2u; // YACRD_NOT_COVERED (NaN -> false)
    return middle_gap ? 1u : 0u;                        // YACRD_CHIMERIC : YACRD_NOT_BAD
}

__global__ __launch_bounds__(kScanBlock) void compact_classify_kernel(
    const u64 *off, const u32 *len, const uint2 *stage, const u32 *counts, const u64 *block_base,
    u32 n_reads, double not_cov, u64 *bad_offsets, uint2 *bad_regions, u64 region_cap,
    uint8_t *read_type, Counters *ctr)
{
    __shared__ u32 sc[kScanBlock / 64];
    const u32 r = blockIdx.x * kScanBlock + threadIdx.x;
    const u32 g = (r < n_reads) ? counts[r] : 0u;
    u32 tot;
    const u32 local = block_excl_add<kScanBlock>(g, sc, tot);
    if (r >= n_reads) return;
    const u64 dst = block_base[blockIdx.x] + local;
    bad_offsets[r] = dst;
    if (r == n_reads - 1) bad_offsets[n_reads] = dst + g;

    const uint2 *slot = stage + (off[r] + 2 * (u64)r);
    const u32 L = len[r];
    u32 bad = 0;
    bool middle = false;
    const bool fits = dst + g <= region_cap;
    for (u32 k = 0; k < g; k++) {
        const uint2 v = slot[k];
        if (fits) bad_regions[dst + k] = v;
        bad += v.y - v.x;
        middle |= (v.x != 0u) & (v.y != L);
    }
    if (!fits) atomicOr(&ctr->region_overflow, 1u);
    read_type[r] = (uint8_t)classify(bad, middle, L, not_cov);
}

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
