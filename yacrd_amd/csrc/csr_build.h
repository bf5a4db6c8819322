// csr_build.h — overlap records in HBM -> the CSR the sweeps consume, built on the GPU.
//
// What FullMemory::add_overlap_and_length does one line at a time on the host (reference
// src/reads2ovl/fullmemory.rs:82-90: push the interval onto its read's vector) as three passes over
// the records, each of them bandwidth-trivial next to the parse that produced the records:
//   count    cnt[read]++ for both reads of every record                      (24 B read per record)
//   scan     offsets = exclusive prefix sums of cnt (u64: a node's batch may exceed 2^32 intervals)
//   scatter  intervals[offsets[read] + --cnt[read]] = (start, end)            (24 B read, 16 B written)
// The order of a read's intervals is whatever the atomics make it; the sweep sorts (the reference's
// first step too, src/stack.rs:66), so results do not depend on it.
#pragma once
#include "device_common.h"

namespace yk {

// Lanes whose read id equals their left neighbour's form a run (PAF is grouped by query: runs of
// tens of lines); the head of a run does ONE atomic for all of it.  `key` must be ~0u on lanes
// without work.  Returns the run's length on its head lane (0 elsewhere) and the lane's rank
// inside its run.
__device__ __forceinline__ u32 run_heads(u32 key, u32 &rank)
{
    const u32 lane = lane_id();
    const u32 prev = (u32)__shfl_up((int)key, 1, 64);
    const bool head = lane == 0 || prev != key;
    const u64 heads = __builtin_amdgcn_ballot_w64(head);
    const u64 below = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
    const u32 my_head = 63u - (u32)__builtin_clzll(below);      // lane 0 is always a head
    const u64 above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const u32 next = above ? lane + 1u + (u32)__builtin_ctzll(above) : 64u;
    rank = lane - my_head;
    return head ? next - lane : 0u;
}

constexpr int kCsrThreads = 256;
constexpr u32 kSkipRead = 0xFFFFFFFEu; // == YACRD_HANDLE_ELSEWHERE (include/yacrd_engine.h)

__global__ __launch_bounds__(kCsrThreads) void csr_count_kernel(const OvlRec *__restrict__ recs, u64 n,
                                                                const u32 *__restrict__ map,
                                                                u64 n_handles, u32 n_reads,
                                                                u32 *cnt, u32 *err)
{
    const u64 stride = (u64)gridDim.x * kCsrThreads;
    // whole wavefronts stay in the loop together (run_heads uses cross-lane operations)
    for (u64 base = (u64)blockIdx.x * kCsrThreads + (threadIdx.x & ~63u); base < n; base += stride) {
        const u64 i = base + lane_id();
        u32 a = ~0u, b = ~0u;
        if (i < n) {
            const uint2 ab = *reinterpret_cast<const uint2 *>(recs + i);
            a = ab.x;
            b = ab.y;
            if (map) {
                a = a < n_handles ? map[a] : ~0u;
                b = b < n_handles ? map[b] : ~0u;
            }
            // kSkipRead: a read that lives on another device (read-partitioned streaming, yacrd_stream_group: a
            // record goes to the devices of both of its reads and each keeps its own half); anything else out of
            // range is an error
            if ((a >= n_reads && a != kSkipRead) || (b >= n_reads && b != kSkipRead)) {
                atomicOr(err, 1u);
                a = b = ~0u;
            }
            if (a == kSkipRead) a = ~0u;
            if (b == kSkipRead) b = ~0u;
        }
        u32 rank;
        const u32 run = run_heads(a, rank);
        if (run && a != ~0u) atomicAdd(&cnt[a], run);
        if (b != ~0u) atomicAdd(&cnt[b], 1u);
    }
}

__global__ __launch_bounds__(kCsrThreads) void csr_scatter_kernel(const OvlRec *__restrict__ recs, u64 n,
                                                                  const u32 *__restrict__ map,
                                                                  u64 n_handles, u32 n_reads,
                                                                  const u64 *__restrict__ off, u32 *cnt,
                                                                  uint2 *__restrict__ iv)
{
    const u64 stride = (u64)gridDim.x * kCsrThreads;
    for (u64 base = (u64)blockIdx.x * kCsrThreads + (threadIdx.x & ~63u); base < n; base += stride) {
        const u64 i = base + lane_id();
        u32 a = ~0u, b = ~0u;
        uint2 ia = make_uint2(0, 0), ib = make_uint2(0, 0);
        if (i < n) {
            const uint2 *p = reinterpret_cast<const uint2 *>(recs + i);
            const uint2 ab = p[0];
            ia = p[1];
            ib = p[2];
            a = ab.x;
            b = ab.y;
            if (map) {
                a = a < n_handles ? map[a] : ~0u;
                b = b < n_handles ? map[b] : ~0u;
            }
            if ((a >= n_reads && a != kSkipRead) || (b >= n_reads && b != kSkipRead)) a = b = ~0u; // reported by the count pass
            if (a == kSkipRead) a = ~0u;
            if (b == kSkipRead) b = ~0u;
        }
        u32 rank;
        const u32 run = run_heads(a, rank);
        u32 top = 0; // the run takes slots [top - run, top) of its read, counted down from the end
        if (run && a != ~0u) top = atomicSub(&cnt[a], run);
        top = (u32)__shfl((int)top, (int)(lane_id() - rank), 64);
        if (a != ~0u) iv[off[a] + (top - 1u - rank)] = ia;
        if (b != ~0u) iv[off[b] + (atomicSub(&cnt[b], 1u) - 1u)] = ib;
    }
}

// ---- exclusive scan u32[n] -> u64[n+1], three small kernels ---------------------------------
constexpr int kScanT = 1024, kScanPer = 4, kScanTile = kScanT * kScanPer;

__device__ __forceinline__ u64 wave_incl_add64(u64 v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}
// exclusive prefix of v over the workgroup; `sc` needs kScanT / 64 words
__device__ __forceinline__ u64 block_excl_add64(u64 v, u64 *sc, u64 &total)
{
    const u64 incl = wave_incl_add64(v);
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 63) sc[wid] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < kScanT / 64; w++) {
        const u64 x = sc[w];
        if (w < wid) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanT) void scan_tile_sums_kernel(const u32 *__restrict__ cnt, u64 n, u64 *part)
{
    __shared__ u64 sc[kScanT / 64];
    const u64 i0 = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanPer;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++)
        if (i0 + k < n) s += cnt[i0 + k];
    u64 tot;
    (void)block_excl_add64(s, sc, tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// one workgroup: part[0..nb) -> exclusive prefixes in place, grand total to *total
__global__ __launch_bounds__(kScanT) void scan_parts_kernel(u64 *part, u64 nb, u64 *total)
{
    __shared__ u64 sc[kScanT / 64];
    u64 carry = 0;
    for (u64 base = 0; base < nb; base += kScanT) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < nb ? part[i] : 0;
        u64 tot;
        const u64 ex = block_excl_add64(v, sc, tot);
        if (i < nb) part[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanT) void scan_tiles_kernel(const u32 *__restrict__ cnt, u64 n,
                                                            const u64 *__restrict__ part, u64 *off)
{
    __shared__ u64 sc[kScanT / 64];
    const u64 i0 = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * kScanPer;
    u32 c[kScanPer];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        c[k] = i0 + k < n ? cnt[i0 + k] : 0u;
        s += c[k];
    }
    u64 tot;
    u64 ex = part[blockIdx.x] + block_excl_add64(s, sc, tot);
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (i0 + k < n) off[i0 + k] = ex;
        ex += c[k];
    }
}

} // namespace yk
