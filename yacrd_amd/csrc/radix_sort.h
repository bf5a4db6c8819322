// radix_sort.h — (u64 key, u32 value) pairs by key: an LSD radix sort for gfx950, eight bits a pass, stable.
//
// What it sorts: the device parser's occupied id-table slots by the first position their id was seen at
// (gpu_paf.hip: first-appearance numbering of the reads, src/reads2ovl/fullmemory.rs:82-90) — a few million pairs whose
// keys have as many significant bits as the file has bytes (x 2): 37 GB of text = 37 bits = five passes.
// (Rounds 3-4 called hipcub::DeviceRadixSort here: a CUDA-API-shaped library on the a1 / a2 path)
//
// A pass over digit d = (key >> shift) & 255, tiles of kRsTile consecutive pairs, one workgroup per tile:
//   rs_hist     the tile's digit counts -> hist[digit][tile] (digit-major, so that ONE exclusive scan over the whole
//               array gives every (digit, tile) its first output slot)
//   scan        csr_build.h's three scan kernels over hist (u32 -> u64)
//   rs_scatter  the tile again, 256 pairs a round in input order: a lane's rank among the lanes of its wavefront
//               with the same digit comes from eight ballots (one per digit bit: the lanes that agree on all of them
//               are its peers, popcount of the peers below it is its rank), the wavefronts' counts per digit meet in
//               LDS, the digit's running offset moves on after every round.  Equal digits keep their input order:
//               that is what makes the passes compose.
// 64-wide throughout: ballots are 64-bit, a tile is 16 rounds of four wavefronts.
#pragma once
#include "device_common.h"

namespace yk {

constexpr int kRsThreads = 256, kRsRounds = 16, kRsTile = kRsThreads * kRsRounds;

__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const u64 *__restrict__ keys, u64 n, u32 shift, u32 n_tiles,
                                                             u32 *__restrict__ hist)
{
    __shared__ u32 s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * kRsTile;
#pragma unroll 4
    for (int r = 0; r < kRsRounds; r++) {
        const u64 i = t0 + (u64)r * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(u32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(u64)threadIdx.x * n_tiles + blockIdx.x] = s_h[threadIdx.x];
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ vals, u64 n,
                                                                u32 shift, u32 n_tiles, const u64 *__restrict__ first,
                                                                u64 *__restrict__ keys_out, u32 *__restrict__ vals_out)
{
    constexpr int NW = kRsThreads / 64;
    __shared__ u64 s_run[256];     // the digit's next output slot
    __shared__ u32 s_wc[NW][256];  // this round's count per wavefront and digit
    const u32 tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    s_run[tid] = first[(u64)tid * n_tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < NW; w++) s_wc[w][tid] = 0;
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * kRsTile;
    const u64 below = (1ull << lane) - 1ull;
    for (int r = 0; r < kRsRounds; r++) { // (uniform)
        const u64 i = t0 + (u64)r * kRsThreads + tid;
        const bool have = i < n;
        u64 key = 0;
        u32 val = 0, d = 0;
        if (have) {
            key = keys[i];
            val = vals[i];
            d = (u32)(key >> shift) & 255u;
        }
        // the lanes of this wavefront that hold the same digit
        u64 peers = __builtin_amdgcn_ballot_w64(have);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const u64 m = __builtin_amdgcn_ballot_w64(((d >> b) & 1u) != 0u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const u32 rank = (u32)__builtin_popcountll(peers & below);
        if (have && rank == 0) s_wc[wv][d] = (u32)__builtin_popcountll(peers); // (the digit's first lane)
        __syncthreads();
        if (have) {
            u64 at = s_run[d] + rank;
#pragma unroll
            for (int w = 0; w < NW; w++) at += (u32)w < wv ? s_wc[w][d] : 0u;
            keys_out[at] = key;
            vals_out[at] = val;
        }
        __syncthreads();
        { // digit `tid`: move on, clear the round's counts
            u32 c = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                c += s_wc[w][tid];
                s_wc[w][tid] = 0;
            }
            s_run[tid] += c;
        }
        __syncthreads();
    }
}

} // namespace yk
