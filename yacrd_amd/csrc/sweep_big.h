// sweep_big.h — regular reads too large for one workgroup's LDS (> 16 384 intervals): the whole
// device works on them.  Same event formulation as sweep_lds.h (reference src/stack.rs:61-139),
// but the 2n keys of a read live in global memory as `P/C` chunks of C = 8192 keys and every
// step is a grid over chunks:
//   fill        event keys (position<<2 | class, device_common.h) + end-like pads up to
//               P = pow2 >= 2n; flags reads the keys cannot express (exact general kernel instead)
//   sort        segmented bitonic: chunk_sort (levels <= C in LDS), then per level M = 2C..P:
//               global_stage for strides >= C, lds_merge for strides < C.  A read stops at M = P.
//   sweep       the chunked passes of sweep_lds.h with the carries between chunks going through
//               small per-chunk arrays and one-workgroup-per-read scans:
//               A delta -> S1 depth carries -> B last flagged / last low -> S2 max carries
//               -> C closings count + tail candidates -> S3 offsets, finish_read -> D write
// Launch count is ~50 for a 2^21-key read, each touching every key once: microseconds of launch
// latency instead of the >100 ms a single workgroup needs for the same read.
#pragma once
#include "device_common.h"

namespace yk {

constexpr int kBigC = 8192;  // keys per chunk
constexpr int kBigT = 256;   // threads per chunk
constexpr int kBigKT = kBigC / kBigT; // keys per thread in the sweep passes

struct BigSeg {    // one big read
    u64 key_off;   // first key in the key buffer
    u64 iv_off;    // first interval in the CSR
    u32 P;         // padded key count (power of two, >= kBigC)
    u32 n;         // intervals
    u32 len;
    u32 read;      // read id
    u32 chunk_off; // first chunk
    u32 pad;
};

struct BigArgs {
    const BigSeg *seg;
    const u32 *chunk_seg; // chunk -> segment
    u32 *keys;
    const uint2 *iv;
    u32 n_chunks;
    u32 n_segs;
    u32 cov;
    // per chunk
    u32 *c_delta, *c_depth_in, *c_mf, *c_ml, *c_mf_in, *c_ml_in, *c_cnt, *c_pos, *c_cand;
    u32 *seg_bad; // per segment: degenerate interval seen
    uint2 *stage;
    u32 *counts;
};

__global__ __launch_bounds__(kBigT) void big_fill_kernel(BigArgs a)
{
    const u32 c = blockIdx.x;
    const BigSeg s = a.seg[a.chunk_seg[c]];
    const u32 e0 = (c - s.chunk_off) * kBigC; // first local key index of this chunk
    u32 *out = a.keys + s.key_off + e0;
    u32 bad = 0, nz = 0;
    for (u32 t = threadIdx.x; t < kBigC / 2; t += kBigT) { // one interval -> two keys
        const u32 i = e0 / 2 + t;
        u32 ks = kNoKey - 1, ke = kNoKey - 1; // end-like pads (0xFFFFFFFE)
        if (i < s.n) make_event_keys(a.iv[s.iv_off + i], ks, ke, bad, nz);
        out[2 * t] = ks;
        out[2 * t + 1] = ke;
    }
    if (__syncthreads_or((int)bad) && threadIdx.x == 0) a.seg_bad[a.chunk_seg[c]] = 1;
}

// levels M <= C inside LDS; direction bit (e & M) uses the read-local index e
__global__ __launch_bounds__(kBigT) void big_chunk_sort_kernel(BigArgs a)
{
    __shared__ u32 k[kBigC];
    const u32 c = blockIdx.x;
    const BigSeg s = a.seg[a.chunk_seg[c]];
    const u32 e0 = (c - s.chunk_off) * kBigC;
    u32 *g = a.keys + s.key_off + e0;
    for (u32 t = threadIdx.x; t < kBigC; t += kBigT) k[t] = g[t];
    __syncthreads();
    for (u32 M = 2; M <= (u32)kBigC; M <<= 1) {
        for (u32 j = M >> 1; j > 0; j >>= 1) {
            for (u32 p = threadIdx.x; p < kBigC / 2; p += kBigT) {
                const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const bool up = ((e0 + i) & M) == 0 || M == s.P;
                const u32 x = k[i], y = k[l];
                if ((x > y) == up) {
                    k[i] = y;
                    k[l] = x;
                }
            }
            __syncthreads();
        }
    }
    for (u32 t = threadIdx.x; t < kBigC; t += kBigT) g[t] = k[t];
}

// one compare-exchange per thread, stride j >= C, level M; reads with P < M are finished
__global__ __launch_bounds__(kBigT) void big_global_stage_kernel(BigArgs a, u32 M, u32 j)
{
    const u32 gidx = blockIdx.x * kBigT + threadIdx.x; // pair index over all chunks
    const u32 c = gidx / (kBigC / 2);
    if (c >= a.n_chunks) return;
    const BigSeg s = a.seg[a.chunk_seg[c]];
    if (s.P < M) return;
    const u32 p = (c - s.chunk_off) * (kBigC / 2) + (gidx % (kBigC / 2));
    const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
    const bool up = (i & M) == 0 || M == s.P;
    u32 *k = a.keys + s.key_off;
    const u32 x = k[i], y = k[l];
    if ((x > y) == up) {
        k[i] = y;
        k[l] = x;
    }
}

// strides C/2 .. 1 of level M >= 2C inside LDS (direction uniform per chunk)
__global__ __launch_bounds__(kBigT) void big_lds_merge_kernel(BigArgs a, u32 M)
{
    __shared__ u32 k[kBigC];
    const u32 c = blockIdx.x;
    const BigSeg s = a.seg[a.chunk_seg[c]];
    if (s.P < M) return;
    const u32 e0 = (c - s.chunk_off) * kBigC;
    u32 *g = a.keys + s.key_off + e0;
    for (u32 t = threadIdx.x; t < kBigC; t += kBigT) k[t] = g[t];
    __syncthreads();
    const bool up = (e0 & M) == 0 || M == s.P;
    for (u32 j = kBigC / 2; j > 0; j >>= 1) {
        for (u32 p = threadIdx.x; p < kBigC / 2; p += kBigT) {
            const u32 i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
            const u32 x = k[i], y = k[l];
            if ((x > y) == up) {
                k[i] = y;
                k[l] = x;
            }
        }
        __syncthreads();
    }
    for (u32 t = threadIdx.x; t < kBigC; t += kBigT) g[t] = k[t];
}

// ---- sweep passes: thread t of chunk c owns keys [t*KT, t*KT+KT) of the chunk ---------------
struct BigChunk {
    BigSeg s;
    const u32 *k; // this thread's keys
    u32 e0;       // read-local index of this thread's first key
};
__device__ __forceinline__ BigChunk big_chunk(const BigArgs &a)
{
    BigChunk b;
    const u32 c = blockIdx.x;
    b.s = a.seg[a.chunk_seg[c]];
    b.e0 = (c - b.s.chunk_off) * kBigC + threadIdx.x * kBigKT;
    b.k = a.keys + b.s.key_off + b.e0;
    return b;
}

__global__ __launch_bounds__(kBigT) void big_pass_a_kernel(BigArgs a)
{
    __shared__ u32 sc[kBigT / 64];
    const BigChunk b = big_chunk(a);
    u32 delta = 0, dup = 0;
    u32 prev = b.e0 ? b.k[-1] : 0u; // last key before this thread's run (same read)
    for (int q = 0; q < kBigKT; q++) {
        const u32 key = b.k[q];
        delta += (key & 1u) ? 1u : 0xFFFFFFFFu;
        dup |= (key == prev && (key & 3u) == 1u && key != 1u) ? 1u : 0u; // two zero-length intervals, one position
        prev = key;
    }
    u32 tot;
    block_excl_add<kBigT>(delta, sc, tot);
    if (threadIdx.x == 0) a.c_delta[blockIdx.x] = tot;
    if (__syncthreads_or((int)dup) && threadIdx.x == 0) a.seg_bad[a.chunk_seg[blockIdx.x]] = 1;
}

// one workgroup per read: carries between its chunks.  which: 0 depth (sum), 1 mf/ml (max),
// 2 closings (sum) + tail candidates (min) + finish_read
__global__ __launch_bounds__(kBigT) void big_scan_kernel(BigArgs a, u32 which)
{
    __shared__ u32 sc[kBigT / 64];
    __shared__ u32 carry[3];
    const BigSeg s = a.seg[blockIdx.x];
    const u32 nc = s.P / kBigC;
    if (threadIdx.x < 3) carry[threadIdx.x] = threadIdx.x == 2 ? kNoKey : 0;
    __syncthreads();
    for (u32 base = 0; base < nc; base += kBigT) {
        const u32 i = base + threadIdx.x, c = s.chunk_off + i;
        const bool in = i < nc;
        u32 tot;
        if (which == 0) {
            const u32 v = in ? a.c_delta[c] : 0;
            const u32 ex = block_excl_add<kBigT>(v, sc, tot);
            if (in) a.c_depth_in[c] = carry[0] + ex;
            __syncthreads();
            if (threadIdx.x == 0) carry[0] += tot;
        } else if (which == 1) {
            const u32 vf = in ? a.c_mf[c] : 0, vl = in ? a.c_ml[c] : 0;
            const u32 exf = block_excl_max<kBigT>(vf, sc, tot);
            u32 tot2;
            const u32 exl = block_excl_max<kBigT>(vl, sc, tot2);
            if (in) {
                a.c_mf_in[c] = max(carry[0], exf);
                a.c_ml_in[c] = max(carry[1], exl);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                carry[0] = max(carry[0], tot);
                carry[1] = max(carry[1], tot2);
            }
        } else {
            const u32 v = in ? a.c_cnt[c] : 0;
            const u32 ex = block_excl_add<kBigT>(v, sc, tot);
            const u32 mn = block_min<kBigT>(in ? a.c_cand[c] : kNoKey, sc);
            if (in) a.c_pos[c] = carry[0] + ex;
            __syncthreads();
            if (threadIdx.x == 0) {
                carry[0] += tot;
                carry[2] = min(carry[2], mn);
            }
        }
        __syncthreads();
    }
    if (which == 2 && threadIdx.x == 0) {
        // totals of the max scans = carries out of the last chunk
        const u32 last = s.chunk_off + nc - 1;
        const u32 mf_t = max(a.c_mf_in[last], a.c_mf[last]), ml_t = max(a.c_ml_in[last], a.c_ml[last]);
        uint2 *slot = a.stage + (s.iv_off + 2 * (u64)s.read);
        a.counts[s.read] = finish_read(slot, carry[0], mf_t ? (mf_t ^ 2u) : 0u, ml_t, carry[2], s.len);
    }
}

// B: last flagged end / last low start of the chunk.  C: closings + tail candidates.  D: write.
template <int PASS>
__global__ __launch_bounds__(kBigT) void big_pass_kernel(BigArgs a)
{
    __shared__ u32 sc[kBigT / 64];
    const BigChunk b = big_chunk(a);
    const u32 c = blockIdx.x;
    const i32 cov = (i32)min(a.cov, 0x7FFFFFFFu);
    u32 delta = 0;
    for (int q = 0; q < kBigKT; q++) delta += (b.k[q] & 1u) ? 1u : 0xFFFFFFFFu;
    u32 tot;
    const i32 depth_in = (i32)(a.c_depth_in[c] + block_excl_add<kBigT>(delta, sc, tot));

    u32 mf = 0, ml = 0; // mf: flagged ends in the flipped domain (device_common.h)
    i32 d = depth_in;
    for (int q = 0; q < kBigKT; q++) {
        const u32 key = b.k[q];
        const bool is_s = key & 1u, gt = d > cov;
        ml = (is_s && !gt) ? key : ml;
        mf = (!is_s && gt) ? max(mf, key ^ 2u) : mf;
        d += is_s ? 1 : -1;
    }
    u32 mf_t, ml_t;
    const u32 mf_ex = block_excl_max<kBigT>(mf, sc, mf_t);
    const u32 ml_ex = block_excl_max<kBigT>(ml, sc, ml_t);
    if (PASS == 0) {
        if (threadIdx.x == 0) {
            a.c_mf[c] = mf_t;
            a.c_ml[c] = ml_t;
        }
        return;
    }
    u32 cmf = max(max(a.c_mf_in[c], mf_ex), kNoFlag), cml = max(a.c_ml_in[c], ml_ex);
    const u32 m = 2 * b.s.n;
    const u32 len_key = b.s.len > kMaxKeyPos ? 0xFFFFFFFFu : (b.s.len << kKeyShift);
    uint2 *slot = a.stage + (b.s.iv_off + 2 * (u64)b.s.read);
    u32 cnt = 0, cand = kNoKey;
    if (PASS == 1) {
        d = depth_in;
        for (int q = 0; q < kBigKT; q++) {
            const u32 key = b.k[q];
            const bool is_s = key & 1u, gt = d > cov, fl = !is_s && gt;
            const bool eff = fl && (key ^ 2u) > cmf;
            cnt += (eff && cml > (cmf ^ 2u)) ? 1u : 0u;
            if (fl && (b.e0 + q + (u32)d == m) && key >= len_key) cand = min(cand, key >> kKeyShift);
            cmf = eff ? (key ^ 2u) : cmf;
            cml = (is_s && !gt) ? key : cml;
            d += is_s ? 1 : -1;
        }
        u32 ctot;
        block_excl_add<kBigT>(cnt, sc, ctot);
        cand = block_min<kBigT>(cand, sc);
        if (threadIdx.x == 0) {
            a.c_cnt[c] = ctot;
            a.c_cand[c] = cand;
        }
    } else {
        if (a.c_cnt[c] == 0) return; // uniform per workgroup
        d = depth_in;
        u32 cm2 = cmf, cl2 = cml;
        for (int q = 0; q < kBigKT; q++) {
            const u32 key = b.k[q];
            const bool is_s = key & 1u, gt = d > cov, fl = !is_s && gt;
            const bool eff = fl && (key ^ 2u) > cm2;
            cnt += (eff && cl2 > (cm2 ^ 2u)) ? 1u : 0u;
            cm2 = eff ? (key ^ 2u) : cm2;
            cl2 = (is_s && !gt) ? key : cl2;
            d += is_s ? 1 : -1;
        }
        u32 ctot;
        u32 pos = a.c_pos[c] + block_excl_add<kBigT>(cnt, sc, ctot);
        if (cnt) {
            d = depth_in;
            for (int q = 0; q < kBigKT; q++) {
                const u32 key = b.k[q];
                const bool is_s = key & 1u, gt = d > cov, fl = !is_s && gt;
                const bool eff = fl && (key ^ 2u) > cmf;
                if (eff && cml > (cmf ^ 2u))
                    slot[pos++] = make_uint2((cmf ^ 2u) >> kKeyShift, cml >> kKeyShift);
                cmf = eff ? (key ^ 2u) : cmf;
                cml = (is_s && !gt) ? key : cml;
                d += is_s ? 1 : -1;
            }
        }
    }
}

// Launch everything for prepared segments (host-built tables already on the device).
inline void launch_big(const BigArgs &a, u32 max_P, hipStream_t st)
{
    const dim3 gc(a.n_chunks), gs(a.n_segs), blk(kBigT);
    hipLaunchKernelGGL(big_chunk_sort_kernel, gc, blk, 0, st, a);
    const u32 pair_blocks = (u32)(((u64)a.n_chunks * (kBigC / 2) + kBigT - 1) / kBigT);
    for (u32 M = 2 * kBigC; M <= max_P && M != 0; M <<= 1) {
        for (u32 j = M >> 1; j >= (u32)kBigC; j >>= 1)
            hipLaunchKernelGGL(big_global_stage_kernel, dim3(pair_blocks), blk, 0, st, a, M, j);
        hipLaunchKernelGGL(big_lds_merge_kernel, gc, blk, 0, st, a, M);
    }
    hipLaunchKernelGGL(big_pass_a_kernel, gc, blk, 0, st, a);
    hipLaunchKernelGGL(big_scan_kernel, gs, blk, 0, st, a, 0u);
    hipLaunchKernelGGL(big_pass_kernel<0>, gc, blk, 0, st, a);
    hipLaunchKernelGGL(big_scan_kernel, gs, blk, 0, st, a, 1u);
    hipLaunchKernelGGL(big_pass_kernel<1>, gc, blk, 0, st, a);
    hipLaunchKernelGGL(big_scan_kernel, gs, blk, 0, st, a, 2u);
    hipLaunchKernelGGL(big_pass_kernel<2>, gc, blk, 0, st, a);
}

} // namespace yk
