// sweep_wave.h — small class (<= 1024 events): one read per wavefront, everything in registers.
//
// Same event formulation as sweep_lds.h (reference src/stack.rs:61-139 for regular reads), but
// the dominant cost — sorting the 2n event keys — runs as a bitonic network over VGPRs:
//   * a lane holds K consecutive keys of the sequence (element index = lane*K + r), so strides
//     < K are register-to-register min/max and the post-sort scans are lane-sequential;
//   * strides >= K exchange between lanes: DPP moves for lane xor 1, 2, 8 (VALU only),
//     ds_swizzle for xor 4, 16 and ds_bpermute for xor 32 (the LDS crossbar is otherwise idle,
//     no LDS memory is touched), each followed by ONE v_med3_u32: med3(x, partner, 0) = min,
//     med3(x, partner, ~0) = max, the third operand being a per-lane constant that encodes
//     "upper lane of the pair" xor "descending block".
// No LDS allocation, no barriers: a 256-thread workgroup is four independent wavefronts.
// Pads are end-like keys (0xFFFFFFFE) and depth compares are signed, so nothing after the sort
// needs a validity mask.
#pragma once
#include "device_common.h"

namespace yk {

constexpr u32 kPadKey = 0xFFFFFFFEu;

__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c)
{
    return max(min(a, b), min(max(a, b), c)); // -> v_med3_u32
}

// DPP controls (gfx9): quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_ror:8, row_shr:n, wave_shr:1
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR8 = 0x128;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138, DPP_BCAST15 = 0x142,
              DPP_BCAST31 = 0x143;

// value of lane ^ D.  XM (cross-lane mode): 0 = DPP for xor 1/2/8, LDS crossbar for 4/16/32;
// 1 = LDS crossbar (ds_swizzle / ds_bpermute) for every stride (fewest VALU instructions).
template <int D, int XM>
__device__ __forceinline__ u32 lane_xor(u32 x, u32 bperm_addr32)
{
    if constexpr (D == 32) return (u32)__builtin_amdgcn_ds_bpermute((int)bperm_addr32, (int)x);
    else if constexpr (XM == 1 || D == 4 || D == 16)
        return (u32)__builtin_amdgcn_ds_swizzle((int)x, (D << 10) | 0x1F);
    else if constexpr (D == 1) return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_XOR1, 0xF, 0xF, false);
    else if constexpr (D == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_XOR2, 0xF, 0xF, false);
    else return (u32)__builtin_amdgcn_mov_dpp((int)x, DPP_ROR8, 0xF, 0xF, false);
}

// wave64 inclusive scans on DPP (row_shr 1/2/4/8, row_bcast 15/31); identity 0
#define YK_DPP0(v, ctrl, rm) (u32) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rm, 0xF, true)
__device__ __forceinline__ u32 wscan_add(u32 v)
{
    v += YK_DPP0(v, DPP_ROW_SHR1, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR2, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR4, 0xF);
    v += YK_DPP0(v, DPP_ROW_SHR8, 0xF);
    v += YK_DPP0(v, DPP_BCAST15, 0xA);
    v += YK_DPP0(v, DPP_BCAST31, 0xC);
    return v;
}
__device__ __forceinline__ u32 wscan_max(u32 v)
{
    v = max(v, YK_DPP0(v, DPP_ROW_SHR1, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR2, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR4, 0xF));
    v = max(v, YK_DPP0(v, DPP_ROW_SHR8, 0xF));
    v = max(v, YK_DPP0(v, DPP_BCAST15, 0xA));
    v = max(v, YK_DPP0(v, DPP_BCAST31, 0xC));
    return v;
}
__device__ __forceinline__ u32 wshift_up1(u32 v) { return YK_DPP0(v, DPP_WAVE_SHR1, 0xF); }

struct LaneConst {
    u32 k[7];   // k[i] = (lane & (1<<i)) ? ~0u : 0u for i < 6; k[6] = 0
    u32 addr32; // byte address of lane ^ 32 for ds_bpermute
};

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// ---- bitonic sort of 64*K keys held as x[K] per lane, element index = lane*K + r ------------
template <int K, int M, int J, int XM>
__device__ __forceinline__ void bitonic_step(u32 (&x)[K], const LaneConst &lc)
{
    constexpr int P = 64 * K;
    constexpr bool lane_dir = (M >= K) && (M < P); // direction bit lives in the lane id
    const u32 dirm = lane_dir ? lc.k[ilog2c(M / K)] : 0u;
    if constexpr (J >= K) { // partner in another lane
        constexpr int D = J / K;
        const u32 sel = lc.k[ilog2c(D)] ^ dirm; // ~0: this lane keeps the larger key
#pragma unroll
        for (int r = 0; r < K; r++) {
            const u32 t = lane_xor<D, XM>(x[r], lc.addr32);
            x[r] = umed3(x[r], t, sel);
        }
    } else { // partner in another register of the same lane
#pragma unroll
        for (int r = 0; r < K; r++) {
            if ((r & J) == 0) {
                const u32 a = x[r], b = x[r | J];
                if constexpr (M < K) {
                    const bool desc = (r & M) != 0;
                    x[r] = desc ? max(a, b) : min(a, b);
                    x[r | J] = desc ? min(a, b) : max(a, b);
                } else if constexpr (lane_dir) {
                    x[r] = umed3(a, b, dirm);
                    x[r | J] = umed3(a, b, ~dirm);
                } else {
                    x[r] = min(a, b);
                    x[r | J] = max(a, b);
                }
            }
        }
    }
}
template <int K, int M, int J, int XM>
__device__ __forceinline__ void bitonic_level(u32 (&x)[K], const LaneConst &lc)
{
    bitonic_step<K, M, J, XM>(x, lc);
    if constexpr (J > 1) bitonic_level<K, M, J / 2, XM>(x, lc);
}
template <int K, int M, int XM>
__device__ __forceinline__ void bitonic_sort(u32 (&x)[K], const LaneConst &lc)
{
    bitonic_level<K, M, M / 2, XM>(x, lc);
    if constexpr (M < 64 * K) bitonic_sort<K, M * 2, XM>(x, lc);
}

// ---- one read, K keys per lane --------------------------------------------------------------
// Returns false when the read has a degenerate interval (caller queues it for the general path).
template <int K, int XM>
__device__ __forceinline__ bool sweep_wave_read(const uint2 *__restrict__ iv, u32 n, u32 len,
                                                u32 cov, uint2 *slot, u32 *count_out,
                                                const LaneConst &lc)
{
    const u32 lane = lane_id();
    const u32 m = 2 * n;
    const i32 c = (i32)min(cov, 0x7FFFFFFFu);

    // ---- coalesced interval loads (8 B/lane), keys straight into registers
    u32 x[K];
    u32 bad = 0;
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
        const u32 i = lane + 64u * j;
        uint2 v = make_uint2(0x7FFFFFFFu, 0x7FFFFFFFu);
        if (i < n) v = iv[i];
        const bool pad = i >= n;
        bad |= (!pad && (v.x >= v.y || v.y >= 0x7FFFFFFFu)) ? 1u : 0u;
        x[2 * j] = pad ? kPadKey : ((v.x << 1) | 1u);
        x[2 * j + 1] = pad ? kPadKey : (v.y << 1);
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) return false;

    bitonic_sort<K, 2, XM>(x, lc);

    // ---- pass 1: depth carried into each lane
    u32 delta = 0;
#pragma unroll
    for (int r = 0; r < K; r++) delta += (x[r] & 1u) ? 1u : 0xFFFFFFFFu;
    const u32 dincl = wscan_add(delta);
    const i32 depth_in = (i32)(dincl - delta);

    // ---- pass 2: flagged ends / low starts, per-lane last of each
    u32 fbits = 0, lbits = 0, mf = 0, ml = 0;
    i32 d = depth_in;
#pragma unroll
    for (int r = 0; r < K; r++) {
        const u32 key = x[r];
        const bool is_s = (key & 1u) != 0;
        const bool low = is_s && d <= c;
        const bool fl = !is_s && d > c;
        ml = low ? key : ml;
        mf = fl ? key : mf;
        lbits |= (low ? 1u : 0u) << r;
        fbits |= (fl ? 1u : 0u) << r;
        d += is_s ? 1 : -1;
    }
    const u32 mf_incl = wscan_max(mf), ml_incl = wscan_max(ml);
    const u32 mf_in = wshift_up1(mf_incl), ml_in = wshift_up1(ml_incl);
    const u32 mf_t = (u32)__builtin_amdgcn_readlane((int)mf_incl, 63);
    const u32 ml_t = (u32)__builtin_amdgcn_readlane((int)ml_incl, 63);

    // ---- pass 3: regions closed in this lane; tail rule candidates (stack.rs:93-105)
    u32 cnt = 0, first_b = 0, first_e = 0, cand = kNoKey;
    {
        u32 cmf = mf_in, cml = ml_in;
        d = depth_in;
#pragma unroll
        for (int r = 0; r < K; r++) {
            const u32 key = x[r];
            const bool fl = (fbits >> r) & 1u, low = (lbits >> r) & 1u;
            const bool close = fl && cml > cmf && !(cmf == 0 && (cml >> 1) == 0);
            if (close && cnt == 0) {
                first_b = cmf >> 1;
                first_e = cml >> 1;
            }
            cnt += close ? 1u : 0u;
            // an end is in the tail when every start precedes it: starts before = (idx + depth)/2
            const bool tail = fl && (lane * K + r + (u32)d == m) && (key >> 1) >= len;
            cand = (tail && cand == kNoKey) ? (key >> 1) : cand;
            cmf = fl ? key : cmf;
            cml = low ? key : cml;
            d += (key & 1u) ? 1 : -1;
        }
    }
    u32 g_closed = 0;
    const u64 any_close = __builtin_amdgcn_ballot_w64(cnt != 0);
    if (any_close) {
        const u32 cincl = wscan_add(cnt);
        u32 pos = cincl - cnt;
        g_closed = (u32)__builtin_amdgcn_readlane((int)cincl, 63);
        if (cnt == 1) {
            slot[pos] = make_uint2(first_b, first_e);
        } else if (cnt > 1) { // several regions close inside one lane: replay it
            u32 cmf = mf_in, cml = ml_in;
#pragma unroll
            for (int r = 0; r < K; r++) {
                const u32 key = x[r];
                const bool fl = (fbits >> r) & 1u, low = (lbits >> r) & 1u;
                if (fl && cml > cmf && !(cmf == 0 && (cml >> 1) == 0))
                    slot[pos++] = make_uint2(cmf >> 1, cml >> 1);
                cmf = fl ? key : cmf;
                cml = low ? key : cml;
            }
        }
    }
    u32 min_ge = kNoKey;
    const u64 any_cand = __builtin_amdgcn_ballot_w64(cand != kNoKey);
    if (any_cand) // keys ascend with the lane id: the first lane holding a candidate has the minimum
        min_ge = (u32)__builtin_amdgcn_readlane((int)cand, (int)__builtin_ctzll(any_cand));
    if (lane == 0) *count_out = finish_read(slot, g_closed, mf_t, ml_t, min_ge, len);
    return true;
}

// One kernel per K so the small K get small register footprints (8 waves/SIMD).
template <int K, int XM>
__global__ __launch_bounds__(256) void sweep_wave_kernel(SweepArgs a)
{
    const u32 lane = lane_id();
    LaneConst lc;
#pragma unroll
    for (int i = 0; i < 6; i++) lc.k[i] = (lane & (1u << i)) ? 0xFFFFFFFFu : 0u;
    lc.k[6] = 0;
    lc.addr32 = (lane ^ 32u) << 2;

    const u32 list_n = *a.list_n;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 nwaves = gridDim.x * 4u;
    for (u32 w = blockIdx.x * 4u + wave; w < list_n; w += nwaves) {
        const u32 r = a.list[w];
        const u64 o = a.off[r];
        const u32 n = (u32)(a.off[r + 1] - o);
        const u32 len = a.len[r];
        uint2 *slot = a.stage + (o + 2 * (u64)r);
        bool ok = true;
        if (n == 0) {
            if (lane == 0) {
                u32 g = 0;
                if (len != 0) slot[g++] = make_uint2(0, len);
                a.counts[r] = g;
            }
        } else {
            ok = sweep_wave_read<K, XM>(a.iv + o, n, len, a.cov, slot, a.counts + r, lc);
        }
        if (!ok && lane == 0) { // degenerate interval: the exact general path takes the read
            a.rej_list[atomicAdd(a.rej_count, 1u)] = r;
            a.counts[r] = 0;
        }
    }
}

template <int K>
inline void launch_sweep_wave(const SweepArgs &sa, u32 n_reads, int num_cu, hipStream_t stream,
                              int xlane_mode)
{
    // One read per wavefront, four wavefronts per workgroup; the dispatcher balances the tail
    // better than persistent waves did (76 vs 82 us on configs[1], profiles/README.md).
    // YACRD_WAVE_BLOCKS_PER_CU=n caps the grid at n workgroups per CU (grid-stride loop) for A/B.
    static const int per_cu = [] {
        const char *s = getenv("YACRD_WAVE_BLOCKS_PER_CU");
        return s ? atoi(s) : 0;
    }();
    const u64 want = ((u64)n_reads + 3) / 4;
    const u64 cap = per_cu > 0 ? (u64)num_cu * (u64)per_cu : want;
    const u32 grid = (u32)(want < cap ? want : cap);
    if (xlane_mode == 1)
        hipLaunchKernelGGL((sweep_wave_kernel<K, 1>), dim3(grid ? grid : 1), dim3(256), 0, stream, sa);
    else
        hipLaunchKernelGGL((sweep_wave_kernel<K, 0>), dim3(grid ? grid : 1), dim3(256), 0, stream, sa);
}

} // namespace yk
