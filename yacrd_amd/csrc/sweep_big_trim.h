// sweep_big_trim.h — reads with more than 16 384 intervals, first attempt: the pile-trimming
// pre-filter (sweep_lds.h) run by the WHOLE device, so that what is left of a read
// fits one workgroup's LDS.  A dovetail pile-up of 763 225 intervals (configs[3]'s largest read)
// keeps ~10 event keys; the segmented device-wide sort of sweep_big.h (~50 launches, every key
// touched in each) is then only needed for reads the filter cannot thin: low coverage throughout,
// or intervals the filter does not take (zero-length, an end beyond the read).
//   hist     grid over chunks of 4096 intervals: starts / ends per bin, privatised in LDS, flushed
//            to the read's table in global memory (non-zero counters only)
//   plan     one workgroup per read: depth scan over the 2048 bins, what every bin keeps
//            (lds_trim_plan's rules), the stand-in keys, slot ranges per (bin, type)
//   scatter  grid over the same chunks: every key asks its bin's cursor (looked at before it is
//            asked: a cursor past its limit stays there) and the survivors land in the read's
//            compact key buffer
//   sweep    one 1024-thread workgroup per read: keys to LDS, lds_sort_and_sweep
// Reference semantics: src/stack.rs:61-139 through the event formulation of sweep_lds.h.
#pragma once
#include "device_common.h"
#include "sweep_lds.h"

namespace yk {

constexpr int kBtNB = 1024, kBtF = 512, kBtSeq = 2 * kBtF + kBtNB; // bins per read
constexpr int kBtT = 256;          // threads of hist / plan / scatter
constexpr int kBtChunk = 4096;     // intervals per workgroup in hist / scatter
constexpr u32 kBtCap = 16384;      // event keys a thinned read may keep (half the 1024-thread LDS array)
constexpr int PER8 = kBtSeq / kBtT; // bins per thread of the plan kernel
using BtGeo = TrimGeo<kBtNB, kBtF>;

enum { BT_PLAIN_NO = 1u, BT_BAD = 2u, BT_TRIMMED = 4u }; // BtSeg.flags

struct BtSeg {
    u64 iv_off; // first interval in the CSR
    u32 n;      // intervals
    u32 len;
    u32 read;
    u32 chunk_off; // first chunk
    u32 flags;     // BT_* (device-written)
    u32 m_new;     // keys kept (device-written)
    u32 n_zl;      // zero-length intervals (device-written)
    u32 pad;
};

struct BtArgs {
    BtSeg *seg;
    const u32 *chunk_seg; // chunk -> segment
    const uint2 *iv;
    u32 n_chunks, n_segs, cov;
    u32 *hist;  // [n_segs][3][kBtSeq]: starts, ends, zero-length intervals (one-position bins only)
    u32 *cur;   // [n_segs][3][kBtSeq]: next slot of (type, bin): 0 ends, 1 starts, 2 zero-length keys
    u32 *lim;   // [n_segs][3][kBtSeq]: one past its last slot
    u32 *tkeys; // [n_segs][kBtCap]
    u32 *drop;  // [n_segs][3 * kBtSeq / 32]: bit = this (type, bin) keeps nothing
    uint2 *stage;
    u32 *counts;
};

__global__ __launch_bounds__(kBtT) void bt_hist_kernel(BtArgs a)
{
    __shared__ u32 hs[kBtSeq], he[kBtSeq], hz[kBtSeq];
    const u32 c = blockIdx.x, si = a.chunk_seg[c];
    const BtSeg s = a.seg[si];
    if (s.len > kMaxKeyPos) { // positions beyond the key range: not for the filter
        if (threadIdx.x == 0) atomicOr(&a.seg[si].flags, (u32)BT_PLAIN_NO);
        return;
    }
    const BtGeo g = BtGeo::make(s.len);
    for (u32 b = threadIdx.x; b < (u32)kBtSeq; b += kBtT) hs[b] = he[b] = hz[b] = 0;
    __syncthreads();
    const u32 i0 = (c - s.chunk_off) * kBtChunk, i1 = min(i0 + (u32)kBtChunk, s.n);
    const uint2 *iv = a.iv + s.iv_off;
    u32 flags = 0, n_zl = 0;
    for (u32 i = i0 + threadIdx.x; i < i1; i += 4 * kBtT) {
        uint2 v4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v4[j] = iv[min(i + (u32)j * kBtT, i1 - 1u)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + (u32)j * kBtT >= i1) break;
            const uint2 v = v4[j];
            if (v.x <= v.y && v.y <= s.len && v.y <= kMaxKeyPos) {
                const u32 is = g.idx((v.x << kKeyShift) | 3u);
                if (v.x == v.y && g.uniform(is)) atomicAdd(&hz[is], 1u);
                else {
                    atomicAdd(&hs[is], 1u);
                    atomicAdd(&he[g.idx(v.y << kKeyShift)], 1u);
                }
                n_zl += v.x == v.y ? 1u : 0u;
            } else {
                flags |= (v.x > v.y || v.y > kMaxKeyPos) ? BT_BAD : BT_PLAIN_NO;
            }
        }
    }
    __syncthreads();
    u32 *gh = a.hist + (size_t)si * 3 * kBtSeq;
    for (u32 b = threadIdx.x; b < (u32)kBtSeq; b += kBtT) {
        if (hs[b]) atomicAdd(&gh[b], hs[b]);
        if (he[b]) atomicAdd(&gh[kBtSeq + b], he[b]);
        if (hz[b]) atomicAdd(&gh[2 * kBtSeq + b], hz[b]);
    }
    if (flags) atomicOr(&a.seg[si].flags, flags);
    if (n_zl) atomicAdd(&a.seg[si].n_zl, n_zl);
}

// one workgroup per read, eight consecutive bins per thread (lds_trim_plan with the table in
// global memory and 32-bit depths)
__global__ __launch_bounds__(kBtT) void bt_plan_kernel(BtArgs a)
{
    constexpr int PER = kBtSeq / kBtT;
    __shared__ u32 sc[kBtT / 64 + 1];
    __shared__ i32 a_of[kBtSeq]; // depth after the kept block of each opaque bin
    const u32 si = blockIdx.x, tid = threadIdx.x;
    BtSeg &sg = a.seg[si];
    const BtSeg s = sg;
    if (s.flags & (BT_BAD | BT_PLAIN_NO)) return; // uniform: the segmented sort / exact path takes it
    const BtGeo g = BtGeo::make(s.len);
    const u32 n_seq = g.n_seq();
    const i32 c = (i32)min(a.cov, 0x3FFFFFFFu);
    const u32 *gh = a.hist + (size_t)si * 3 * kBtSeq;
    u32 *cur = a.cur + (size_t)si * 3 * kBtSeq, *lim = a.lim + (size_t)si * 3 * kBtSeq;
    u32 *tk = a.tkeys + (size_t)si * kBtCap;

    i32 S[PER], E[PER];
    u32 delta = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        S[k] = (i32)gh[PER * tid + k];
        E[k] = (i32)gh[kBtSeq + PER * tid + k];
        delta += (u32)(S[k] - E[k]);
    }
    u32 tot;
    i32 D = (i32)block_excl_add<kBtT>(delta, sc, tot);
    i32 keep_e[PER], keep_s[PER], keep_z[PER], B[PER];
    bool opaque[PER];
    u32 last_own = 0; // index + 1 of this thread's last opaque bin
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = PER * tid + k;
        if (g.uniform(i)) {
            const i32 n_de = min(max(D - c - 1, 0), E[k]);
            keep_e[k] = E[k] - n_de;
            keep_s[k] = min(max(c + 1 - (D - E[k]), 0), S[k]);
            keep_z[k] = (D - E[k] <= c) ? 2 * (i32)gh[2 * kBtSeq + i] : 0;
            opaque[k] = keep_e[k] + keep_s[k] + keep_z[k] > 0;
        } else {
            const bool deep = D - E[k] > c;
            keep_e[k] = deep ? 0 : E[k];
            keep_s[k] = deep ? 0 : S[k];
            keep_z[k] = 0;
            opaque[k] = !deep && i < n_seq;
        }
        B[k] = D - (E[k] - keep_e[k]);
        a_of[i] = D - E[k] + keep_s[k];
        if (opaque[k]) last_own = i + 1u;
        D += S[k] - E[k];
    }
    u32 any;
    const u32 prev = block_excl_max<kBtT>(last_own, sc, any); // (its barriers publish a_of as well)
    u32 nsyn[PER], synkey[PER], mine = 0;
    i32 a_prev = prev ? a_of[prev - 1u] : 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        nsyn[k] = 0;
        synkey[k] = 0;
        if (opaque[k]) {
            const i32 net = B[k] - a_prev;
            const u32 fpk = g.first_pos(PER * tid + k) << kKeyShift;
            nsyn[k] = (u32)(net < 0 ? -net : net);
            synkey[k] = net > 0 ? fpk - 1u : fpk;
            a_prev = a_of[PER * tid + k];
        }
        mine += (u32)(keep_e[k] + keep_s[k] + keep_z[k]) + nsyn[k];
    }
    u32 m_new;
    u32 base = block_excl_add<kBtT>(mine, sc, m_new);
    const bool fits = m_new <= kBtCap && any != 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = PER * tid + k;
        // ends take [base, base + keep_e), starts the keep_s slots after them
        cur[i] = base;
        lim[i] = fits ? base + (u32)keep_e[k] : base;
        cur[kBtSeq + i] = base + (u32)keep_e[k];
        lim[kBtSeq + i] = fits ? base + (u32)(keep_e[k] + keep_s[k]) : base + (u32)keep_e[k];
        cur[2 * kBtSeq + i] = base + (u32)(keep_e[k] + keep_s[k]);
        lim[2 * kBtSeq + i] = cur[2 * kBtSeq + i] + (fits ? (u32)keep_z[k] : 0u);
        base += (u32)(keep_e[k] + keep_s[k] + keep_z[k]);
        if (fits)
            for (u32 t = 0; t < nsyn[k]; t++) tk[base + t] = synkey[k];
        base += nsyn[k];
    }
    // which (type, bin) keep nothing: one byte per thread and type (its eight bins), read back by
    // every scatter workgroup so that the keys of dropped bins never touch the cursors
    unsigned char *db = reinterpret_cast<unsigned char *>(a.drop + (size_t)si * (3 * kBtSeq / 32));
    u32 de = 0, ds = 0, dz = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        de |= (keep_e[k] == 0 ? 1u : 0u) << k;
        ds |= (keep_s[k] == 0 ? 1u : 0u) << k;
        dz |= (keep_z[k] == 0 ? 1u : 0u) << k;
    }
    db[tid] = (unsigned char)de;
    db[kBtSeq / 8 + tid] = (unsigned char)ds;
    db[2 * kBtSeq / 8 + tid] = (unsigned char)dz;
    if (tid == 0) {
        sg.m_new = m_new;
        if (fits) sg.flags = s.flags | BT_TRIMMED;
    }
}

__global__ __launch_bounds__(kBtT) void bt_scatter_kernel(BtArgs a)
{
    // which cursors this workgroup has seen at their limit (they stay there): a pile of 10^5 equal keys
    // then costs one look at global memory per workgroup, not one per key
    __shared__ u32 done[3 * kBtSeq / 32]; // bit per (type, bin): nothing (more) to take there
    const u32 c = blockIdx.x, si = a.chunk_seg[c];
    const BtSeg s = a.seg[si];
    if (!(s.flags & BT_TRIMMED)) return;
    const BtGeo g = BtGeo::make(s.len);
    u32 *cur = a.cur + (size_t)si * 3 * kBtSeq;
    const u32 *lim = a.lim + (size_t)si * 3 * kBtSeq;
    u32 *tk = a.tkeys + (size_t)si * kBtCap;
    static_assert(PER8 == 8, "the plan kernel writes one byte per thread and type");
    for (u32 b = threadIdx.x; b < 3u * kBtSeq / 32u; b += kBtT) done[b] = a.drop[(size_t)si * (3 * kBtSeq / 32) + b];
    __syncthreads();
    const u32 i0 = (c - s.chunk_off) * kBtChunk, i1 = min(i0 + (u32)kBtChunk, s.n);
    const uint2 *iv = a.iv + s.iv_off;
    auto put = [&](u32 key, u32 at) {
        if ((done[at >> 5] >> (at & 31u)) & 1u) return;
        const u32 l = lim[at];
        if (__hip_atomic_load(&cur[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < l) {
            const u32 p = atomicAdd(&cur[at], 1u);
            if (p < l) tk[p] = key;
        } else {
            atomicOr(&done[at >> 5], 1u << (at & 31u));
        }
    };
    for (u32 i = i0 + threadIdx.x; i < i1; i += 4 * kBtT) {
        uint2 v4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v4[j] = iv[min(i + (u32)j * kBtT, i1 - 1u)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + (u32)j * kBtT >= i1) break;
            u32 ks, ke, b2 = 0, z2 = 0;
            make_event_keys(v4[j], ks, ke, b2, z2);
            const u32 is = g.idx(ks), ie = g.idx(ke);
            const bool zu = z2 != 0 && g.uniform(is); // zero-length at a one-position bin: its own cursor
            put(ks, (zu ? 2u : 1u) * kBtSeq + is);
            put(ke, (zu ? 2u : 0u) * kBtSeq + ie);
        }
    }
}

// one 1024-thread workgroup per read that was thinned: its keys to LDS, sort, sweep
__global__ __launch_bounds__(1024) void bt_sweep_kernel(BtArgs a, u32 *rej_list, u32 *rej_count)
{
    constexpr int T = 1024;
    __shared__ u32 keys[2 * kBtCap];
    __shared__ u32 sc[T / 64 + 1];
    const u32 tid = threadIdx.x;
    const BtSeg s = a.seg[blockIdx.x];
    if (!(s.flags & BT_TRIMMED)) return;
    LaneConst lc;
#pragma unroll
    for (int i = 0; i < 6; i++) lc.k[i] = (tid & (1u << i)) ? 0xFFFFFFFFu : 0u;
    lc.k[6] = 0;
    lc.addr32 = ((tid & 63u) ^ 32u) << 2;
    const u32 *tk = a.tkeys + (size_t)blockIdx.x * kBtCap;
    u32 ms = 0;
    for (u32 i = tid; i < s.m_new; i += T) {
        const u32 k = tk[i];
        keys[i] = k;
        ms = (k & 1u) ? max(ms, k) : ms;
    }
    const u32 max_start = block_max<T>(ms, sc); // (ends with a barrier: the keys are in place)
    uint2 *slot = a.stage + (s.iv_off + 2 * (u64)s.read);
    lds_sort_and_sweep<T>(keys, sc, s.m_new, s.len, a.cov, max_start, 2u * s.n_zl, lc, slot, a.counts,
                          s.read, rej_list, rej_count);
}

} // namespace yk
