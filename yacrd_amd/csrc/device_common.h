// device_common.h — shared device-side types and wave/workgroup primitives (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

namespace yk {

constexpr u32 kNoKey = 0xFFFFFFFFu;

// Two consecutive intervals in ONE 16-byte load.  An interval pair of a read starts at iv + off[r] + 2P: 8-byte aligned,
// not 16 — fine for a global load (dword alignment is all gfx950 asks for), but a plain `*(const uint4 *)` tells the
// compiler 16; this vector type says 8.
// NT: non-temporal (`nt`) — for launches that STREAM their input from HBM (the two-items build of the screen: inputs beyond the
// 256 MiB Infinity Cache): read once, it should not displace anything: -1 .. -5 % on the screens of configs[4] / [2]
// (profiles/r06/a_ab_nt.log).  Not for everything (round 6 had it everywhere first): a batch that fits the Infinity Cache and is
// looked at again — the next launch of the same engine on the same 82 MB, the follow-on's second look at a deferred read —
// comes back from HBM instead: configs[1]'s screen 18 -> 21-22 us, its pipelined batches 24.5 -> 29 us (profiles/r06/d_*, x_*).
typedef u32 u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <bool NT = false>
__device__ __forceinline__ uint4 load_pair(const uint2 *p)
{
    u32x4_a8 q;
    if constexpr (NT) q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a8 *>(p));
    else q = *reinterpret_cast<const u32x4_a8 *>(p);
    return make_uint4(q.x, q.y, q.z, q.w);
}
// off[r] and off[r + 1] in one load (r may be odd: 8-byte aligned)
typedef u64 u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ ulonglong2 load_extent(const u64 *p)
{
    const u64x2_a8 q = *reinterpret_cast<const u64x2_a8 *>(p);
    return make_ulonglong2(q.x, q.y);
}

struct OvlRec { // == yacrd_ovl_rec: one overlap line, both reads (handles) and their intervals
    u32 a, b, sa, ea, sb, eb;
};

// Event keys of the regular sweeps: position << 2 | class, so that sorting the keys reproduces
// the reference's push/pop order (src/stack.rs:66-91) at one position:
//   0 end of a regular interval     (popped first: `head <= interval.0`, stack.rs:72-81)
//   1 start of a zero-length interval (start == end sorts before the regular intervals that
//   2 end   of a zero-length interval  start there, and is popped first at the next step)
//   3 start of a regular interval
// Reads with start > end, a position >= 2^30 - 1, or two zero-length intervals at one position
// > 0 (S,S,E,E would not be S,E,S,E) go to the exact general path instead.  (0,0) intervals are
// inert in the reference (their pop re-assigns last_covered = 0 while it still is 0) and in the
// keys (their flipped end key 0 never beats "none" = 1), duplicates included.
constexpr u32 kKeyShift = 2;
constexpr u32 kMaxKeyPos = 0x3FFFFFFEu;
__device__ __forceinline__ void make_event_keys(uint2 v, u32 &ks, u32 &ke, u32 &bad, u32 &zero_len)
{
    const u32 z = (v.x == v.y) ? 2u : 0u;
    bad |= (v.x > v.y || v.y > kMaxKeyPos) ? 1u : 0u;
    zero_len += z;              // 2 per zero-length interval
    ks = ((v.x << kKeyShift) | 3u) - z; // class 3, or 1 when zero-length
    ke = (v.y << kKeyShift) + z;        // class 0, or 2 when zero-length
}
// Flagged ends are tracked as tk = key ^ 2 (class 0 <-> 2): under max, the FIRST flagged end of
// a position wins (a regular end beats the zero-length interval's own end), because later
// flagged ends at that position leave last_covered unchanged and must not close the run of low
// starts sitting there.  tk = 1 is "none": its true key 1 ^ 2 = 3 is the key of a start at
// position 0, which turns `last_low > true(tk)` into the reference's `first_covered != 0`.
constexpr u32 kNoFlag = 1u;

// Size classes by events per read (m = 2 * intervals).
constexpr u32 kSmallEvents = 1024;    // one read per wavefront
constexpr u32 kMedium1Events = 8192;  // one read per 256-thread workgroup, 32 KiB LDS
constexpr u32 kMedium2Events = 32768; // one read per 1024-thread workgroup, 128 KiB LDS
// Small reads sort in registers.  R<K>: four reads per wavefront, one per 16-lane DPP row, K keys
// per lane (<= 16*K events); W<K>: one read per wavefront, K keys per lane (<= 64*K events).
// H16: two reads per wavefront (32-lane halves), 16 keys per lane (<= 512 events).
enum { CLS_R2 = 0, CLS_R4, CLS_R8, CLS_R16, CLS_H16, CLS_W2, CLS_W4, CLS_W8, CLS_W16, CLS_MED1,
       CLS_MED2, CLS_GENERAL, CLS_COUNT };

// Device-side counters written by the plan kernel and the sweeps.
struct Counters {
    u32 n[16];               // reads per class (plan kernel)
    u64 iv[16];              // intervals per class (plan kernel)
    u32 rej_small;           // reads with a degenerate interval found by a wave sweep (n <= 512)
    u32 rej_med;             // ... by the 256-thread LDS sweep (n <= 4096)
    u32 rej_big;             // ... by the 1024-thread LDS sweep (n <= 16384): global-memory path
    u32 region_overflow;     // compaction ran out of bad_regions capacity
    u32 scan_ticket;         // dynamic workgroup id of the single-pass scan
    u32 ob_unsupported;      // one_batch_kernel met a read it does not handle (> 256 intervals): the batch takes the default path
    u32 prefiltered;         // reads that took the pre-filtered sort (only counted when asked)
    u32 over_med;            // M2 reads whose filtered keys do not fit the 256-thread kernel's LDS
    u32 fb_med[2];           // M1 / M2 reads the workgroup screen (screen_wg.h) left to the trimming filter + sort
    u32 fb_big;              // BIG reads the device-wide screen (screen_big.h) left to sweep_big_trim.h / sweep_big.h
    u32 fbq_head[2];         // screen_wg_fused_kernel's queue of M1 / M2 reads (fb_med[] is its tail): slots claimed
    u32 fbq_done[2];         // (rounds 4-5: workgroups that had finished screening; unused since nobody waits, round 6)
    u32 bs_chunks;           // chunks of the device-wide screen (written by its setup kernel)
    u32 fused_gave_up;       // (rounds 4-5: a workgroup of screen_wg_fused_kernel ran out of looks at its queue slot; never set since round 6)
    u64 total_regions;       // G, written by the last scan workgroup
    u32 fb_stream[2];        // M1 / M2 reads the one-wavefront screen (screen_stream.h) left to screen_wg_fused_kernel
    // reads the screen deferred and finish_compact_kernel sorted, and their intervals: one atomic each per
    // workgroup of 1024 reads.  (Counting where they are found does not work on this 8-XCD part:
    // same-address atomics are performed at the memory side one after the other, ~8 ns each — 1 500 of
    // them held a 10 us kernel for 22 us, 3 300 returning ones a 20 us kernel for 45 us; and on the cache
    // line of n[], which every wavefront of a sweep reads when it starts, they stall those loads as well.)
    alignas(128) u32 deferred;
    u64 deferred_iv;
};

struct SweepArgs {
    const u64 *off;      // [R+1]
    const uint2 *iv;     // [I] (start,end)
    const u32 *len;      // [R]
    const u32 *list;     // read ids of this class
    const u32 *list_n;   // device-side count
    const uint4 *rec;    // screen_wg_kernel: (first interval's index lo, hi, intervals, read) per entry of `list`, or null (round 6: what plan_kernel knows anyway saves the workgroup a dependent round trip)
    u32 first;           // first list entry this launch covers (remainder launches after a short grid)
    u32 cov;
    u32 prefilter;       // 1: drop events in bins deeper than cov before the sort (sweep_wave.h); 2: and count
    uint2 *stage;        // per-read slot of n+2 regions at off[r] + 2r
    u32 *counts;         // [R] regions per read; kDeferredMark / kClosedForm: see below
    uint2 *closed;       // [R] (a, b) of the reads whose counts[] says kClosedForm (what plan_kernel leaves there)
    u32 *rej_list;       // reads this sweep cannot take (degenerate interval): append here
    u32 *rej_count;
    u32 *over_list;      // sweep_lds_kernel: reads with more events than its LDS holds even after the
    u32 *over_count;     // pre-filter: append here (the 1024-thread kernel takes them)
    Counters *ctr;
};
constexpr u32 kDeferredMark = 0xFFFFFFFFu; // in counts[r]: "deferred", not a region count (<= intervals + 2)
// in counts[r]: the read's regions are (0, a) if a != 0 and (b, len) if b != len, with (a, b) = closed[r].  The
// register classes' screen answers this way instead of through the read's stage slot: the slots lie ~1.6 KB apart,
// and two scattered 8-byte stores per read here + two scattered loads in the follow-on kernel cost 0.16 ms of
// a 0.97 ms step on a 2 M-read input whose reads all have their two end regions (bench.py --jitter 30); closed[] is
// written and read in (nearly) read order.
constexpr u32 kClosedForm = 0xFFFFFFFEu;

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }

// ---- wave64 inclusive scans on DPP (row_shr 1/2/4/8, row_bcast 15/31; identity 0): six VALU
// instructions instead of six ds_bpermute round trips --------------------------------------
#define YK_DPP_ZERO(v, ctrl, rm) (u32) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rm, 0xF, true)
__device__ __forceinline__ u32 wave_incl_add(u32 v)
{
    v += YK_DPP_ZERO(v, 0x111, 0xF); // row_shr:1
    v += YK_DPP_ZERO(v, 0x112, 0xF); // row_shr:2
    v += YK_DPP_ZERO(v, 0x114, 0xF); // row_shr:4
    v += YK_DPP_ZERO(v, 0x118, 0xF); // row_shr:8
    v += YK_DPP_ZERO(v, 0x142, 0xA); // row_bcast:15 -> rows 1, 3
    v += YK_DPP_ZERO(v, 0x143, 0xC); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ u32 wave_incl_max(u32 v)
{
    v = max(v, YK_DPP_ZERO(v, 0x111, 0xF));
    v = max(v, YK_DPP_ZERO(v, 0x112, 0xF));
    v = max(v, YK_DPP_ZERO(v, 0x114, 0xF));
    v = max(v, YK_DPP_ZERO(v, 0x118, 0xF));
    v = max(v, YK_DPP_ZERO(v, 0x142, 0xA));
    v = max(v, YK_DPP_ZERO(v, 0x143, 0xC));
    return v;
}
__device__ __forceinline__ u32 wave_shift_up1(u32 v) { return YK_DPP_ZERO(v, 0x138, 0xF); } // wave_shr:1
__device__ __forceinline__ u32 wave_min(u32 v)
{
#pragma unroll
    for (u32 d = 32; d > 0; d >>= 1) v = min(v, (u32)__shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ u32 wave_max(u32 v)
{
#pragma unroll
    for (u32 d = 32; d > 0; d >>= 1) v = max(v, (u32)__shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ u32 wave_or(u32 v)
{
#pragma unroll
    for (u32 d = 32; d > 0; d >>= 1) v |= (u32)__shfl_xor(v, d, 64);
    return v;
}

// ---- workgroup-wide (T threads, T % 64 == 0) exclusive scans through LDS scratch ------------
// `sc` needs T/64 u32.  Both calls end with a barrier so `sc` can be reused immediately.
template <int T>
__device__ __forceinline__ u32 block_excl_add(u32 v, u32 *sc, u32 &total)
{
    u32 incl = wave_incl_add(v);
    if (T == 64) {
        total = __shfl(incl, 63, 64);
        return incl - v;
    }
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 63) sc[wid] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < T / 64; w++) {
        u32 x = sc[w];
        if (w < wid) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}
template <int T>
__device__ __forceinline__ u32 block_excl_max(u32 v, u32 *sc, u32 &total)
{
    u32 incl = wave_incl_max(v);
    const u32 prev = wave_shift_up1(incl); // lane 0 gets 0
    if (T == 64) {
        total = __shfl(incl, 63, 64);
        return prev;
    }
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 63) sc[wid] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < T / 64; w++) {
        u32 x = sc[w];
        if (w < wid) base = max(base, x);
        tot = max(tot, x);
    }
    __syncthreads();
    total = tot;
    return max(base, prev);
}
template <int T>
__device__ __forceinline__ u32 block_min(u32 v, u32 *sc)
{
    v = wave_min(v);
    if (T == 64) return v;
    if (lane_id() == 0) sc[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = kNoKey;
#pragma unroll
    for (u32 w = 0; w < T / 64; w++) r = min(r, sc[w]);
    __syncthreads();
    return r;
}
template <int T>
__device__ __forceinline__ u32 block_or(u32 v, u32 *sc)
{
    v = wave_or(v);
    if (T == 64) return v;
    if (lane_id() == 0) sc[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = 0;
#pragma unroll
    for (u32 w = 0; w < T / 64; w++) r |= sc[w];
    __syncthreads();
    return r;
}
template <int T>
__device__ __forceinline__ u32 block_max(u32 v, u32 *sc)
{
    v = wave_max(v);
    if (T == 64) return v;
    if (lane_id() == 0) sc[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = 0;
#pragma unroll
    for (u32 w = 0; w < T / 64; w++) r = max(r, sc[w]);
    __syncthreads();
    return r;
}

// ---- final assembly shared by every regular-read sweep --------------------------------------
// After the event sweep a read is described by:
//   g      closed regions already written to slot[0..g)
//   mf_t   true key of the last effective flagged end (0 = none);  ml_t  key of the last low start
//   min_ge smallest flagged tail end >= len (kNoKey = none)
// This reproduces reference src/stack.rs:93-113 (tail loop, prepend, append) and the part of
// the equal-begin merge (:119-136) that touches the last region.  Returns the region count.
__device__ __forceinline__ u32 finish_read(uint2 *slot, u32 g, u32 mf_t, u32 ml_t, u32 min_ge,
                                           u32 len)
{
    const u32 lcf = (min_ge != kNoKey) ? min_ge : (mf_t >> kKeyShift);
    if (ml_t > mf_t) { // the last low start comes after the last flagged end: open run
        const u32 b = mf_t >> kKeyShift, e = ml_t >> kKeyShift;
        if (mf_t == 0) { // nothing ever exceeded the coverage threshold
            if (e != 0 && len != 0) slot[g++] = make_uint2(0, max(e, len));
            else if (e != 0) slot[g++] = make_uint2(0, e);
            else if (len != 0) slot[g++] = make_uint2(0, len);
        } else {
            slot[g++] = make_uint2(b, (lcf != len) ? max(e, len) : e);
        }
    } else if (lcf != len) {
        slot[g++] = make_uint2(lcf, len);
    }
    return g;
}

} // namespace yk
