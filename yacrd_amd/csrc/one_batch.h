// one_batch.h — a short batch in ONE launch (round 4 prototype).
//
// What a batch of 10 000 - 400 000 reads pays for on the default path is not bytes but dispatches: plan_kernel (bins the
// reads by size), a host sync on its class counts (or a prediction of them), the fused screen launch over the class
// lists, finish_compact_kernel — three dependent kernels, each with its ramp-up, drain and launch gap: 64-66 us for
// 100 000 reads of which the memory system is busy for ~15.  Here the batch is one grid of one-wavefront workgroups:
//   S  a wavefront takes 4 x ITEMS consecutive reads: their offsets (one load), their classes (a ballot: <= 128 intervals
//      -> 16-lane groups, <= 256 -> 32-lane halves; nothing is binned across the batch, so there is nothing to plan),
//      then the healthy-read screen of sweep_wave.h (screen_reads: the fused launch's code); the verdicts go to counts[] /
//      closed[] with agent-scope stores;
//   A  the reads the screen left are sorted right there, one per turn on 64 lanes (finish_item, finish_compact.h), and
//      their regions and counts said again at agent scope;
//   -  the wavefront ARRIVES at its slab of 128 reads (one returning atomic on the slab's word, which also carries the
//      slab's sorted reads and their intervals); every wavefront but the last to arrive is done;
//   B  the last one takes the slab through the follow-on step: region counts -> exclusive scan (decoupled look-back over
//      the slabs' words, 256 of them per round trip), regions into the CSR, type_of_read; the slab that ends the batch
//      sums the arrival words and sends the counter block home.
// No class lists, no class counts, no prediction, no fences (an agent-scope release / acquire pair writes back / invalidates
// a whole L2 per use: the first version with them took 111 us); what the host waits for is one kernel.
// Measured (configs[1]: 100 000 reads / 10 M intervals; profiles/r04/q_*): S + A 26.6 us under rocprofv3 (the fused screen
// alone: 18), with the arrivals 27.8, whole kernel 41-45; one batch at a time 51-55 us against 64-66 on the default path.
// Timestamps inside the kernel (-DYK_OB_STAMPS) put the time in the screening wavefronts, not in phase B: they live 6-20 us
// (three dependent trips, two screens, a sort in every fifth), 1 563 of them per XCD on 768 slots; a slab's own phase B is
// 3-4 us, and a later slab's look-back simply ends when the screening in front of it does.
// A read of more than 256 intervals (the workgroup / device-wide classes) is not handled here: the kernel raises
// Counters::ob_unsupported and the engine runs the batch through the default path (engine.hip), as it does for a batch in
// which the sort rejected a read (exact path) or the regions outgrew their buffer.
#pragma once
#include "finish_compact.h"

namespace yk {

#ifndef YK_OB_ITEMS
#define YK_OB_ITEMS 2 // reads per lane group and turn of the screen: eight reads per wavefront (half the arrivals of ITEMS = 1: 46 against 56 us)
#endif
#ifndef YK_OB_OCC
#define YK_OB_OCC 6 // wavefronts per SIMD the register budget allows: 80 VGPRs, 8 bytes of scratch (LDS: 5 KB per wavefront)
#endif
constexpr int kObItems = YK_OB_ITEMS;
constexpr int kObReads = 4 * kObItems;           // consecutive reads per wavefront
#ifndef YK_OB_SLAB
#define YK_OB_SLAB 128
#endif
#ifndef YK_OB_NAP
#define YK_OB_NAP 32 // x 64 cycles between two looks at the word a slab's look-back waits for
#endif
#ifndef YK_OB_LOOK
#define YK_OB_LOOK 4 // scan words per lane and round trip of the look-back
#endif
constexpr int kObLook = YK_OB_LOOK;
constexpr u32 kObMaxReads = 400000u;               // batches below this many reads (the 32-bit sum of the deferred reads' intervals; engine.hip bounds the waiters)
constexpr int kObSlab = YK_OB_SLAB;                 // reads per slab (one arrival counter, one scan word, one pass of phase B)
constexpr int kObPer = kObSlab / 64;             // reads per lane in phase B
static_assert(kObSlab % kObReads == 0 && kObSlab / kObReads < 0xFFFF, "arrivals are counted in 16 bits");

struct OneBatchArgs {
    CompactArgs2 c;   // sweep (off / iv / len / cov / prefilter / stage / counts / closed / rej_list / rej_count / ctr), scan_state, outputs
    u64 *slab_ctr;    // [slabs] arrivals | deferred reads << 16 | their intervals << 32, zero at launch
    u32 n_slabs;
    u32 *zero;        // the engine's other control block, zeroed here for the next run (as plan_kernel does)
    u32 zero_words;
};

// The screen's verdicts, written so that a wavefront on ANOTHER XCD (its own L2) reads them inside this launch: stores at
// agent scope (write-through), read back with agent-scope loads.  A deferred read is noted in the wavefront's LDS.
struct VerdictsAcrossXcds {
    const SweepArgs &a;
    u32 *s_def; // [kObReads]
    u32 r0;
    __device__ __forceinline__ void closed(u32 r, u32 ra, u32 rb, u32 len) const
    {
        if (ra != 0 || rb != len) {
            __hip_atomic_store(reinterpret_cast<u64 *>(a.closed + r), (u64)ra | ((u64)rb << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.counts + r, kClosedForm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(a.counts + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.prefilter == 2) atomicAdd(&a.ctr->prefiltered, 1u);
    }
    __device__ __forceinline__ void deferred(u32 r) const { s_def[r - r0] = 1u; }
};

__global__ __launch_bounds__(64, YK_OB_OCC) void one_batch_kernel(OneBatchArgs ob)
{
    static_assert(!YK_HOLE_FORM, "the hole form answers through counts[] with plain stores");
    const CompactArgs2 &c = ob.c;
    const SweepArgs &a = c.sweep;
    Counters *ctr = a.ctr;
    for (u32 i = blockIdx.x * 64u + threadIdx.x; i < ob.zero_words; i += gridDim.x * 64u) ob.zero[i] = 0;
    // Workgroups are dealt out to the 8 XCDs round robin (XCD = blockIdx.x mod 8); as in the fused launch every XCD — its
    // own L2 — takes a contiguous eighth of the batch (neighbouring reads share cache lines).  (The other arrangement —
    // a slab's wavefronts on one XCD and the slabs round the XCDs, so that the batch is finished front to back and a
    // slab's look-back finds its predecessors done — measured slower: 59 against 53.5 us for the batch,
    // profiles/r04/q_one_launch_phases.log; -DYK_OB_ROUND_ROBIN builds it.)
    u32 w = blockIdx.x;
#ifdef YK_OB_ROUND_ROBIN
    {
        constexpr u32 kWavesPerSlab = (u32)(kObSlab / kObReads);
        const u32 x = w & 7u, i = w >> 3;
        w = ((i / kWavesPerSlab) * 8u + x) * kWavesPerSlab + i % kWavesPerSlab;
    }
#else
    {
        const u32 nb = gridDim.x, x = w & 7u, q = nb >> 3, rem = nb & 7u;
        w = x * q + min(x, rem) + (w >> 3);
    }
#endif
    __shared__ u32 s_def[kObReads];
#ifdef YK_OB_STAMPS
    const u64 ob_stamp0 = wall_clock64();
#endif
    const u32 lane = lane_id();
    const u32 r0 = w * (u32)kObReads;
    if (r0 >= c.n_reads) return; // (the grid is padded to whole rounds of eight slabs)

    // ---- S: the wavefront's reads through the screen, by size class
    u32 n = 0, len_l = 0;
    u64 o_l = 0;
    bool in = false, huge = false;
    {
        in = lane < (u32)kObReads && r0 + lane < c.n_reads;
        if (in) {
            struct alignas(8) U64x2 { u64 x, y; }; // (8-byte aligned only: r0 + lane may be odd)
            const U64x2 oo = *reinterpret_cast<const U64x2 *>(a.off + (r0 + lane)); // off[r], off[r + 1]
            len_l = a.len[r0 + lane];
            const u64 nn = oo.y - oo.x;
            huge = nn > 256;
            n = huge ? 0u : (u32)nn;
            o_l = oo.x;
        }
        if (lane < (u32)kObReads) s_def[lane] = 0u;
    }
    const u32 m16 = (u32)__builtin_amdgcn_ballot_w64(in && !huge && n <= 128u);
    u32 m32 = (u32)__builtin_amdgcn_ballot_w64(in && !huge && n > 128u);
    const u32 mh = (u32)__builtin_amdgcn_ballot_w64(huge);
    if (mh) { // (the engine runs the batch again on the default path)
        if (huge) __hip_atomic_store(a.counts + (r0 + lane), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            const u32 t = atomicOr(&ctr->ob_unsupported, 1u);
            asm volatile("" ::"v"(t)); // (returning: performed before this wavefront's arrival below)
        }
    }
    wave_lds_sync();
    const VerdictsAcrossXcds sink{a, s_def, r0};
    // (the screen gets extent and length of its reads from the lanes that classified them: one dependent trip less)
    auto known_from = [&](u32 i, u64 &o, u32 &nn, u32 &ll) {
        o = (u64)(u32)__shfl((int)(u32)o_l, (int)i, 64) | ((u64)(u32)__shfl((int)(u32)(o_l >> 32), (int)i, 64) << 32);
        nn = (u32)__shfl((int)n, (int)i, 64);
        ll = (u32)__shfl((int)len_l, (int)i, 64);
    };
    if (m16) { // reads of up to 128 intervals: 16-lane groups, group g takes the wavefront's g-th (and g + 4-th ...) read
        u32 r[kObItems];
        bool act[kObItems];
        ReadsKnown<kObItems> kn;
#pragma unroll
        for (int t = 0; t < kObItems; t++) {
            const u32 i = (u32)t * 4u + (lane >> 4);
            act[t] = ((m16 >> i) & 1u) != 0;
            r[t] = act[t] ? r0 + i : 0u;
            known_from(i, kn.o[t], kn.n[t], kn.len[t]);
        }
        screen_reads<16, kObItems, false, 1>(a, r, act, sink, &kn);
        wave_lds_sync(); // (the next turn zeroes the table)
    }
    while (m32) { // up to 256: 32-lane halves, two (2 x ITEMS) of them per turn (uniform)
        u32 r[kObItems];
        bool act[kObItems];
        ReadsKnown<kObItems> kn;
#pragma unroll
        for (int t = 0; t < kObItems; t++) {
            const u32 i0 = m32 ? (u32)__builtin_ctz(m32) : 0u;
            const bool v0 = m32 != 0;
            m32 &= m32 - 1u;
            const u32 i1 = m32 ? (u32)__builtin_ctz(m32) : 0u;
            const bool v1 = m32 != 0;
            m32 &= m32 - (m32 ? 1u : 0u);
            act[t] = lane < 32u ? v0 : v1;
            r[t] = act[t] ? r0 + (lane < 32u ? i0 : i1) : 0u;
            known_from(lane < 32u ? i0 : i1, kn.o[t], kn.n[t], kn.len[t]);
        }
        screen_reads<32, kObItems, false, 1>(a, r, act, sink, &kn);
        wave_lds_sync();
    }

    // ---- A: what the screen left, one read per turn on all 64 lanes
    u32 n_def = 0, iv_def = 0;
    {
        u32 todo = (u32)__builtin_amdgcn_ballot_w64(lane < (u32)kObReads && s_def[lane] != 0u);
        while (todo) { // (uniform)
            const u32 i = (u32)__builtin_ctz(todo);
            todo &= todo - 1u;
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
            // The sort answers with plain stores (counts[rr], the stage slot): said again at agent scope by the lane that
            // wrote the count, so that the wavefront that takes the slab through phase B — maybe on another XCD, behind
            // another L2 — reads them inside this launch.  (A release fence here and an acquire fence there do the same
            // by writing back / invalidating a whole L2 each time: 111 us for the batch instead of ..., profiles/r04/p_*.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the sort's stores — other lanes' among them — have reached the L2)
            u32 gr = 0;
            if (lane == 63u) gr = a.counts[rr];
            gr = (u32)__builtin_amdgcn_readlane((int)gr, 63);
            uint2 *slot = a.stage + (o + 2 * (u64)rr);
            for (u32 j = lane; j < gr; j += 64u) { // (one trip for the whole slot, not one per region)
                const uint2 v = slot[j];
                __hip_atomic_store(reinterpret_cast<u64 *>(slot + j), (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 63u) __hip_atomic_store(a.counts + rr, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

#if defined(YK_OB_EXPERIMENT) && YK_OB_EXPERIMENT == 1 // (timing only: phases S and A)
    return;
#endif
    // ---- the wavefront arrives at its slab; the last one to arrive takes the slab through phase B
#ifdef YK_OB_STAMPS // (diagnosis: where the last slabs' time goes; wall clock, 100 MHz)
    const u64 st_start = ob_stamp0, st_sa = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store above has been acknowledged
#ifdef YK_OB_STAMPS
    const u64 st_acked = wall_clock64();
#endif
    const u32 slab = r0 / (u32)kObSlab;
    const u32 slab0 = slab * (u32)kObSlab;
    const u32 slab_reads = min(c.n_reads - slab0, (u32)kObSlab);
    u64 seen = 0;
    if (lane == 0)
        seen = __hip_atomic_fetch_add(&ob.slab_ctr[slab], 1ull | ((u64)n_def << 16) | ((u64)iv_def << 32), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    const u32 seen_lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)seen);
    if ((seen_lo & 0xFFFFu) + 1u != (slab_reads + (u32)kObReads - 1u) / (u32)kObReads) return;
#ifdef YK_OB_STAMPS
    const u64 st_arrived = wall_clock64();
    u32 st_hops = 0;
#endif

#if defined(YK_OB_EXPERIMENT) && YK_OB_EXPERIMENT == 2 // (timing only: S, A and the arrivals)
    return;
#endif
    // ---- B: region counts -> scan (decoupled look-back over the slabs) -> CSR, type_of_read
    // Round trips, not bytes, are what this phase costs (it ends the launch): counts[], closed forms, lengths and offsets of
    // the whole slab go out together; the sorted reads' regions (their first three: nearly all have fewer) are asked for
    // before the look-back and arrive during it.
    auto slot_at = [&](u64 at) { // (written at agent scope by the wavefront that sorted the read)
        const u64 v = __hip_atomic_load(reinterpret_cast<const u64 *>(a.stage + at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((u32)v, (u32)(v >> 32));
    };
    u32 g[kObPer], L[kObPer], excl[kObPer];
    uint2 ab[kObPer];
    u64 so[kObPer];
    u32 cfm = 0; // bit k: this lane's k-th read has a closed form
#pragma unroll
    for (int k = 0; k < kObPer; k++) {
        const u32 i = (u32)k * 64u + lane;
        const bool inb = i < slab_reads;
        const u32 r = inb ? slab0 + i : slab0;
        g[k] = __hip_atomic_load(a.counts + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 v = __hip_atomic_load(reinterpret_cast<const u64 *>(a.closed + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab[k] = make_uint2((u32)v, (u32)(v >> 32)); // (meaningful when counts[] says so)
        L[k] = a.len[r];
        so[k] = a.off[r] + 2 * (u64)r;
        if (!inb) g[k] = 0u;
    }
    u32 tot = 0;
#pragma unroll
    for (int k = 0; k < kObPer; k++) {
        if (g[k] == kClosedForm) {
            cfm |= 1u << k;
            g[k] = (ab[k].x != 0u ? 1u : 0u) + (ab[k].y != L[k] ? 1u : 0u);
        }
        const u32 incl = wave_incl_add(g[k]);
        excl[k] = tot + incl - g[k];
        tot += (u32)__shfl((int)incl, 63, 64);
    }
    if (slab > 0 && lane == 0) __hip_atomic_store(&c.scan_state[slab], (1ull << 62) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint2 s0[kObPer], s1[kObPer], s2[kObPer];
#pragma unroll
    for (int k = 0; k < kObPer; k++) {
        const bool sorted = !((cfm >> k) & 1u) && g[k] != 0u;
        s0[k] = s1[k] = s2[k] = make_uint2(0u, 0u);
        if (sorted) s0[k] = slot_at(so[k]);
        if (sorted && g[k] > 1u) s1[k] = slot_at(so[k] + 1);
        if (sorted && g[k] > 2u) s2[k] = slot_at(so[k] + 2);
    }
    constexpr u64 kPre = 2ull << 62, kVal = (1ull << 62) - 1;
#ifdef YK_OB_STAMPS
    u64 st_loaded = wall_clock64();
#endif
    u64 base = 0;
    if (slab > 0) {
        u32 polls = 0;
#ifdef YK_OB_STAMPS
        st_loaded = wall_clock64();
#endif
        for (i32 hi = (i32)slab - 1;; hi -= 64 * kObLook) {
            // 64 x kObLook predecessors per round trip: lane l looks at hi - l, hi - 64 - l, ... (the nearest first)
            u64 part;
            bool found;
            for (;;) { // until the window holds no empty entry before its nearest prefix
                u64 v[kObLook];
#pragma unroll
                for (int j = 0; j < kObLook; j++) {
                    const i32 idx = hi - 64 * j - (i32)lane;
                    v[j] = idx >= 0 ? __hip_atomic_load(&c.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kPre; // before the first slab: prefix 0
                }
                part = 0;
                found = false;
                i32 hole = -1; // the nearest empty word in front of the window's nearest prefix
#pragma unroll
                for (int j = 0; j < kObLook; j++) {
                    if (!found && hole < 0) { // (uniform)
                        const u64 pre = __builtin_amdgcn_ballot_w64((v[j] >> 62) == 2);
                        const u64 before = pre ? ((pre & (0 - pre)) - 1ull) : ~0ull; // lanes nearer than this group's nearest prefix
                        const u64 empty = __builtin_amdgcn_ballot_w64((v[j] >> 62) == 0) & before;
                        if (empty) hole = hi - 64 * j - (i32)__builtin_ctzll(empty);
                        const u32 first_pre = pre ? (u32)__builtin_ctzll(pre) : 64u;
                        part += lane <= first_pre ? (v[j] & kVal) : 0;
                        found = pre != 0;
                    }
                }
                if (hole < 0) break;
                // Wait for THAT word, one load per look, asleep in between: a slab at the front of an XCD's share waits for
                // the slabs of the XCD before it, i.e. for most of the launch, and so do all the slabs behind it — seven
                // eighths of them.  (Looking at the whole window every 64 cycles instead, each of those ~700 wavefronts kept
                // 256 loads of the same 6 KB in flight: the screening wavefronts took 13-20 us each instead of 6-8,
                // profiles/r04/q_one_launch_stamps.log.)
                // Progress: under the contiguous-eighth map the first slabs of XCD x's share have LOW blockIdx and wait for
                // XCD x - 1's last slabs, whose blockIdx is near the grid's end — up to 7/8 of the slab-finishing wavefronts
                // sit here while those are dispatched.  The engine only takes this kernel where that fits (engine.hip:
                // 8 XCDs, twice as many resident wave slots as such waiters); the bound (~20 ms) is there so that anything
                // else — a CU mask, a device shared with another process — shows as a batch sent down the default path,
                // not as a hung device.
                for (;;) {
                    u64 w = 0;
                    if (lane == 0) w = __hip_atomic_load(&c.scan_state[hole], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_readfirstlane((int)(u32)(w >> 62)) != 0) break;
                    if (++polls > (1u << 14)) {
                        if (lane == 0 && c.host_ctr) __hip_atomic_store(&c.host_ctr->ob_unsupported, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                    __builtin_amdgcn_s_sleep(YK_OB_NAP);
                }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
            base += part;
#ifdef YK_OB_STAMPS
            st_hops++;
#endif
            if (found) break;
        }
    }
    const bool last_slab = slab + 1u == ob.n_slabs;
#ifdef YK_OB_STAMPS
    const u64 st_looked = wall_clock64();
#endif
    if (lane == 0) {
        __hip_atomic_store(&c.scan_state[slab], kPre | (base + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last_slab) ctr->total_regions = base + tot;
    }
    // The counters go home from the slab that ends the batch: its look-back has seen every slab's aggregate, so every
    // wavefront of the batch has arrived — the arrival words hold the final numbers of sorted reads and their intervals
    // (summed here: no wavefront adds to the shared counters for them), and nobody writes a counter any more.
    if (c.host_ctr && last_slab) {
        u64 def = 0; // reads | intervals << 32
        for (u32 i = lane; i < ob.n_slabs; i += 64u) {
            const u64 v = __hip_atomic_load(&ob.slab_ctr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            def += ((v >> 16) & 0xFFFFull) | ((v >> 32) << 32);
        }
        u32 def_n = (u32)def, def_iv = (u32)(def >> 32);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            def_n += (u32)__shfl_xor((int)def_n, d, 64);
            def_iv += (u32)__shfl_xor((int)def_iv, d, 64);
        }
        const u64 total = base + tot;
        const u32 *src = reinterpret_cast<const u32 *>(ctr);
        u32 *dst = reinterpret_cast<u32 *>(c.host_ctr);
        constexpr u32 kWords = (u32)(sizeof(Counters) / 4);
        for (u32 i = lane; i < kWords; i += 64u) {
            u32 v = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i == (u32)(offsetof(Counters, total_regions) / 4)) v = (u32)total;
            if (i == (u32)(offsetof(Counters, total_regions) / 4) + 1u) v = (u32)(total >> 32);
            if (i == (u32)(offsetof(Counters, region_overflow) / 4)) v = total > c.region_cap ? 1u : 0u;
            if (i == (u32)(offsetof(Counters, scan_ticket) / 4)) v = ob.n_slabs; // (the host's sign that this copy happened)
            if (i == (u32)(offsetof(Counters, deferred) / 4)) v = def_n;
            if (i == (u32)(offsetof(Counters, deferred_iv) / 4)) v = def_iv;
            if (i == (u32)(offsetof(Counters, deferred_iv) / 4) + 1u) v = 0u; // (< 2^32: fewer than 400 000 reads of at most 256 intervals)
            if (i == (u32)(offsetof(Counters, ob_unsupported) / 4)) {
                if (v) __hip_atomic_store(&dst[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (2 = a look-back gave up: written there directly)
                continue;
            }
            dst[i] = v;
        }
    }
    // offsets, regions, types
#pragma unroll
    for (int k = 0; k < kObPer; k++) {
        const u32 i = (u32)k * 64u + lane;
        if (i < slab_reads) {
            const u32 r = slab0 + i, Lr = L[k], gk = g[k];
            const u64 dst = base + excl[k];
            c.bad_offsets[r] = dst;
            if (r == c.n_reads - 1) c.bad_offsets[c.n_reads] = dst + gk;
            u32 bad = 0;
            bool middle = false;
            const bool fits = dst + gk <= c.region_cap;
            if ((cfm >> k) & 1u) { // (neither region lies in the middle: the first begins at 0, the second ends at len)
                u32 j = 0;
                if (ab[k].x != 0u && fits) c.bad_regions[dst + j++] = make_uint2(0u, ab[k].x);
                if (ab[k].y != Lr && fits) c.bad_regions[dst + j] = make_uint2(ab[k].y, Lr);
                bad = ab[k].x + (Lr - ab[k].y);
            } else if (gk) {
                if (fits) c.bad_regions[dst] = s0[k];
                bad = s0[k].y - s0[k].x;
                middle = (s0[k].x != 0u) & (s0[k].y != Lr);
                if (gk > 1u) {
                    if (fits) c.bad_regions[dst + 1] = s1[k];
                    bad += s1[k].y - s1[k].x;
                    middle |= (s1[k].x != 0u) & (s1[k].y != Lr);
                }
                if (gk > 2u) {
                    if (fits) c.bad_regions[dst + 2] = s2[k];
                    bad += s2[k].y - s2[k].x;
                    middle |= (s2[k].x != 0u) & (s2[k].y != Lr);
                }
                for (u32 j = 3; j < gk; j++) {
                    const uint2 v = slot_at(so[k] + j);
                    if (fits) c.bad_regions[dst + j] = v;
                    bad += v.y - v.x;
                    middle |= (v.x != 0u) & (v.y != Lr);
                }
            }
            if (!fits) atomicOr(&ctr->region_overflow, 1u);
            c.read_type[r] = (uint8_t)classify(bad, middle, Lr, c.not_cov);
        }
    }
#ifdef YK_OB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st_end = wall_clock64();
    if (lane == 0 && (slab % 49u == 48u || last_slab || slab < 2u))
        printf("slab %u of %u: wave start %llu | S+A %llu | stores acked %llu | arrived %llu | loads+scan %llu | look-back %llu (%u hops) | outputs acked %llu (x10 ns, from the wave's start)\n",
               slab, ob.n_slabs, (unsigned long long)st_start, (unsigned long long)(st_sa - st_start), (unsigned long long)(st_acked - st_start),
               (unsigned long long)(st_arrived - st_start), (unsigned long long)(st_loaded - st_start), (unsigned long long)(st_looked - st_start),
               st_hops, (unsigned long long)(st_end - st_start));
#endif
}

} // namespace yk
