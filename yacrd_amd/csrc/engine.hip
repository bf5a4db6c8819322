// engine.hip — C ABI of include/yacrd_engine.h: buffer management, launch sequence, timing.
//
// Launch sequence per run (one HIP stream per engine):
//   plan (also zeroes the other control block for the next run) -> [sync: class counts, unless predicted]
//   -> sweeps of the non-empty classes
//   -> exact general path (LDS scratch for rejected reads; global scratch for huge reads)
//   -> compact_classify (single-pass scan) -> sync.
//   Rare redo: degenerate reads too large for LDS, or bad_regions too small.
// Several engines may share a device (one host thread each): their batches pipeline, the engines
// take turns with the dominant sweep launch (BigLane).
#include "engine_internal.h"

#include <time.h>

#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "device_common.h"
#include "plan_compact.h"
#include "sweep_big.h"
#include "sweep_big_trim.h"
#include "sweep_general.h"
#include "sweep_lds.h"
#include "sweep_wave.h"
#include "finish_compact.h"
#include "screen_wg.h"
#include "screen_stream.h"
#include "screen_big.h"
#include "one_batch.h"

using namespace yke;

// The deferring build of the fused launch (healthy-read screen) is used from this many intervals in the
// classes R16 + H16 on (YACRD_DEFER_MIN_IV overrides, for A/B), unless the previous batches deferred more
// than a quarter of what they screened: then the sorting build runs for kProbeEvery - 1 batches before the
// screen is tried again.
static uint64_t defer_min_intervals()
{
    static const uint64_t v = [] {
        const char *e = std::getenv("YACRD_DEFER_MIN_IV");
        return e ? std::strtoull(e, nullptr, 10) : 4000000ull;
    }();
    return v;
}
static constexpr uint32_t kProbeEvery = 16;

namespace yke {
std::string &err_slot()
{
    thread_local std::string s;
    return s;
}
} // namespace yke

namespace {

// finish the deferred reads + scan + compact + classify: one kernel (finish_compact.h), then the totals
// come home.  `sa` carries the run's inputs, stage / counts and the small classes' rejection list.
// Batches of this many reads and more take the follow-on step as kernels of its own (finish_compact.h: mark_list_kernel +
// deferred_list_kernel + scan_compact_kernel); YACRD_SPLIT_MIN_READS overrides (A/B: 0 = always, a huge number = never).
static uint64_t split_min_reads()
{
    static const uint64_t v = [] {
        const char *e = std::getenv("YACRD_SPLIT_MIN_READS");
        return e ? std::strtoull(e, nullptr, 10) : (uint64_t)yk::kPlanSmallReads;
    }();
    return v;
}

static uint64_t fused_share_override()
{
    static const uint64_t v = [] {
        const char *e = std::getenv("YACRD_FUSED_SHARE");
        return e ? std::min<uint64_t>(std::strtoull(e, nullptr, 10), 4096) : (uint64_t)0;
    }();
    return v;
}

static uint64_t fused_grid_mult()
{
    static const uint64_t v = [] {
        const char *e = std::getenv("YACRD_TEST_FUSED_GRID_MULT");
        return e ? std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) : (uint64_t)1;
    }();
    return v;
}

int launch_compact(yacrd_engine *e, const yk::SweepArgs &sa, u32 n_reads, double not_cov, bool screened)
{
    const u32 nb = (n_reads + yk::kScanBlock - 1) / yk::kScanBlock;
    yk::Counters *ctr = e->ctrl2[e->ctrl_cur].as<yk::Counters>();
    yk::CompactArgs2 ca;
    ca.sweep = sa;
    ca.sweep.first = 0;
    ca.sweep.list = nullptr;
    ca.sweep.list_n = nullptr;
    ca.sweep.over_list = nullptr;
    ca.sweep.over_count = nullptr;
    ca.sweep.rej_list = e->lists.as<u32>() + (size_t)yk::CLS_COUNT * e->last_list_stride;
    ca.sweep.rej_count = &ctr->rej_small;
    ca.scan_state = reinterpret_cast<u64 *>(ctr + 1);
    ca.n_reads = n_reads;
    ca.not_cov = not_cov;
    ca.bad_offsets = e->bad_offsets.as<u64>();
    ca.bad_regions = e->bad_regions.as<uint2>();
    ca.region_cap = (u64)(e->bad_regions.cap / sizeof(uint2));
    ca.read_type = e->read_type.as<uint8_t>();
    if ((uint64_t)n_reads >= split_min_reads()) {
        ca.host_ctr = e->h_ctr;
        if (screened) {
            // the marks listed, the list dealt out evenly over a grid that is resident as a whole (finish_compact.h)
            const u32 slabs = (n_reads + yk::kMarkReads - 1) / yk::kMarkReads;
            yk::DeferList dl;
            dl.shard_cap = (slabs / yk::kDeferShards + 1u) * (u32)yk::kMarkReads;
            HIP_TRY(e->dlist.reserve((size_t)yk::kDeferShards * dl.shard_cap * sizeof(u32)));
            dl.list = e->dlist.as<u32>();
            dl.count = reinterpret_cast<u32 *>(ca.scan_state + nb);
            if (e->compact_calls++) // (a redo of the follow-on step: the list starts over)
                HIP_TRY(hipMemsetAsync(dl.count, 0, (size_t)yk::kDeferShards * yk::kDeferShardStride * sizeof(u32), e->stream));
            hipLaunchKernelGGL(yk::mark_list_kernel, dim3(slabs), dim3(yk::kMarkThreads), 0, e->stream, ca.sweep.counts, n_reads, dl);
            hipLaunchKernelGGL(yk::deferred_list_kernel, dim3((u32)e->num_cu * (u32)(YK_LIST_OCC * 256 / yk::kListThreads)), dim3(yk::kListThreads), 0,
                               e->stream, ca.sweep, dl);
        }
        // (eight reads per thread: 65.1 us against 55.6 on configs[4], profiles/r05/m_*: half the tickets, but twice the latency chain per thread)
        hipLaunchKernelGGL(yk::scan_compact_kernel<4>, dim3((n_reads + 4 * yk::kScanThreads - 1) / (4 * yk::kScanThreads)), dim3(yk::kScanThreads), 0, e->stream, ca);
        return YACRD_OK;
    }
#ifdef YK_NO_HANDOVER
    ca.host_ctr = nullptr;
    hipLaunchKernelGGL(yk::finish_compact_kernel, dim3(nb), dim3(yk::kScanBlock), 0, e->stream, ca);
    HIP_TRY(hipMemcpyAsync(e->h_ctr, ctr, sizeof(yk::Counters), hipMemcpyDeviceToHost, e->stream));
#else
    // The counter block goes home from inside the kernel (its last slab: finish_compact.h): no copy command —
    // a 4 us dispatch of its own — behind it.  (Rounds 2-3 tried this with a ticket per workgroup to find the
    // last one to FINISH: 1 953 same-address tickets on 2 M reads cost more than the copy.  The slab that ends
    // the batch needs no ticket: its look-back has seen everyone else's aggregate.)
    ca.host_ctr = e->h_ctr;
    hipLaunchKernelGGL(yk::finish_compact_kernel, dim3(nb), dim3(yk::kScanBlock), 0, e->stream, ca);
#endif
    return YACRD_OK;
}

// Sizes / offsets of `count` listed reads, to the host (one sync).
int gather_reads(yacrd_engine *e, const u64 *d_off, const u32 *d_len, const u32 *d_list, u32 count,
                 hipStream_t stream, std::vector<yk::GatherOut> &out)
{
    HIP_TRY(e->gen_sizes.reserve((size_t)count * sizeof(yk::GatherOut)));
    hipLaunchKernelGGL(yk::gather_general_sizes_kernel, dim3((count + 255) / 256), dim3(256), 0,
                       stream, d_off, d_len, d_list, count, e->gen_sizes.as<yk::GatherOut>());
    out.resize(count);
    HIP_TRY(hipMemcpyAsync(out.data(), e->gen_sizes.p, (size_t)count * sizeof(yk::GatherOut),
                           hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (const auto &g : out)
        if (g.n >= 0x7FFFFFFFull) return fail(YACRD_EINVAL, "a read has >= 2^31 - 1 intervals");
    return YACRD_OK;
}

// Global-memory exact path over `count` reads listed at d_list (host knows the count).
int run_general_global(yacrd_engine *e, const u64 *d_off, const uint2 *d_iv, const u32 *d_len,
                       const u32 *d_list, u32 count, u32 cov, hipStream_t stream, u64 *iv_total)
{
    std::vector<yk::GatherOut> info;
    int rc = gather_reads(e, d_off, d_len, d_list, count, stream, info);
    if (rc) return rc;
    HIP_TRY(e->gen_scratch_off.reserve((size_t)count * sizeof(u64)));
    std::vector<u64> offs(count);
    u64 tot = 0, gen_iv = 0;
    for (u32 i = 0; i < count; i++) {
        offs[i] = tot;
        tot += 3 * info[i].n + 2;
        gen_iv += info[i].n;
    }
    if (iv_total) *iv_total = gen_iv;
    HIP_TRY(e->gen_scratch.reserve((size_t)tot * sizeof(u64)));
    HIP_TRY(hipMemcpyAsync(e->gen_scratch_off.p, offs.data(), (size_t)count * sizeof(u64),
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream)); // `offs` must outlive the copy
    yk::GeneralArgs ga;
    ga.off = d_off;
    ga.iv = d_iv;
    ga.len = d_len;
    ga.list = d_list;
    ga.scratch_off = e->gen_scratch_off.as<u64>();
    ga.scratch = e->gen_scratch.as<u64>();
    ga.cov = cov;
    ga.stage = e->stage.as<uint2>();
    ga.counts = e->counts.as<u32>();
    hipLaunchKernelGGL(yk::sweep_general_kernel, dim3(count), dim3(yk::kGenThreads), 0, stream, ga);
    return YACRD_OK;
}

// Reads with more than 16 384 intervals: device-wide segmented sort + chunked sweep
// (sweep_big.h).  Reads that turn out to hold a degenerate interval are redone by the exact
// general kernel afterwards.
// Reads with more than 16 384 intervals that the trimming filter could not thin: device-wide
// segmented sort + chunked sweep (sweep_big.h).  Reads that turn out to hold a degenerate interval
// are appended to `redo` for the exact general kernel.
int run_big_sort(yacrd_engine *e, const uint2 *d_iv, const std::vector<yk::GatherOut> &info, u32 cov,
                 hipStream_t stream, std::vector<u32> &redo)
{
    const u32 count = (u32)info.size();
    std::vector<yk::BigSeg> segs(count);
    std::vector<u32> chunk_seg;
    u64 key_off = 0;
    u32 max_P = 0;
    for (u32 i = 0; i < count; i++) {
        u64 P = yk::kBigC;
        while (P < 2 * info[i].n) P <<= 1;
        if (P > 0x80000000ull) return fail(YACRD_EINVAL, "a read has more than 2^30 intervals");
        yk::BigSeg &sg = segs[i];
        sg.key_off = key_off;
        sg.iv_off = info[i].iv_off;
        sg.P = (u32)P;
        sg.n = (u32)info[i].n;
        sg.len = info[i].len;
        sg.read = info[i].read;
        sg.chunk_off = (u32)chunk_seg.size();
        sg.pad = 0;
        for (u64 c = 0; c < P / yk::kBigC; c++) chunk_seg.push_back(i);
        key_off += P;
        max_P = std::max(max_P, (u32)P);
    }
    const size_t n_chunks = chunk_seg.size();
    if (n_chunks >= 0x7FFFFFFFull / (yk::kBigC / 2)) return fail(YACRD_EINVAL, "big path: too many keys");
    const size_t tab_bytes = (size_t)count * sizeof(yk::BigSeg) + n_chunks * sizeof(u32) +
                             9 * n_chunks * sizeof(u32) + (size_t)count * sizeof(u32) + 64;
    HIP_TRY(e->big_tab.reserve(tab_bytes));
    HIP_TRY(e->big_keys.reserve((size_t)key_off * sizeof(u32)));
    char *base = e->big_tab.as<char>();
    yk::BigArgs a;
    a.seg = reinterpret_cast<yk::BigSeg *>(base);
    u32 *w = reinterpret_cast<u32 *>(base + (size_t)count * sizeof(yk::BigSeg));
    a.chunk_seg = w;
    w += n_chunks;
    u32 **per_chunk[] = {&a.c_delta, &a.c_depth_in, &a.c_mf, &a.c_ml, &a.c_mf_in,
                         &a.c_ml_in, &a.c_cnt, &a.c_pos, &a.c_cand};
    for (u32 **pp : per_chunk) {
        *pp = w;
        w += n_chunks;
    }
    a.seg_bad = w;
    a.keys = e->big_keys.as<u32>();
    a.iv = d_iv;
    a.n_chunks = (u32)n_chunks;
    a.n_segs = count;
    a.cov = cov;
    a.stage = e->stage.as<uint2>();
    a.counts = e->counts.as<u32>();
    HIP_TRY(hipMemcpyAsync((void *)a.seg, segs.data(), (size_t)count * sizeof(yk::BigSeg),
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync((void *)a.chunk_seg, chunk_seg.data(), n_chunks * sizeof(u32),
                           hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(a.seg_bad, 0, (size_t)count * sizeof(u32), stream));
    hipLaunchKernelGGL(yk::big_fill_kernel, dim3((u32)n_chunks), dim3(yk::kBigT), 0, stream, a);
    yk::launch_big(a, max_P, stream);
    std::vector<u32> bad(count);
    HIP_TRY(hipMemcpyAsync(bad.data(), a.seg_bad, (size_t)count * sizeof(u32), hipMemcpyDeviceToHost,
                           stream));
    HIP_TRY(hipStreamSynchronize(stream)); // host tables must outlive the copies
    HIP_TRY(hipGetLastError());
    for (u32 i = 0; i < count; i++)
        if (bad[i]) redo.push_back(info[i].read);
    return YACRD_OK;
}

int run_big(yacrd_engine *e, const u64 *d_off, const uint2 *d_iv, const u32 *d_len,
            const u32 *d_list, u32 count, u32 cov, hipStream_t stream, u64 *iv_total)
{
    std::vector<yk::GatherOut> info;
    int rc = gather_reads(e, d_off, d_len, d_list, count, stream, info);
    if (rc) return rc;
    if (iv_total) {
        *iv_total = 0;
        for (const auto &g : info) *iv_total += g.n;
    }
    std::vector<u32> redo; // reads for the exact general path (an interval the keys cannot express)

    // ---- first attempt: thin the reads with the device-wide pile-trimming filter (sweep_big_trim.h);
    // what fits one workgroup's LDS afterwards is swept there, the rest takes the segmented sort below
    if (!(e->flags & YACRD_F_NO_PREFILTER)) {
        std::vector<yk::BtSeg> bs(count);
        std::vector<u32> chunk_seg;
        for (u32 i = 0; i < count; i++) {
            yk::BtSeg &b = bs[i];
            b.iv_off = info[i].iv_off;
            b.n = (u32)info[i].n;
            b.len = info[i].len;
            b.read = info[i].read;
            b.chunk_off = (u32)chunk_seg.size();
            b.flags = 0;
            b.m_new = 0;
            b.n_zl = 0;
            b.pad = 0;
            for (u64 c = 0; c < (info[i].n + yk::kBtChunk - 1) / yk::kBtChunk; c++) chunk_seg.push_back(i);
        }
        const size_t n_chunks = chunk_seg.size();
        const size_t tab_words = (size_t)count * 3 * yk::kBtSeq;
        HIP_TRY(e->bt_tab.reserve((size_t)count * sizeof(yk::BtSeg) + n_chunks * sizeof(u32) + 64));
        HIP_TRY(e->bt_hist.reserve(tab_words * sizeof(u32)));
        HIP_TRY(e->bt_cur.reserve((2 * tab_words + (size_t)count * (3 * yk::kBtSeq / 32)) * sizeof(u32)));
        HIP_TRY(e->bt_keys.reserve((size_t)count * yk::kBtCap * sizeof(u32)));
        yk::BtArgs ba;
        ba.seg = e->bt_tab.as<yk::BtSeg>();
        ba.chunk_seg = reinterpret_cast<const u32 *>(e->bt_tab.as<char>() + (size_t)count * sizeof(yk::BtSeg));
        ba.iv = d_iv;
        ba.n_chunks = (u32)n_chunks;
        ba.n_segs = count;
        ba.cov = cov;
        ba.hist = e->bt_hist.as<u32>();
        ba.cur = e->bt_cur.as<u32>();
        ba.lim = e->bt_cur.as<u32>() + tab_words;
        ba.drop = e->bt_cur.as<u32>() + 2 * tab_words;
        ba.tkeys = e->bt_keys.as<u32>();
        ba.stage = e->stage.as<uint2>();
        ba.counts = e->counts.as<u32>();
        HIP_TRY(hipMemcpyAsync((void *)ba.seg, bs.data(), (size_t)count * sizeof(yk::BtSeg),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync((void *)ba.chunk_seg, chunk_seg.data(), n_chunks * sizeof(u32),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(ba.hist, 0, tab_words * sizeof(u32), stream));
        yk::Counters *ctr = e->ctrl2[e->ctrl_cur].as<yk::Counters>();
        u32 *rej = e->lists.as<u32>() + (size_t)(yk::CLS_COUNT + 2) * e->last_list_stride;
        hipLaunchKernelGGL(yk::bt_hist_kernel, dim3((u32)n_chunks), dim3(yk::kBtT), 0, stream, ba);
        hipLaunchKernelGGL(yk::bt_plan_kernel, dim3(count), dim3(yk::kBtT), 0, stream, ba);
        hipLaunchKernelGGL(yk::bt_scatter_kernel, dim3((u32)n_chunks), dim3(yk::kBtT), 0, stream, ba);
        hipLaunchKernelGGL(yk::bt_sweep_kernel, dim3(count), dim3(1024), 0, stream, ba, rej, &ctr->rej_big);
        HIP_TRY(hipMemcpyAsync(bs.data(), (void *)ba.seg, (size_t)count * sizeof(yk::BtSeg),
                               hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream)); // host tables must outlive the copies
        HIP_TRY(hipGetLastError());
        std::vector<yk::GatherOut> rest;
        for (u32 i = 0; i < count; i++) {
            if (bs[i].flags & yk::BT_TRIMMED) continue;
            if (bs[i].flags & yk::BT_BAD) redo.push_back(info[i].read);
            else rest.push_back(info[i]);
        }
        info.swap(rest);
        count = (u32)info.size();
    }
    if (count) {
        rc = run_big_sort(e, d_iv, info, cov, stream, redo);
        if (rc) return rc;
    }
    if (!redo.empty()) { // degenerate interval in a huge read: exact path, single workgroup each
        HIP_TRY(e->big_redo.reserve(redo.size() * sizeof(u32)));
        HIP_TRY(hipMemcpyAsync(e->big_redo.p, redo.data(), redo.size() * sizeof(u32),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        rc = run_general_global(e, d_off, d_iv, d_len, e->big_redo.as<u32>(), (u32)redo.size(), cov,
                                stream, nullptr);
        if (rc) return rc;
    }
    return YACRD_OK;
}
// The run's final wait: spinning (hipStreamSynchronize) or, with YACRD_F_BLOCKING_WAIT, polling
// an event and sleeping in between (hipEventSynchronize spins as well, blocking-sync flag or not).
int wait_for_stream(yacrd_engine *e)
{
    if (e->flags & YACRD_F_BLOCKING_WAIT) {
        HIP_TRY(hipEventRecord(e->ev_done, e->stream));
        for (;;) {
            const hipError_t q = hipEventQuery(e->ev_done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_TRY(q);
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
    } else {
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    HIP_TRY(hipGetLastError());
    return YACRD_OK;
}

int conclude_run(yacrd_engine *e, yk::Counters c0, bool predicted, uint64_t n_reads64, uint64_t n_iv,
                 const int *cls_b, const int *cls_e, bool fused_marked, bool screened, float extra_ms);
int finish_pending(yacrd_engine *e);

} // namespace

namespace yke {
// A workgroup of the fused workgroup screen ran out of looks at its queue slot (screen_wg.h: the grid was not resident as a
// whole): whatever the batch's kernels wrote is void — the batch from the start, the workgroup classes down the three-launch chain.
static int rerun_without_fused(yacrd_engine *e, const u64 *d_off, const uint2 *d_iv, const u32 *d_len, uint64_t n_reads64,
                               uint64_t n_iv, uint32_t cov, double not_cov)
{
    e->fused_off = true;
    e->pred_valid = false;
    const int rc = run_on_device(e, d_off, d_iv, d_len, n_reads64, n_iv, cov, not_cov, false);
    e->fused_off = false;
    return rc;
}

int run_on_device(yacrd_engine *e, const u64 *d_off, const uint2 *d_iv, const u32 *d_len,
                  uint64_t n_reads64, uint64_t n_iv, uint32_t cov, double not_cov, bool defer)
{
    if (n_reads64 >= 0xFFFFFFFFull) return fail(YACRD_EINVAL, "n_reads must be < 2^32 - 1");
    const u32 n_reads = (u32)n_reads64;
    if (e->pending.active) return fail(YACRD_EINVAL, "a submitted batch is pending: yacrd_engine_wait first");
    e->has_result = false;
    e->timing = yacrd_timing{};

    HIP_TRY(e->bad_offsets.reserve((n_reads64 + 1) * sizeof(u64)));
    if (n_reads == 0) {
        HIP_TRY(hipMemsetAsync(e->bad_offsets.p, 0, sizeof(u64), e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->last_reads = 0;
        e->last_regions = 0;
        e->has_result = true;
        return YACRD_OK;
    }

    // YACRD_F_ONE_LAUNCH: a short batch as one kernel (one_batch.h) — unless a debug flag pins another path
    // — and only where its look-back's progress argument holds (one_batch.h): the slabs at the front of an XCD's
    // contiguous eighth wait for the XCD before it, so up to 7/8 of the slab-finishing wavefronts sit in wave slots for
    // most of the launch; that needs the 8-XCD part and far more resident slots than waiters.  The cap on the reads is
    // the kernel's own (32-bit deferred-interval sum), not the follow-on step's A/B knob.
    const uint64_t ob_waiters = (n_reads64 + yk::kObSlab - 1) / yk::kObSlab * 7 / 8;
    const bool one_launch = (e->flags & YACRD_F_ONE_LAUNCH) && !e->one_launch_off && n_reads64 < (uint64_t)yk::kObMaxReads &&
                            e->num_xcc == 8 && 2 * ob_waiters <= (uint64_t)e->num_cu * 4 * YK_OB_OCC &&
                            !(e->flags & (YACRD_F_FORCE_GENERAL | YACRD_F_WAVE_ONLY | YACRD_F_FORCE_LDS_SORT | YACRD_F_NO_HALVES |
                                          YACRD_F_NO_PREFILTER | YACRD_F_NO_DEFER | YACRD_F_XLANE_DS | YACRD_F_NO_FUSED_LAUNCH));
    // (scan-state words: one per slab of the follow-on kernel; one_batch_kernel: one per slab + its arrival counters)
    const u32 ob_slabs = (n_reads + yk::kObSlab - 1) / yk::kObSlab;
    const u32 nb = one_launch ? 2 * ob_slabs : (n_reads + yk::kScanBlock - 1) / yk::kScanBlock;
    constexpr int kLists = yk::CLS_COUNT + 9; // class lists + three rejection lists + M2 overflow + what the screens leave of M1 / M2 / BIG + what the one-wavefront screen leaves of M1 / M2
    HIP_TRY(e->lists.reserve((size_t)kLists * n_reads * sizeof(u32)));
    // (long batches: + the shard counters of the follow-on step's list of marked reads, behind the scan words)
    const size_t shard_ctr_bytes = (!one_launch && n_reads64 >= split_min_reads()) ? (size_t)yk::kDeferShards * yk::kDeferShardStride * sizeof(u32) : 0;
    const size_t ctrl_bytes = (sizeof(yk::Counters) + (size_t)nb * sizeof(u64) + shard_ctr_bytes + 255) & ~(size_t)255;
    e->compact_calls = 0;
    e->ctrl_cur ^= 1;
    const int cur = e->ctrl_cur, other = cur ^ 1;
    for (int i = 0; i < 2; i++) {
        const void *before = e->ctrl2[i].p;
        HIP_TRY(e->ctrl2[i].reserve(ctrl_bytes));
        if (e->ctrl2[i].p != before) e->ctrl_clean[i] = 0; // fresh allocation: nothing zeroed yet
    }
    HIP_TRY(e->stage.reserve((size_t)(n_iv + 2 * n_reads64) * sizeof(uint2)));
    HIP_TRY(e->counts.reserve((size_t)n_reads * sizeof(u32)));
    HIP_TRY(e->closed.reserve((size_t)n_reads * sizeof(uint2)));
    HIP_TRY(e->read_type.reserve((size_t)n_reads));
    if (e->bad_regions.cap < (size_t)(4 * n_reads64 + 1024) * sizeof(uint2))
        HIP_TRY(e->bad_regions.reserve((size_t)(4 * n_reads64 + 1024) * sizeof(uint2)));

    // (a read of M1 has more than 512 intervals, one of M2 more than 4 096: that many records at most)
    const u32 mrec_cap1 = (u32)std::min<u64>(n_reads64, n_iv / 513 + 1), mrec_cap2 = (u32)std::min<u64>(n_reads64, n_iv / 4097 + 1);
    HIP_TRY(e->mrec.reserve(((size_t)mrec_cap1 + mrec_cap2) * sizeof(uint4)));
    u32 *lists = e->lists.as<u32>();
    e->last_list_stride = n_reads;
    auto list_of = [&](int i) { return lists + (size_t)i * n_reads; };
    u32 *rej_small = list_of(yk::CLS_COUNT), *rej_med = list_of(yk::CLS_COUNT + 1),
        *rej_big = list_of(yk::CLS_COUNT + 2), *over_med = list_of(yk::CLS_COUNT + 3);
    u32 *const fb_med[2] = {list_of(yk::CLS_COUNT + 4), list_of(yk::CLS_COUNT + 5)};
    u32 *const fb_big = list_of(yk::CLS_COUNT + 6);
    u32 *const fb_stream[2] = {list_of(yk::CLS_COUNT + 7), list_of(yk::CLS_COUNT + 8)};
    yk::Counters *ctr = e->ctrl2[cur].as<yk::Counters>();
    const bool full = (e->flags & YACRD_F_TIMING_FULL) != 0;
    const int xm = (e->flags & YACRD_F_XLANE_DS) ? 1 : 0;

    // ---- plan: bin reads by size class.  Only classes that hold reads are launched (an empty
    // launch costs ~4 us and there are twelve classes), so the host needs the class counts.
    // Normally that is one sync right here.  When the previous run on this engine had the same
    // shape (reads, intervals, flags) its class set is used as a PREDICTION instead: the sweeps
    // are launched without waiting, sized for n_reads, and the prediction is validated against
    // the real counts at the final sync (a class that was not predicted is launched then and
    // the compaction redone).  Streams of similar batches never pay the mid-pipeline sync.
    // The control block (counters + scan state) must start out zero.  The engine alternates
    // between two blocks and the plan kernel of a run zeroes the other one, so in steady state no
    // run starts with a fill (hipMemsetAsync = two fill kernels + their launch gaps).
    if (e->ctrl_clean[cur] < ctrl_bytes) HIP_TRY(hipMemsetAsync(ctr, 0, ctrl_bytes, e->stream));
    e->ctrl_clean[cur] = 0;
    const size_t other_bytes = std::min<size_t>(e->ctrl2[other].cap, (size_t)1 << 30) & ~(size_t)3;
    if (full) HIP_TRY(hipEventRecord(e->ev[EV_START], e->stream));
    if (one_launch) {
        yk::OneBatchArgs oa;
        yk::SweepArgs &s = oa.c.sweep;
        s.off = d_off, s.iv = d_iv, s.len = d_len, s.list = nullptr, s.list_n = nullptr, s.rec = nullptr, s.first = 0, s.cov = cov;
        s.prefilter = (e->flags & YACRD_F_COUNT_PREFILTERED) ? 2u : 1u;
        s.stage = e->stage.as<uint2>(), s.counts = e->counts.as<u32>(), s.closed = e->closed.as<uint2>();
        s.rej_list = rej_small, s.rej_count = &ctr->rej_small, s.over_list = nullptr, s.over_count = nullptr, s.ctr = ctr;
        oa.c.scan_state = reinterpret_cast<u64 *>(ctr + 1);
        oa.c.n_reads = n_reads;
        oa.c.not_cov = not_cov;
        oa.c.bad_offsets = e->bad_offsets.as<u64>();
        oa.c.bad_regions = e->bad_regions.as<uint2>();
        oa.c.region_cap = (u64)(e->bad_regions.cap / sizeof(uint2));
        oa.c.read_type = e->read_type.as<uint8_t>();
        oa.c.host_ctr = e->h_ctr;
        oa.slab_ctr = oa.c.scan_state + ob_slabs;
        oa.n_slabs = ob_slabs;
        oa.zero = e->ctrl2[other].as<u32>();
        oa.zero_words = (u32)(other_bytes / 4);
        e->h_ctr->ob_unsupported = 0; // (written from the device only when set)
        e->h_ctr->scan_ticket = 0;    // (the slab that ends the batch sends the counters home: nb tickets then)
        // (one wavefront per kObReads reads; slabs go round the XCDs, so the grid is whole rounds of eight slabs)
        hipLaunchKernelGGL(yk::one_batch_kernel, dim3(((ob_slabs + 7) / 8) * 8 * (u32)(yk::kObSlab / yk::kObReads)), dim3(64), 0, e->stream, oa);
        e->ctrl_clean[other] = other_bytes;
        if (full) HIP_TRY(hipEventRecord(e->ev[EV_COMPACT], e->stream));
        Pending &p = e->pending;
        p = Pending{};
        p.active = true, p.one_launch = true;
        p.d_off = d_off, p.d_iv = d_iv, p.d_len = d_len, p.n_reads = n_reads64, p.n_iv = n_iv, p.cov = cov, p.not_cov = not_cov;
        for (int i = 0; i < 12; i++) p.cls_b[i] = p.cls_e[i] = -1;
        return defer ? YACRD_OK : finish_pending(e); // (yacrd_engine_submit_device: the caller waits later)
    }
    {
        const u32 plan_mode = (u32)((e->flags & YACRD_F_FORCE_GENERAL) ? 1
                                    : (e->flags & (YACRD_F_WAVE_ONLY | YACRD_F_FORCE_LDS_SORT)) ? 2
                                    : (e->flags & YACRD_F_NO_HALVES) ? 3 : 0);
        if (n_reads < yk::kPlanSmallReads)
            hipLaunchKernelGGL((yk::plan_kernel<1, yk::kPlanSmallBlock>), dim3((n_reads + yk::kPlanSmallBlock - 1) / yk::kPlanSmallBlock),
                               dim3(yk::kPlanSmallBlock), 0, e->stream, d_off, n_reads, lists, ctr, plan_mode, e->ctrl2[other].as<u32>(), (u32)(other_bytes / 4), e->counts.as<u32>(), e->mrec.as<uint4>(), mrec_cap1, mrec_cap2);
        else // (eight reads per thread — 611 workgroups instead of 1 221 on configs[4] — measured 32.0 us against 29.9: profiles/r05/m_*)
            hipLaunchKernelGGL(yk::plan_kernel<4>, dim3((n_reads + 4 * yk::kPlanBlock - 1) / (4 * yk::kPlanBlock)),
                               dim3(yk::kPlanBlock), 0, e->stream, d_off, n_reads, lists, ctr, plan_mode,
                               e->ctrl2[other].as<u32>(), (u32)(other_bytes / 4), e->counts.as<u32>(), e->mrec.as<uint4>(), mrec_cap1, mrec_cap2);
    }
    e->ctrl_clean[other] = other_bytes;
    if (full) HIP_TRY(hipEventRecord(e->ev[EV_PLAN], e->stream));

    // (round 6: a batch that holds reads beyond the workgroup classes is predicted too — until then it always waited for the
    // plan's counts, ~20 us of configs[3]'s step —: the device-wide screen is launched for the previous run's count and intervals
    // of such reads, which must come out the same, and must have left nothing to the host-driven sort)
    const bool big_predictable = !(e->flags & YACRD_F_NO_PREFILTER) && e->pred.fb_big == 0;
    const bool predicted = e->pred_valid && e->pred_reads == n_reads64 && e->pred_iv == n_iv &&
                           (e->pred.n[yk::CLS_GENERAL] == 0 || big_predictable) &&
                           !(e->flags & (YACRD_F_FORCE_GENERAL | YACRD_F_NO_PREDICTION));
    const u32 pred_big_n = predicted ? e->pred.n[yk::CLS_GENERAL] : 0u;
    const u64 pred_big_iv = predicted ? e->pred.iv[yk::CLS_GENERAL] : 0ull;
    struct LaunchSet {
        u32 n[12];     // reads used to size the grid; 0 = class not launched
        u32 hint[12];  // the class's size as far as the host knows it (its count, or the prediction + a margin): grids of kernels that cover a list of any length
        u32 first[12]; // first list entry the launch covers (remainder launches)
        u64 iv[12];    // intervals (to pick the dominant class)
    } ls{};
    yk::Counters c0{};
    const bool tight_grids = !(e->flags & YACRD_F_FORCE_LDS_SORT);
    if (predicted) {
        // Register-sort classes: one workgroup per 4..16 reads, so the grid is sized for the
        // predicted count plus a margin (idle workgroups are not free: 19 000 of them cost ~3 us);
        // reads beyond the grid are swept by a remainder launch after the validation.  The LDS
        // classes run grid-stride loops and need no margin.
        for (int cls = 0; cls < yk::CLS_GENERAL; cls++) {
            const u64 p = e->pred.n[cls];
            const u64 cap = (cls <= yk::CLS_W16 && tight_grids) ? ((p + p / 8 + 64 + 15) & ~(u64)15)
                                                               : n_reads64;
            ls.n[cls] = p ? (u32)std::min<u64>(cap, (n_reads64 + 15) & ~(u64)15) : 0;
            ls.hint[cls] = p ? (u32)std::min<u64>(p + p / 8 + 64, n_reads64) : 0;
            ls.iv[cls] = e->pred.iv[cls];
        }
    } else {
        HIP_TRY(hipMemcpyAsync(e->h_ctr, ctr, sizeof(yk::Counters), hipMemcpyDeviceToHost,
                               e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        c0 = *e->h_ctr;
        for (int cls = 0; cls < yk::CLS_GENERAL; cls++) {
            ls.n[cls] = c0.n[cls];
            ls.hint[cls] = c0.n[cls];
            ls.iv[cls] = c0.iv[cls];
        }
    }

    yk::SweepArgs sa;
    sa.first = 0;
    sa.off = d_off;
    sa.iv = d_iv;
    sa.len = d_len;
    sa.cov = cov;
    sa.prefilter = (e->flags & YACRD_F_NO_PREFILTER) ? 0u : (e->flags & YACRD_F_COUNT_PREFILTERED) ? 2u : 1u;
    sa.stage = e->stage.as<uint2>();
    sa.counts = e->counts.as<u32>();
    sa.closed = e->closed.as<uint2>();
    sa.ctr = ctr;
    sa.rec = nullptr;
    sa.over_list = over_med;
    sa.over_count = &ctr->over_med;

    // Kernel-level timing.  An event costs ~3 us of stream time, so by default only the class
    // with the most intervals (the dominant kernel) is bracketed; YACRD_F_TIMING_FULL brackets
    // every launched class and the phases.
    int cls_b[12], cls_e[12]; // event indices bracketing each class, -1 = not recorded
    for (int i = 0; i < 12; i++) cls_b[i] = cls_e[i] = -1;
    int n_cls_ev = 0, dom_cls = -1;
    bool timing_on = !(e->flags & YACRD_F_NO_TIMING);
    if ((e->flags & YACRD_F_TIMING_SAMPLED) && !full && (e->run_seq++ % 8u) != 0) timing_on = false;
    auto before_class = [&](int cls) -> hipError_t {
        if (!timing_on || !(full || cls == dom_cls)) return hipSuccess;
        if (n_cls_ev > 0 && full) { // the previous class's end mark is this one's begin mark
            cls_b[cls] = n_cls_ev - 1;
            return hipSuccess;
        }
        cls_b[cls] = n_cls_ev;
        return hipEventRecord(e->ev_cls[n_cls_ev++], e->stream);
    };
    auto mark_class = [&](int cls) -> hipError_t {
        if (!timing_on || !(full || cls == dom_cls)) return hipSuccess;
        cls_e[cls] = n_cls_ev;
        return hipEventRecord(e->ev_cls[n_cls_ev++], e->stream);
    };

    bool fused_marked = false, screened = false;
    struct MissGuard { // (a miss found by THIS call is reported by its conclude_run and forgotten when the call returns)
        yacrd_engine *e;
        bool before;
        ~MissGuard() { e->miss_pending = before; }
    } miss_guard{e, e->miss_pending};
    e->prev_build = e->last_build;
    e->last_build = -1;
    // sweeps of the classes in `set` (what the register sweeps reject is looked at after the final sync)
    auto launch_sweeps = [&](const LaunchSet &set, bool again = false) -> int {
        sa.rej_list = rej_small;
        sa.rej_count = &ctr->rej_small;
        // the row / half-wavefront classes in one launch
        const bool fuse = !(e->flags & (YACRD_F_FORCE_LDS_SORT | YACRD_F_XLANE_DS |
                                        YACRD_F_NO_FUSED_LAUNCH));
        if (fuse) {
            yk::FusedArgs fa;
            fa.base = sa;
            // R16 / H16: the healthy-read screen, the rest left to finish_compact_kernel — unless the last
            // batches failed the screen too often (e->nodefer_left: see conclude_run)
            const u64 fused_iv = set.iv[yk::CLS_R16] + set.iv[yk::CLS_H16];
            const bool defer = sa.prefilter && !(e->flags & YACRD_F_NO_DEFER) &&
                               ((e->flags & YACRD_F_ALWAYS_DEFER) ||
                                (fused_iv >= defer_min_intervals() && e->nodefer_left == 0));
            // two groups of list entries per wavefront in the screened classes when the launch streams from
            // HBM (more than the 256 MiB Infinity Cache holds): twice the loads in flight per wavefront
            // ... unless the last screened batch left more than a tenth of its reads to the sort: then the one-item build WITH
            // the second looks (sliding windows, sweep_wave.h) takes the next kProbeEvery - 1 — dovetail ends spread by
            // hundreds of positions need them (configs[1] at sigma = 100: 79 % decided against 95 %; configs[2] at 300: 79 %
            // against 97 %), reads that do not are a tenth faster without (conclude_run: wide_left)
            const bool wide = defer && !(e->flags & YACRD_F_SCREEN_ITEMS_2) && ((e->flags & YACRD_F_SCREEN_WIDE) || e->wide_left > 0);
            const int items = !defer ? 1
                              : (wide || (e->flags & YACRD_F_SCREEN_ITEMS_1)) ? 1
                              : ((e->flags & YACRD_F_SCREEN_ITEMS_2) || fused_iv >= 40000000ull) ? 2 : 1;
            e->last_items = (uint32_t)items;
            e->last_wide = wide;
            if (!again) e->last_build = !defer ? 0 : wide ? 3 : items;
            fa.base.over_list = nullptr; // (the deferring build marks its reads in counts[])
            fa.base.over_count = nullptr;
            fa.n_entries = 0;
            u32 blocks = 0;
            bool has_dom = false;
            for (int cls = yk::CLS_R2; cls <= yk::CLS_H16; cls++) {
                if (!set.n[cls]) continue;
                const u32 per = yk::sweep_group_reads_per_block(cls, defer ? yk::kDeferWaves : yk::kFusedWaves) *
                                (cls >= yk::CLS_R16 ? (u32)items : 1u);
                blocks += (set.n[cls] + per - 1) / per;
                fa.cls[fa.n_entries] = (u32)cls;
                fa.block_end[fa.n_entries] = blocks;
                fa.list[fa.n_entries] = list_of(cls);
                fa.list_n[fa.n_entries] = &ctr->n[cls];
                fa.first[fa.n_entries] = set.first[cls];
                fa.n_entries++;
                has_dom |= cls == dom_cls;
            }
            if (fa.n_entries) {
                const bool mark = timing_on && (full || has_dom);
                // Engines that share a device (batches pipelined over several engines) may take
                // turns with this launch (YACRD_F_SWEEP_TURNS: a GPU-side wait on the previous engine's
                // end-of-sweep event, taken before the bracket opens, so that the bracket times the
                // kernel alone).  Not the default any more: the hand-over costs ~10 us per batch
                // (configs[1], three engines: 47.0 us per batch with turns, 38.8 without), and a sweep
                // that fills the GPU leaves the next one little room to overlap anyway (its bracket
                // grows by ~1 us: what the roofline then reports is on the safe side).
                BigLane &lane = g_big_lane[e->device & 63];
                std::lock_guard<std::mutex> turn(lane.mu);
                const bool shared = (e->flags & YACRD_F_SWEEP_TURNS) && lane.owner != nullptr && lane.owner != e;
                if (shared) HIP_TRY(hipStreamWaitEvent(e->stream, lane.last, 0));
                // start / stop events attached to the launch itself (hipExtLaunchKernelGGL): the
                // kernel's own dispatch timestamps, no event packets before and after it
                const bool chain = (e->flags & YACRD_F_SWEEP_TURNS) && (shared || lane.n_engines > 1);
                if (defer && items == 2)
                    hipExtLaunchKernelGGL(yk::sweep_small_fused_defer2_kernel, dim3(blocks), dim3(64), 0,
                                          e->stream, mark ? e->ev_cls[22] : (hipEvent_t) nullptr,
                                          (mark || chain) ? e->ev_cls[23] : (hipEvent_t) nullptr, 0, fa);
                else if (defer && wide)
                    hipExtLaunchKernelGGL(yk::sweep_small_fused_defer_wide_kernel, dim3(blocks), dim3(64 * yk::kDeferWaves), 0,
                                          e->stream, mark ? e->ev_cls[22] : (hipEvent_t) nullptr,
                                          (mark || chain) ? e->ev_cls[23] : (hipEvent_t) nullptr, 0, fa);
                else if (defer)
                    hipExtLaunchKernelGGL(yk::sweep_small_fused_defer_kernel, dim3(blocks), dim3(64 * yk::kDeferWaves), 0,
                                          e->stream, mark ? e->ev_cls[22] : (hipEvent_t) nullptr,
                                          (mark || chain) ? e->ev_cls[23] : (hipEvent_t) nullptr, 0, fa);
                else
                    hipExtLaunchKernelGGL(yk::sweep_small_fused_kernel, dim3(blocks), dim3(64 * yk::kFusedWaves), 0,
                                          e->stream, mark ? e->ev_cls[22] : (hipEvent_t) nullptr,
                                          (mark || chain) ? e->ev_cls[23] : (hipEvent_t) nullptr, 0, fa);
                if (mark || chain) {
                    lane.last = e->ev_cls[23];
                    lane.owner = e;
                }
                if (mark) fused_marked = true;
                if (defer && (set.n[yk::CLS_R16] || set.n[yk::CLS_H16])) screened = true;
            }
        }
        for (int cls = yk::CLS_R2; cls <= yk::CLS_W16; cls++) { // register sort per lane group
            if (!set.n[cls] || (fuse && cls <= yk::CLS_H16)) continue;
            HIP_TRY(before_class(cls));
            sa.list = list_of(cls);
            sa.list_n = &ctr->n[cls];
            sa.first = set.first[cls];
            if (e->flags & YACRD_F_FORCE_LDS_SORT) {
                const u32 grid = (u32)std::min<uint64_t>(set.n[cls], (uint64_t)e->num_cu * 32);
                hipLaunchKernelGGL((yk::sweep_lds_kernel<64, (int)yk::kSmallEvents>), dim3(grid),
                                   dim3(64), 0, e->stream, sa);
            } else {
                switch (cls) {
                case yk::CLS_R2: yk::launch_sweep_group<16, 2>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_R4: yk::launch_sweep_group<16, 4>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_R8: yk::launch_sweep_group<16, 8>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_R16: yk::launch_sweep_group<16, 16>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_H16: yk::launch_sweep_group<32, 16>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_W2: yk::launch_sweep_group<64, 2>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_W4: yk::launch_sweep_group<64, 4>(sa, set.n[cls], e->stream, xm); break;
                case yk::CLS_W8: yk::launch_sweep_group<64, 8>(sa, set.n[cls], e->stream, xm); break;
                default: yk::launch_sweep_group<64, 16>(sa, set.n[cls], e->stream, xm); break;
                }
            }
            HIP_TRY(mark_class(cls));
        }
        sa.first = 0;
        if (full && timing_on) HIP_TRY(hipEventRecord(e->ev[EV_SMALL], e->stream));

        // One read per workgroup.  First the healthy-read screen (screen_wg.h: the read in registers, one pass,
        // closed form); what it cannot finish lands in a fallback list for the trimming filter + LDS sort.
        for (int k = 0; k < 2; k++) {
            const int cls = k == 0 ? yk::CLS_MED1 : yk::CLS_MED2;
            if (!set.n[cls]) continue;
            HIP_TRY(before_class(cls));
            sa.list = list_of(cls);
            sa.list_n = &ctr->n[cls];
            if (sa.prefilter && !(e->flags & YACRD_F_NO_FUSED_SCREEN) && !e->fused_off && e->screen_fused_wgs_per_cu > 0) {
                if (again) { // (a second pass over the class: the lists start over)
                    HIP_TRY(hipMemsetAsync(&ctr->fb_stream[k], 0, sizeof(u32), e->stream));
                    HIP_TRY(hipMemsetAsync(&ctr->fb_med[k], 0, sizeof(u32), e->stream));
                }
                // Two launches (round 6, profiles/r06/m_*): the SCREEN alone — screen_wg_kernel: 100 VGPRs, 8 KB of LDS, no scratch,
                // one read per workgroup handed out by the dispatcher: 0.118-0.128 ms on configs[3] — appends what it cannot
                // decide (a fiftieth of the generator's reads) to a list, and screen_wg_fused_kernel takes THAT list: table again,
                // filtered exact sweep, the whole-read sort for what is left.  With the fallback code inside the screening
                // launch the same screens took 0.163 ms before a single fallback read ran (128 VGPRs, 88 bytes of scratch
                // written by every workgroup, 74 KB of LDS), and 0.212 with them.
                // YACRD_F_STREAM_SCREEN (A/B): the first launch is the one-wavefront-per-read screen (screen_stream.h).
                const bool stream_first = (e->flags & YACRD_F_STREAM_SCREEN) != 0;
                {
                    yk::SweepArgs ss = sa;
                    ss.over_list = fb_stream[k];
                    ss.over_count = &ctr->fb_stream[k];
                    const u32 want = std::max<u32>(set.hint[cls], 1u);
                    if (stream_first) {
                        hipLaunchKernelGGL(yk::screen_stream_kernel, dim3((u32)std::min<uint64_t>(want, (uint64_t)e->num_cu * 96)), dim3(64), 0, e->stream, ss);
                    } else {
                        static const uint64_t wgk_grid = [] { const char *ev = std::getenv("YACRD_WGK_GRID"); return ev ? std::strtoull(ev, nullptr, 10) : (uint64_t)0; }(); // (A/B)
                        const u32 gs1 = (u32)std::min<uint64_t>(want, wgk_grid ? wgk_grid : (uint64_t)e->num_cu * 64);
                        ss.rec = e->mrec.as<uint4>() + (k == 0 ? 0u : mrec_cap1);
                        hipLaunchKernelGGL(yk::screen_wg_kernel, dim3(gs1), dim3(yk::kWsT), 0, e->stream, ss);
                    }
                }
                yk::ScreenFusedArgs fa;
                fa.sweep = sa;
                fa.sweep.list = fb_stream[k], fa.sweep.list_n = &ctr->fb_stream[k];
                fa.sweep.over_list = over_med;
                fa.sweep.over_count = &ctr->over_med;
                fa.sweep.rej_list = k == 0 ? rej_med : rej_big;
                fa.sweep.rej_count = k == 0 ? &ctr->rej_med : &ctr->rej_big;
                fa.n_fallback = &ctr->fb_med[k];
                // (its list is what the screen left: one read per workgroup, a stride loop should the list be longer than the grid.
                //  YACRD_TEST_FUSED_GRID_MULT / YACRD_FUSED_SHARE, tests / A/B: a multiple of that grid, reads per workgroup and turn)
                const u64 known = std::max<u32>(set.hint[cls], 1u);
                u64 share = 1;
                if (fused_share_override()) share = std::min<u64>(fused_share_override(), (u64)yk::kFusedShareMax);
                fa.share = (u32)share;
                const u32 gs = (u32)std::min<u64>(std::min<u64>((known + share - 1) / share, (u64)e->num_cu * 4) * fused_grid_mult(), 0x7FFFFFFFull);
                hipLaunchKernelGGL(yk::screen_wg_fused_kernel, dim3(gs), dim3(yk::kWsT), 0, e->stream, fa);
                if (k == 1) { // what does not fit the in-kernel fallback's 16 384 events even filtered (usually nothing)
                    sa.list = over_med;
                    sa.list_n = &ctr->over_med;
                    sa.rej_list = rej_big;
                    sa.rej_count = &ctr->rej_big;
                    sa.over_list = over_med; // (cannot happen: 32 768 events hold every M2 read)
                    sa.over_count = &ctr->over_med;
                    const u32 grid = (u32)std::min<uint64_t>(set.n[cls], (uint64_t)e->num_cu);
                    hipLaunchKernelGGL((yk::sweep_lds_kernel<1024, (int)yk::kMedium2Events>), dim3(grid), dim3(1024), 0,
                                       e->stream, sa);
                }
                HIP_TRY(mark_class(cls));
                continue;
            }
            if (sa.prefilter) {
                if (again) HIP_TRY(hipMemsetAsync(&ctr->fb_med[k], 0, sizeof(u32), e->stream)); // (its entries are done)
                sa.over_list = fb_med[k];
                sa.over_count = &ctr->fb_med[k];
                static const uint64_t wgk_grid = [] { const char *ev = std::getenv("YACRD_WGK_GRID"); return ev ? std::strtoull(ev, nullptr, 10) : (uint64_t)0; }(); // (A/B)
                const u32 gs = (u32)std::min<uint64_t>(set.n[cls], wgk_grid ? wgk_grid : (uint64_t)e->num_cu * 4);
                hipLaunchKernelGGL(yk::screen_wg_kernel, dim3(gs), dim3(yk::kWsT), 0, e->stream, sa);
                sa.list = fb_med[k];
                sa.list_n = &ctr->fb_med[k];
            }
            sa.rej_list = k == 0 ? rej_med : rej_big;
            sa.rej_count = k == 0 ? &ctr->rej_med : &ctr->rej_big;
            sa.over_list = over_med;
            sa.over_count = &ctr->over_med;
            if (k == 0) {
                const u32 grid = (u32)std::min<uint64_t>(set.n[cls], (uint64_t)e->num_cu * 4);
                hipLaunchKernelGGL((yk::sweep_lds_kernel<256, (int)yk::kMedium1Events>), dim3(grid), dim3(256), 0,
                                   e->stream, sa);
            } else {
                if (sa.prefilter) {
                    // first through the 256-thread kernel (five workgroups per CU): a read whose
                    // filtered keys fit its 8192-key array is done there, the others land in over_med
                    const u32 g256 = (u32)std::min<uint64_t>(set.n[cls], (uint64_t)e->num_cu * 5);
                    hipLaunchKernelGGL((yk::sweep_lds_kernel<256, (int)yk::kMedium1Events>), dim3(g256), dim3(256), 0,
                                       e->stream, sa);
                    sa.list = over_med;
                    sa.list_n = &ctr->over_med;
                }
                const u32 grid = (u32)std::min<uint64_t>(set.n[cls], (uint64_t)e->num_cu);
                hipLaunchKernelGGL((yk::sweep_lds_kernel<1024, (int)yk::kMedium2Events>), dim3(grid), dim3(1024), 0,
                                   e->stream, sa);
            }
            HIP_TRY(mark_class(cls));
        }
        if (full && timing_on) HIP_TRY(hipEventRecord(e->ev[EV_MED], e->stream));

        if (set.n[yk::CLS_MED1]) {
            sa.list = rej_med;
            sa.list_n = &ctr->rej_med;
            sa.rej_list = rej_big;
            sa.rej_count = &ctr->rej_big;
            hipLaunchKernelGGL((yk::sweep_general_lds_kernel<1024, 4096>), dim3(e->num_cu),
                               dim3(1024), 0, e->stream, sa);
        }
        return YACRD_OK;
    };
    // Reads beyond the LDS classes (the host knows their number and their intervals: batches that hold such
    // reads are never predicted).  The device-wide healthy-read screen (screen_big.h) first; what it cannot
    // decide lands in fb_big and takes the trimming filter / the segmented sort after the final sync.
    // YACRD_F_FORCE_GENERAL / YACRD_F_NO_PREFILTER: the host-driven paths directly.
    bool big_screened = false;
    auto launch_huge = [&](u32 count, u64 iv_big, u64 *iv_total, hipStream_t hs) -> int {
        if (e->flags & YACRD_F_FORCE_GENERAL)
            return run_general_global(e, d_off, d_iv, d_len, list_of(yk::CLS_GENERAL), count, cov, e->stream, iv_total);
        if (!sa.prefilter)
            return run_big(e, d_off, d_iv, d_len, list_of(yk::CLS_GENERAL), count, cov, e->stream, iv_total);
        if (iv_total) *iv_total = iv_big;
        const u64 max_chunks = iv_big / yk::kBsChunk + count;
        if (max_chunks >= 0x7FFFFFFFull) return fail(YACRD_EINVAL, "too many intervals in reads beyond 16384");
        HIP_TRY(e->bs_seg.reserve((size_t)count * sizeof(yk::BsSeg)));
        HIP_TRY(e->bs_chunk.reserve((size_t)max_chunks * sizeof(u32)));
        HIP_TRY(e->bs_hist.reserve((size_t)count * 2 * yk::kBsBins * sizeof(u32)));
        yk::BsArgs ba;
        ba.off = d_off, ba.iv = d_iv, ba.len = d_len;
        ba.list = list_of(yk::CLS_GENERAL);
        ba.list_n = &ctr->n[yk::CLS_GENERAL];
        ba.seg = e->bs_seg.as<yk::BsSeg>();
        ba.chunk_seg = e->bs_chunk.as<u32>();
        ba.n_chunks = &ctr->bs_chunks;
        ba.hist = e->bs_hist.as<u32>();
        ba.max_chunks = (u32)max_chunks, ba.max_segs = count;
        ba.cov = cov, ba.count_healthy = sa.prefilter == 2 ? 1u : 0u;
        ba.stage = sa.stage, ba.counts = sa.counts;
        ba.fb_list = fb_big, ba.fb_count = &ctr->fb_big;
        ba.ctr = ctr;
        if (big_screened) HIP_TRY(hipMemsetAsync(&ctr->fb_big, 0, sizeof(u32), hs)); // (a second pass: its entries are done)
        HIP_TRY(hipMemsetAsync(ba.hist, 0, (size_t)count * 2 * yk::kBsBins * sizeof(u32), hs));
        hipLaunchKernelGGL(yk::bs_setup_kernel, dim3(1), dim3(1024), 0, hs, ba);
        hipLaunchKernelGGL(yk::bs_minmax_kernel, dim3((u32)max_chunks), dim3(yk::kBsT), 0, hs, ba);
        hipLaunchKernelGGL(yk::bs_hist_kernel, dim3((u32)max_chunks), dim3(yk::kBsT), 0, hs, ba);
        hipLaunchKernelGGL(yk::bs_verdict_kernel, dim3(count), dim3(yk::kBsVT), 0, hs, ba);
        big_screened = true;
        return YACRD_OK;
    };

    for (int cls = 0; cls < yk::CLS_GENERAL; cls++)
        if (ls.n[cls] && (dom_cls < 0 || ls.iv[cls] > ls.iv[dom_cls])) dom_cls = cls;
    if (full) HIP_TRY(hipEventRecord(e->ev[EV_S0], e->stream));
    // The device-wide screen's four short launches (a dozen microseconds each, dependent, a few hundred workgroups) go out on a
    // stream of their own BESIDE the other classes' sweeps (round 6): behind them, as until round 5, they were 45 us of
    // configs[3]'s 0.31 ms with the device mostly idle.  Fork behind the plan, join in front of the follow-on step.
    int rc = YACRD_OK;
    u64 gen_iv = 0;
    const u32 huge_n = predicted ? pred_big_n : c0.n[yk::CLS_GENERAL];
    const u64 huge_iv = predicted ? pred_big_iv : c0.iv[yk::CLS_GENERAL];
    const bool huge_now = huge_n != 0;
    if (huge_now && sa.prefilter && !(e->flags & YACRD_F_FORCE_GENERAL) && e->side == nullptr) {
        if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            e->side = nullptr; // (no side stream: the launches go out behind the other classes', as until round 5)
        }
    }
    const bool huge_beside = huge_now && sa.prefilter && !(e->flags & YACRD_F_FORCE_GENERAL) && e->side != nullptr;
    if (huge_beside) {
        HIP_TRY(hipEventRecord(e->ev_fork, e->stream));
        HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        rc = launch_huge(huge_n, huge_iv, &gen_iv, e->side);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e->ev_join, e->side));
    }
    rc = launch_sweeps(ls);
    if (rc) return rc;
    if (huge_now && !huge_beside) {
        rc = launch_huge(huge_n, huge_iv, &gen_iv, e->stream);
        if (rc) return rc;
    }
    if (huge_beside) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    if (full) HIP_TRY(hipEventRecord(e->ev[EV_GEN], e->stream));

    // ---- follow-on kernel: scan + compact + classify
    rc = launch_compact(e, sa, n_reads, not_cov, screened);
    if (rc) return rc;
    if (full) HIP_TRY(hipEventRecord(e->ev[EV_COMPACT], e->stream));
    if (defer && predicted) { // yacrd_engine_submit_device: the caller waits later (finish_pending)
        Pending &p = e->pending;
        p.active = true;
        p.d_off = d_off;
        p.d_iv = d_iv;
        p.d_len = d_len;
        p.n_reads = n_reads64;
        p.n_iv = n_iv;
        p.cov = cov;
        p.not_cov = not_cov;
        for (int cls = 0; cls < 12; cls++) {
            p.grid_n[cls] = cls < yk::CLS_GENERAL ? ls.n[cls] : 0;
            p.cls_b[cls] = cls_b[cls];
            p.cls_e[cls] = cls_e[cls];
        }
        p.fused_marked = fused_marked;
        p.screened = screened;
        p.big_n = pred_big_n, p.big_iv = pred_big_iv;
        return YACRD_OK;
    }
    // (spinning on the pinned counter block instead of this call was tried: 0.0816 vs 0.078 ms/step)
    rc = wait_for_stream(e);
    if (rc) return rc;
    timing_on = false;
    if (e->h_ctr->fused_gave_up && !e->fused_off) return rerun_without_fused(e, d_off, d_iv, d_len, n_reads64, n_iv, cov, not_cov);

    // ---- rare slow paths: a class the prediction missed; degenerate reads too large for the
    // LDS exact path; region overflow.  Each ends with a redo of the compaction.
    float extra_ms = 0.f;
    bool redo = false;
    if (predicted) {
        c0 = *e->h_ctr; // the plan's real counts
        LaunchSet missing{};
        bool any_missing = false;
        for (int cls = 0; cls < yk::CLS_GENERAL; cls++)
            if (c0.n[cls] > ls.n[cls]) { // class not predicted, or larger than its grid
                missing.first[cls] = (cls <= yk::CLS_W16 && tight_grids) ? ls.n[cls] : 0;
                missing.n[cls] = c0.n[cls] - missing.first[cls];
                missing.hint[cls] = missing.n[cls];
                any_missing = true;
            }
        const bool big_mismatch = c0.n[yk::CLS_GENERAL] != pred_big_n || c0.iv[yk::CLS_GENERAL] != pred_big_iv;
        if (any_missing || big_mismatch) {
            e->miss_pending = true; // (the prediction did not hold: yacrd_timing.prediction_misses)
            if (!redo) HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
            if (any_missing && (rc = launch_sweeps(missing, true))) return rc;
            if (big_mismatch && c0.n[yk::CLS_GENERAL] && (rc = launch_huge(c0.n[yk::CLS_GENERAL], c0.iv[yk::CLS_GENERAL], &gen_iv, e->stream))) return rc;
            // the rejection counters may have grown: bring them home before looking at rej_big
            HIP_TRY(hipMemcpyAsync(e->h_ctr, ctr, sizeof(yk::Counters), hipMemcpyDeviceToHost,
                                   e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            redo = true;
        }
    }
    // Reads a register sweep (or the follow-on kernel's sort) rejected — an interval the keys cannot
    // express: the exact path with its scratch in LDS, then the compaction once more.  Rare, so nothing is
    // launched for them before their number is known.  (A redo may reject more: the loop below looks again.)
    auto exact_small = [&]() {
        sa.list = rej_small;
        sa.list_n = &ctr->rej_small;
        sa.rej_list = rej_med; // cannot happen (n <= 512), kept well defined
        sa.rej_count = &ctr->rej_med;
        // (a single wavefront per read was tried: 23.9 us vs 20.2 us — the stages are latency
        // chains, more threads shorten each one)
        hipLaunchKernelGGL((yk::sweep_general_lds_kernel<256, 512>), dim3(e->num_cu * 2), dim3(256), 0,
                           e->stream, sa);
    };
    u32 rej_small_done = 0;
    if (e->h_ctr->rej_small) {
        if (!redo) HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
        exact_small();
        rej_small_done = e->h_ctr->rej_small;
        redo = true;
    }
    if (big_screened && e->h_ctr->fb_big) { // BIG reads the screen could not decide: trimming filter / segmented sort
        if (!redo) HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
        rc = run_big(e, d_off, d_iv, d_len, fb_big, e->h_ctr->fb_big, cov, e->stream, nullptr);
        if (rc) return rc;
        // (its rejections — a degenerate interval in a huge read — went the exact way inside run_big; the other
        // rejection counters may have grown)
        HIP_TRY(hipMemcpyAsync(e->h_ctr, ctr, sizeof(yk::Counters), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        redo = true;
    }
    const u32 n_rej_big = e->h_ctr->rej_big;
    if (n_rej_big) {
        if (!redo) HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
        rc = run_general_global(e, d_off, d_iv, d_len, rej_big, n_rej_big, cov, e->stream, nullptr);
        if (rc) return rc;
        redo = true;
    }
    for (int attempt = 0; attempt < 4; attempt++) {
        if (!redo && e->h_ctr->rej_small > rej_small_done) { // the redone follow-on kernel rejected reads of its own
            HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
            exact_small();
            rej_small_done = e->h_ctr->rej_small;
            redo = true;
        }
        if (!redo) {
            if (!e->h_ctr->region_overflow) break;
            HIP_TRY(e->bad_regions.reserve((size_t)(e->h_ctr->total_regions + 16) * sizeof(uint2)));
            HIP_TRY(hipEventRecord(e->ev[EV_X0], e->stream));
        }
        redo = false;
        // reset the scan state, the ticket and the overflow flag (keep the class counters)
        HIP_TRY(hipMemsetAsync(&ctr->region_overflow, 0, 3 * sizeof(u32), e->stream));
        HIP_TRY(hipMemsetAsync(ctr + 1, 0, (size_t)nb * sizeof(u64), e->stream));
        rc = launch_compact(e, sa, n_reads, not_cov, screened);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e->ev[EV_X1], e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        extra_ms += ev_ms(e->ev[EV_X0], e->ev[EV_X1]);
    }
    if (e->h_ctr->region_overflow) return fail(YACRD_EINTERNAL, "bad_regions overflow persisted");
    if (e->h_ctr->rej_small > rej_small_done) return fail(YACRD_EINTERNAL, "rejected reads persisted");

    return conclude_run(e, c0, predicted, n_reads64, n_iv, cls_b, cls_e, fused_marked, screened, extra_ms);
}
} // namespace yke

namespace {

// Bookkeeping after a run's last sync: result sizes, the next run's prediction, timing.
int conclude_run(yacrd_engine *e, yk::Counters c0, bool predicted, uint64_t n_reads64, uint64_t n_iv,
                 const int *cls_b, const int *cls_e, bool fused_marked, bool screened, float extra_ms)
{
    const bool full = (e->flags & YACRD_F_TIMING_FULL) != 0;
    const u32 n_reads = (u32)n_reads64;
    const yk::Counters c1 = *e->h_ctr;
    if (predicted) c0 = c1; // class counts are final either way
    e->last_reads = n_reads;
    e->last_regions = c1.total_regions;
    e->has_result = true;
    e->pred = c1;
    e->pred_reads = n_reads64;
    e->pred_iv = n_iv;
    e->pred_valid = true;

    yacrd_timing &t = e->timing;
    // An event costs ~3 us of stream time: by default only the dominant kernel is bracketed
    // (class_ms / fused_ms); phases and the whole-pipeline time need YACRD_F_TIMING_FULL.
    t.plan_ms = t.sweep_small_ms = t.sweep_medium_ms = t.sweep_general_ms = t.compact_ms = 0.f;
    t.total_ms = 0.f;
    if (full) {
        t.plan_ms = ev_ms(e->ev[EV_START], e->ev[EV_PLAN]);
        t.sweep_small_ms = ev_ms(e->ev[EV_S0], e->ev[EV_SMALL]);
        t.sweep_medium_ms = ev_ms(e->ev[EV_SMALL], e->ev[EV_MED]);
        t.sweep_general_ms = ev_ms(e->ev[EV_MED], e->ev[EV_GEN]);
        t.compact_ms = ev_ms(e->ev[EV_GEN], e->ev[EV_COMPACT]);
        t.total_ms = (predicted ? ev_ms(e->ev[EV_START], e->ev[EV_COMPACT])
                                : t.plan_ms + ev_ms(e->ev[EV_S0], e->ev[EV_COMPACT])) + extra_ms;
    }
    t.n_small = 0;
    t.iv_small = 0;
    for (int cls = yk::CLS_R2; cls <= yk::CLS_W16; cls++) {
        t.n_small += c0.n[cls];
        t.iv_small += c0.iv[cls];
    }
    t.n_medium = (uint64_t)c0.n[yk::CLS_MED1] + c0.n[yk::CLS_MED2];
    t.n_general = (uint64_t)c0.n[yk::CLS_GENERAL] + c1.rej_small + c1.rej_med + c1.rej_big;
    t.iv_medium = c0.iv[yk::CLS_MED1] + c0.iv[yk::CLS_MED2];
    t.iv_general = c0.iv[yk::CLS_GENERAL];
    static_assert(yk::CLS_GENERAL == 11, "yacrd_timing.class_* follows the CLS_ order");
    for (int cls = 0; cls <= yk::CLS_GENERAL; cls++) {
        t.class_reads[cls] = c0.n[cls];
        t.class_intervals[cls] = c0.iv[cls];
        t.class_ms[cls] = 0.f;
        if (cls < yk::CLS_GENERAL && cls_b[cls] >= 0 && cls_e[cls] >= 0)
            t.class_ms[cls] = ev_ms(e->ev_cls[cls_b[cls]], e->ev_cls[cls_e[cls]]);
    }
    t.class_ms[yk::CLS_GENERAL] = (full && c0.n[yk::CLS_GENERAL]) ? t.sweep_general_ms : 0.f;
    t.fused_ms = 0.f;
    t.timed_runs = fused_marked ? 1u : 0u;
    for (int cls = 0; cls < yk::CLS_GENERAL; cls++)
        if (cls_b[cls] >= 0 && cls_e[cls] >= 0) t.timed_runs = 1u;
    t.deferred_reads = c1.deferred;
    t.deferred_intervals = c1.deferred_iv;
    // The next batch's build of the fused launch: when this one screened and more than a quarter of what it
    // screened had to be sorted after all, the sorting build takes the next kProbeEvery - 1 batches.
    if (screened) {
        const uint64_t looked_at = (uint64_t)c0.n[yk::CLS_R16] + c0.n[yk::CLS_H16];
        // (a one-look batch that left more than a tenth is followed by the build with the second looks first: only when
        // THAT leaves more than a quarter does the sorting build take over)
        const bool can_widen = !e->last_wide && !(e->flags & YACRD_F_SCREEN_ITEMS_2);
        e->nodefer_left = (!(e->flags & YACRD_F_ALWAYS_DEFER) && !can_widen && 4 * (uint64_t)c1.deferred > looked_at) ? kProbeEvery - 1 : 0;
        if (!e->last_wide) e->wide_left = 10 * (uint64_t)c1.deferred > looked_at ? kProbeEvery - 1 : 0;
        else if (e->wide_left) e->wide_left--;
    } else if (e->nodefer_left) {
        e->nodefer_left--;
    }
    t.screened = screened ? 1u : 0u;
    t.screen_items = screened ? e->last_items : 0u;
    t.screen_wide = (screened && e->last_wide) ? 1u : 0u;
    t.fused_reruns = e->fused_off ? 1u : 0u;
    t.predicted = predicted ? 1u : 0u;
    t.prediction_misses = e->miss_pending ? 1u : 0u;
    t.build_switches = (e->last_build >= 0 && e->prev_build >= 0 && e->last_build != e->prev_build) ? 1u : 0u;
    t.sorting_build = e->last_build == 0 ? 1u : 0u;
    t.fused_reads = t.fused_intervals = 0;
    t.prefiltered_reads = c1.prefiltered;
    if (fused_marked) t.fused_ms = ev_ms(e->ev_cls[22], e->ev_cls[23]);
    if (!(e->flags & (YACRD_F_FORCE_LDS_SORT | YACRD_F_XLANE_DS | YACRD_F_NO_FUSED_LAUNCH))) {
        // what the one launch of the classes R2..H16 holds (whether or not this run timed it)
        for (int cls = yk::CLS_R2; cls <= yk::CLS_H16; cls++) {
            t.fused_reads += c0.n[cls];
            t.fused_intervals += c0.iv[cls];
        }
    }

    yacrd_timing &ts = e->timing_sum;
    const yacrd_timing keep = ts;
    ts = t; // count fields follow the last run
    ts.h2d_ms = keep.h2d_ms;
    ts.d2h_ms = keep.d2h_ms;
    ts.plan_ms = keep.plan_ms + t.plan_ms;
    ts.sweep_small_ms = keep.sweep_small_ms + t.sweep_small_ms;
    ts.sweep_medium_ms = keep.sweep_medium_ms + t.sweep_medium_ms;
    ts.sweep_general_ms = keep.sweep_general_ms + t.sweep_general_ms;
    ts.compact_ms = keep.compact_ms + t.compact_ms;
    ts.total_ms = keep.total_ms + t.total_ms;
    for (int i = 0; i < 12; i++) ts.class_ms[i] = keep.class_ms[i] + t.class_ms[i];
    ts.fused_ms = keep.fused_ms + t.fused_ms;
    ts.screened = keep.screened + t.screened;
    ts.one_launch = keep.one_launch;
    ts.fused_reruns = keep.fused_reruns + t.fused_reruns;
    ts.screen_wide = keep.screen_wide + t.screen_wide;
    ts.predicted = keep.predicted + t.predicted;
    ts.prediction_misses = keep.prediction_misses + t.prediction_misses;
    ts.build_switches = keep.build_switches + t.build_switches;
    ts.sorting_build = keep.sorting_build + t.sorting_build;
    ts.timed_runs = keep.timed_runs + t.timed_runs;
    e->timing_runs++;
    return YACRD_OK;
}

// Second half of yacrd_engine_submit_device: wait, check the prediction the launches were sized
// with, conclude.  Anything the prediction did not cover (a class beyond its grid, reads for the
// device-wide or the exact path, a region overflow) sends the batch through the synchronous,
// unpredicted path once more: rare, and that path has every redo.
int finish_pending(yacrd_engine *e)
{
    Pending &p = e->pending;
    if (!p.active) return YACRD_OK;
    p.active = false;
    int rc = wait_for_stream(e);
    if (rc) return rc;
    if (p.one_launch) {
        // a read beyond 256 intervals, one the sort rejected (the exact path's) or more regions than bad_regions holds:
        // the default path has every redo — the batch takes it from the start
        const yk::Counters c1 = *e->h_ctr;
        const u32 slabs = (u32)((p.n_reads + yk::kObSlab - 1) / yk::kObSlab);
        if (c1.ob_unsupported || c1.rej_small || c1.region_overflow || c1.scan_ticket != slabs) {
            if (c1.region_overflow) HIP_TRY(e->bad_regions.reserve((size_t)(c1.total_regions + 16) * sizeof(uint2)));
            e->one_launch_off = true;
            rc = run_on_device(e, p.d_off, p.d_iv, p.d_len, p.n_reads, p.n_iv, p.cov, p.not_cov);
            e->one_launch_off = false;
            return rc;
        }
        e->last_reads = p.n_reads;
        e->last_regions = c1.total_regions;
        e->has_result = true;
        yacrd_timing &t = e->timing;
        t = yacrd_timing{};
        if (e->flags & YACRD_F_TIMING_FULL) t.total_ms = ev_ms(e->ev[EV_START], e->ev[EV_COMPACT]);
        t.n_small = p.n_reads, t.iv_small = p.n_iv;
        t.deferred_reads = c1.deferred, t.deferred_intervals = c1.deferred_iv, t.prefiltered_reads = c1.prefiltered;
        t.screened = 1u, t.screen_items = (uint32_t)yk::kObItems, t.one_launch = 1u;
        yacrd_timing &ts = e->timing_sum;
        const yacrd_timing keep = ts;
        ts = t;
        ts.h2d_ms = keep.h2d_ms, ts.d2h_ms = keep.d2h_ms;
        ts.plan_ms = keep.plan_ms, ts.sweep_small_ms = keep.sweep_small_ms, ts.sweep_medium_ms = keep.sweep_medium_ms;
        ts.sweep_general_ms = keep.sweep_general_ms, ts.compact_ms = keep.compact_ms, ts.total_ms = keep.total_ms + t.total_ms;
        for (int i = 0; i < 12; i++) ts.class_ms[i] = keep.class_ms[i];
        ts.fused_ms = keep.fused_ms, ts.screened = keep.screened + 1u, ts.timed_runs = keep.timed_runs;
        ts.one_launch = keep.one_launch + 1u;
        ts.screen_wide = keep.screen_wide;
        ts.fused_reruns = keep.fused_reruns, ts.predicted = keep.predicted, ts.prediction_misses = keep.prediction_misses;
        ts.build_switches = keep.build_switches, ts.sorting_build = keep.sorting_build;
        e->timing_runs++;
        return YACRD_OK;
    }
    const yk::Counters c = *e->h_ctr;
    if (c.fused_gave_up && !e->fused_off) return yke::rerun_without_fused(e, p.d_off, p.d_iv, p.d_len, p.n_reads, p.n_iv, p.cov, p.not_cov);
    bool ok = !c.rej_small && c.n[yk::CLS_GENERAL] == p.big_n && c.iv[yk::CLS_GENERAL] == p.big_iv && !c.fb_big && !c.rej_big && !c.region_overflow;
    for (int cls = 0; cls < yk::CLS_GENERAL; cls++) ok = ok && c.n[cls] <= p.grid_n[cls];
    if (!ok) {
        e->pred_valid = false;
        e->miss_pending = true; // (yacrd_timing.prediction_misses: the re-run's conclude_run reports it)
        const int rcm = run_on_device(e, p.d_off, p.d_iv, p.d_len, p.n_reads, p.n_iv, p.cov, p.not_cov);
        e->miss_pending = false;
        return rcm;
    }
    return conclude_run(e, c, true, p.n_reads, p.n_iv, p.cls_b, p.cls_e, p.fused_marked, p.screened, 0.f);
}

} // namespace

namespace yke {
// host -> HBM at PCIe rate.  A pinned source (yacrd_pinned_alloc, hipHostMalloc, hipHostRegister)
// is one direct DMA, asynchronous on the engine's stream.  A pageable source would make the runtime
// stage it through its own pinned buffers from ONE thread (9.6 GB/s measured for configs[1]'s 80 MB):
// here a few threads copy 4 MiB pieces into the engine's pinned bounce buffers and enqueue their
// DMAs, so memcpy and PCIe overlap.  On return the source is no longer needed in either case.
int h2d(yacrd_engine *e, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return YACRD_OK;
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError(); // an unregistered pointer is reported as an error: not one
    if (pinned || bytes < ((size_t)1 << 20)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
        return YACRD_OK;
    }
    constexpr size_t kPiece = yacrd_engine::kBounceBytes;
    for (int b = 0; b < yacrd_engine::kBounce; b++) {
        if (e->bounce[b]) continue;
        HIP_TRY(hipHostMalloc(&e->bounce[b], kPiece));
        HIP_TRY(hipEventCreateWithFlags(&e->bounce_ev[b], hipEventDisableTiming));
    }
    const size_t n_pieces = (bytes + kPiece - 1) / kPiece;
    const int T = (int)std::min<size_t>(yacrd_engine::kBounce / 2, n_pieces);
    std::vector<hipError_t> errs((size_t)T, hipSuccess);
    auto work = [&](int t) {
        if (hipSetDevice(e->device) != hipSuccess) {
            errs[t] = hipErrorInvalidDevice;
            return;
        }
        int turn = 0;
        for (size_t piece = (size_t)t; piece < n_pieces; piece += (size_t)T, turn ^= 1) {
            const int b = 2 * t + turn; // two buffers per thread: one fills while the other flies
            if (e->bounce_busy[b]) {
                const hipError_t w = hipEventSynchronize(e->bounce_ev[b]);
                if (w != hipSuccess) errs[t] = w;
            }
            const size_t at2 = piece * kPiece, n = std::min(kPiece, bytes - at2);
            std::memcpy(e->bounce[b], (const char *)src + at2, n);
            hipError_t c = hipMemcpyAsync((char *)dst + at2, e->bounce[b], n, hipMemcpyHostToDevice, e->stream);
            if (c == hipSuccess) c = hipEventRecord(e->bounce_ev[b], e->stream);
            if (c != hipSuccess) errs[t] = c;
            e->bounce_busy[b] = true;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (hipError_t er : errs) HIP_TRY(er);
    return YACRD_OK;
}

int fetch_result(yacrd_engine *e, yacrd_result *out)
{
    if (!out) return fail(YACRD_EINVAL, "out is null");
    std::memset(out, 0, sizeof(*out));
    if (e->pending.active) return fail(YACRD_EINVAL, "a submitted batch is pending: yacrd_engine_wait first");
    if (!e->has_result) return fail(YACRD_EINVAL, "no result to fetch");
    const uint64_t R = e->last_reads, G = e->last_regions;
    out->bad_offsets = (uint64_t *)std::malloc((size_t)(R + 1) * sizeof(uint64_t));
    out->bad_regions = (uint32_t *)std::malloc((size_t)(2 * G + 2) * sizeof(uint32_t));
    out->read_type = (uint8_t *)std::malloc((size_t)R + 1);
    if (!out->bad_offsets || !out->bad_regions || !out->read_type) {
        yacrd_result_free(out);
        return fail(YACRD_ENOMEM, "host allocation failed");
    }
    // D2H lands in the engine's pinned staging block (direct DMA; into the pageable result arrays
    // the runtime stages at ~10 GB/s) and is copied out from there; results beyond 1 GiB go direct.
    const size_t b_off = (size_t)(R + 1) * sizeof(uint64_t), b_reg = (size_t)G * sizeof(uint2), b_typ = (size_t)R;
    const size_t o_reg = (b_off + 63) & ~(size_t)63, o_typ = (o_reg + b_reg + 63) & ~(size_t)63;
    const size_t need = o_typ + b_typ + 64;
    bool staged = need <= ((size_t)1 << 30);
    if (staged && need > e->h_out_cap) {
        if (e->h_out) (void)hipHostFree(e->h_out);
        e->h_out = nullptr;
        e->h_out_cap = 0;
        const size_t want = need + need / 4;
        if (hipHostMalloc(&e->h_out, want) == hipSuccess) e->h_out_cap = want;
        else {
            (void)hipGetLastError();
            staged = false;
        }
    }
    char *h = (char *)e->h_out;
    void *t_off = staged ? (void *)h : (void *)out->bad_offsets;
    void *t_reg = staged ? (void *)(h + o_reg) : (void *)out->bad_regions;
    void *t_typ = staged ? (void *)(h + o_typ) : (void *)out->read_type;
    HIP_TRY(hipEventRecord(e->ev_d2h0, e->stream));
    HIP_TRY(hipMemcpyAsync(t_off, e->bad_offsets.p, b_off, hipMemcpyDeviceToHost, e->stream));
    if (G) HIP_TRY(hipMemcpyAsync(t_reg, e->bad_regions.p, b_reg, hipMemcpyDeviceToHost, e->stream));
    if (R) HIP_TRY(hipMemcpyAsync(t_typ, e->read_type.p, b_typ, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(e->ev_d2h1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (staged) {
        std::memcpy(out->bad_offsets, t_off, b_off);
        if (G) std::memcpy(out->bad_regions, t_reg, b_reg);
        if (R) std::memcpy(out->read_type, t_typ, b_typ);
    }
    e->timing.d2h_ms = ev_ms(e->ev_d2h0, e->ev_d2h1);
    out->n_reads = R;
    out->n_regions = G;
    return YACRD_OK;
}
} // namespace yke

namespace {
} // namespace

extern "C" {

int yacrd_abi_version(void) { return YACRD_ABI_VERSION; }

uint64_t yacrd_debug_last_counters(const yacrd_engine *e, void *dst, uint64_t bytes)
{
    if (e && dst && e->h_ctr) std::memcpy(dst, e->h_ctr, (size_t)std::min<uint64_t>(bytes, sizeof(yk::Counters)));
    return sizeof(yk::Counters);
}

const char *yacrd_last_error(void) { return yke::err_slot().c_str(); }

int yacrd_engine_create(const yacrd_engine_cfg *cfg, yacrd_engine **out)
{
    if (!out) return fail(YACRD_EINVAL, "out is null");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(YACRD_ENODEV, "no HIP device visible (libyacrd_hip needs an MI355X / gfx950)");
    int dev = cfg ? cfg->device_id : -1;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= count) return fail(YACRD_EINVAL, "device_id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(YACRD_ENODEV, std::string("device is ") + prop.gcnArchName +
                                      ", this library is built for gfx950 only");
    DeviceGuard guard(dev);
    yacrd_engine *e = new (std::nothrow) yacrd_engine();
    if (!e) return fail(YACRD_ENOMEM, "host allocation failed");
    e->device = dev;
    e->flags = cfg ? cfg->flags : 0;
    e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        int xcc = 0;
        if (hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) xcc = 0;
        (void)hipGetLastError();
        e->num_xcc = xcc;
    }
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, yk::screen_wg_fused_kernel, yk::kWsT, 0) == hipSuccess && per_cu > 0)
            e->screen_fused_wgs_per_cu = std::min(per_cu, 4);
        (void)hipGetLastError();
    }
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    // (e->side is made by the first batch that needs it: a stream is a hardware queue's worth of scheduling state, and engines
    //  that pipeline short batches — three per device in bench.py — should not double their number for nothing)
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming);
    for (int i = 0; i < EV_COUNT && err == hipSuccess; i++) err = hipEventCreate(&e->ev[i]);
    for (int i = 0; i < 24 && err == hipSuccess; i++) err = hipEventCreate(&e->ev_cls[i]);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_done, hipEventBlockingSync | hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_h2d0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_h2d1);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_d2h0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_d2h1);
    if (err == hipSuccess) err = hipHostMalloc((void **)&e->h_ctr, sizeof(yk::Counters));
    if (err != hipSuccess) {
        yacrd_engine_destroy(e);
        return fail(YACRD_ENODEV, std::string("engine setup: ") + hipGetErrorString(err));
    }
    {
        BigLane &lane = g_big_lane[dev & 63];
        std::lock_guard<std::mutex> g(lane.mu);
        lane.n_engines++;
        e->in_lane = true;
    }
    *out = e;
    return YACRD_OK;
}

void yacrd_engine_destroy(yacrd_engine *e)
{
    if (!e) return;
    DeviceGuard guard(e->device);
    e->pending.active = false;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->in_lane) {
        BigLane &lane = g_big_lane[e->device & 63];
        std::lock_guard<std::mutex> g(lane.mu);
        lane.n_engines--;
        if (lane.owner == e) { // its event dies with it (its work is done: synchronized above)
            lane.owner = nullptr;
            lane.last = nullptr;
        }
    }
    DevBuf *bufs[] = {&e->in_off, &e->in_iv, &e->in_len, &e->lists, &e->ctrl2[0], &e->ctrl2[1], &e->stage,
                      &e->counts, &e->closed, &e->gen_sizes, &e->gen_scratch_off,
                      &e->gen_scratch, &e->big_tab, &e->big_keys, &e->big_redo, &e->bt_tab, &e->bt_hist, &e->bt_cur, &e->bt_keys, &e->bs_seg, &e->bs_chunk, &e->bs_hist,
                      &e->bad_offsets, &e->bad_regions, &e->read_type, &e->dlist};
    for (DevBuf *b : bufs) b->release();
    if (e->h_ctr) (void)hipHostFree(e->h_ctr);
    if (e->h_out) (void)hipHostFree(e->h_out);
    if (e->paf_arena) (void)hipHostFree(e->paf_arena);
    if (e->paf_scratch && e->paf_scratch_free) e->paf_scratch_free(e->paf_scratch);
    for (int b = 0; b < yacrd_engine::kBounce; b++) {
        if (e->bounce[b]) (void)hipHostFree(e->bounce[b]);
        if (e->bounce_ev[b]) (void)hipEventDestroy(e->bounce_ev[b]);
    }
    for (int i = 0; i < EV_COUNT; i++)
        if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (int i = 0; i < 24; i++)
        if (e->ev_cls[i]) (void)hipEventDestroy(e->ev_cls[i]);
    hipEvent_t extra[] = {e->ev_h2d0, e->ev_h2d1, e->ev_d2h0, e->ev_d2h1, e->ev_done};
    for (hipEvent_t x : extra)
        if (x) (void)hipEventDestroy(x);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int yacrd_engine_run_device(yacrd_engine *e, const void *d_offsets, const void *d_intervals,
                            const void *d_lengths, uint64_t n_reads, uint64_t n_intervals,
                            uint32_t coverage, double not_coverage, yacrd_device_result *out)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (n_reads && (!d_offsets || !d_lengths)) return fail(YACRD_EINVAL, "null device input");
    if (n_intervals && !d_intervals) return fail(YACRD_EINVAL, "null device intervals");
    if (e->host_pending) return fail(YACRD_EINVAL, "a submitted batch is pending: collect it first");
    DeviceGuard guard(e->device);
    int rc = run_on_device(e, (const u64 *)d_offsets, (const uint2 *)d_intervals,
                           (const u32 *)d_lengths, n_reads, n_intervals, coverage, not_coverage);
    if (rc) return rc;
    if (out) {
        out->n_reads = e->last_reads;
        out->n_regions = e->last_regions;
        out->d_bad_offsets = e->bad_offsets.p;
        out->d_bad_regions = e->bad_regions.p;
        out->d_read_type = e->read_type.p;
    }
    return YACRD_OK;
}

int yacrd_engine_submit_device(yacrd_engine *e, const void *d_offsets, const void *d_intervals,
                               const void *d_lengths, uint64_t n_reads, uint64_t n_intervals,
                               uint32_t coverage, double not_coverage)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (n_reads && (!d_offsets || !d_lengths)) return fail(YACRD_EINVAL, "null device input");
    if (n_intervals && !d_intervals) return fail(YACRD_EINVAL, "null device intervals");
    if (e->host_pending) return fail(YACRD_EINVAL, "a submitted batch is pending: collect it first");
    DeviceGuard guard(e->device);
    return run_on_device(e, (const u64 *)d_offsets, (const uint2 *)d_intervals,
                         (const u32 *)d_lengths, n_reads, n_intervals, coverage, not_coverage, true);
}

int yacrd_engine_wait(yacrd_engine *e, yacrd_device_result *out)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    DeviceGuard guard(e->device);
    int rc = finish_pending(e);
    if (rc) return rc;
    if (!e->has_result) return fail(YACRD_EINVAL, "nothing was submitted");
    if (out) {
        out->n_reads = e->last_reads;
        out->n_regions = e->last_regions;
        out->d_bad_offsets = e->bad_offsets.p;
        out->d_bad_regions = e->bad_regions.p;
        out->d_read_type = e->read_type.p;
    }
    return YACRD_OK;
}

int yacrd_engines_run_device_batches(yacrd_engine *const *engines, uint32_t n_engines,
                                     const yacrd_device_batch *batches, uint32_t n_batches,
                                     yacrd_batch_done_fn done, void *user, yacrd_device_result *last)
{
    if (!engines || n_engines == 0 || (!batches && n_batches)) return fail(YACRD_EINVAL, "bad argument");
    for (uint32_t j = 0; j < n_engines; j++)
        if (!engines[j]) return fail(YACRD_EINVAL, "engine is null");
    std::vector<int64_t> inflight(n_engines, -1); // batch in flight on each engine
    yacrd_device_result res{};
    auto finish = [&](uint32_t j) -> int {
        int rc = yacrd_engine_wait(engines[j], &res);
        if (rc) return rc;
        const uint32_t b = (uint32_t)inflight[j];
        inflight[j] = -1;
        if (done && done(user, b, engines[j], &res)) return fail(YACRD_EINVAL, "the batch callback asked to stop");
        return YACRD_OK;
    };
    int rc = YACRD_OK;
    for (uint32_t i = 0; i < n_batches && rc == YACRD_OK; i++) {
        const uint32_t j = i % n_engines;
        if (inflight[j] >= 0) rc = finish(j);
        if (rc) break;
        const yacrd_device_batch &b = batches[i];
        rc = yacrd_engine_submit_device(engines[j], b.d_offsets, b.d_intervals, b.d_lengths, b.n_reads,
                                        b.n_intervals, b.coverage, b.not_coverage);
        if (rc == YACRD_OK) inflight[j] = i;
    }
    // drain in batch order (also after an error: no engine is left with a pending batch)
    for (uint32_t k = 0; k < n_engines; k++) {
        uint32_t j = 0;
        int64_t lo = -1;
        for (uint32_t q = 0; q < n_engines; q++)
            if (inflight[q] >= 0 && (lo < 0 || inflight[q] < lo)) lo = inflight[q], j = q;
        if (lo < 0) break;
        const int rc2 = finish(j);
        if (rc == YACRD_OK) rc = rc2;
        inflight[j] = -1;
    }
    if (rc == YACRD_OK && last) *last = res;
    return rc;
}

// validate a host CSR and enqueue its way to HBM on the engine's stream (shared by run / submit)
static int stage_host_inputs(yacrd_engine *e, const uint64_t *offsets, const uint32_t *intervals,
                             const uint32_t *lengths, uint64_t n_reads, uint64_t *n_iv_out)
{
    if (n_reads && (!offsets || !lengths)) return fail(YACRD_EINVAL, "null input");
    const uint64_t n_iv = n_reads ? offsets[n_reads] : 0;
    if (n_reads && offsets[0] != 0) return fail(YACRD_EINVAL, "offsets[0] must be 0");
    for (uint64_t r = 0; r < n_reads; r++)
        if (offsets[r + 1] < offsets[r]) return fail(YACRD_EINVAL, "offsets must be non-decreasing");
    if (n_iv && !intervals) return fail(YACRD_EINVAL, "null intervals");
    HIP_TRY(e->in_off.reserve((size_t)(n_reads + 1) * sizeof(uint64_t)));
    HIP_TRY(e->in_iv.reserve((size_t)(n_iv + 1) * sizeof(uint2)));
    HIP_TRY(e->in_len.reserve((size_t)(n_reads + 1) * sizeof(uint32_t)));
    HIP_TRY(hipEventRecord(e->ev_h2d0, e->stream));
    int rc = YACRD_OK;
    if (n_reads) {
        rc = h2d(e, e->in_off.p, offsets, (size_t)(n_reads + 1) * sizeof(uint64_t));
        if (!rc) rc = h2d(e, e->in_len.p, lengths, (size_t)n_reads * sizeof(uint32_t));
    }
    if (!rc && n_iv) rc = h2d(e, e->in_iv.p, intervals, (size_t)n_iv * sizeof(uint2));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev_h2d1, e->stream));
    *n_iv_out = n_iv;
    return YACRD_OK;
}

int yacrd_engine_run(yacrd_engine *e, const uint64_t *offsets, const uint32_t *intervals,
                     const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                     double not_coverage, yacrd_result *out)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (!out) return fail(YACRD_EINVAL, "out is null");
    std::memset(out, 0, sizeof(*out));
    if (e->pending.active) return fail(YACRD_EINVAL, "a submitted batch is pending: yacrd_engine_wait first");
    // (a submit that completed synchronously leaves pending.active false: the result buffers still
    // belong to that batch until it is collected)
    if (e->host_pending) return fail(YACRD_EINVAL, "a submitted batch is pending: collect it first");
    DeviceGuard guard(e->device);
    uint64_t n_iv = 0;
    int rc = stage_host_inputs(e, offsets, intervals, lengths, n_reads, &n_iv);
    if (rc) return rc;
    rc = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), n_reads,
                       n_iv, coverage, not_coverage);
    if (rc) return rc;
    e->timing.h2d_ms = ev_ms(e->ev_h2d0, e->ev_h2d1);
    e->timing_sum.h2d_ms += e->timing.h2d_ms;
    rc = fetch_result(e, out);
    if (!rc) e->timing_sum.d2h_ms += e->timing.d2h_ms;
    return rc;
}

int yacrd_engine_submit(yacrd_engine *e, const uint64_t *offsets, const uint32_t *intervals,
                        const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                        double not_coverage)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (e->pending.active || e->host_pending)
        return fail(YACRD_EINVAL, "a submitted batch is pending: collect it first");
    DeviceGuard guard(e->device);
    uint64_t n_iv = 0;
    int rc = stage_host_inputs(e, offsets, intervals, lengths, n_reads, &n_iv);
    if (rc) return rc;
    rc = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), n_reads,
                       n_iv, coverage, not_coverage, true);
    if (rc) return rc;
    e->host_pending = true;
    return YACRD_OK;
}

int yacrd_engine_collect(yacrd_engine *e, yacrd_result *out)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (!out) return fail(YACRD_EINVAL, "out is null");
    std::memset(out, 0, sizeof(*out));
    if (!e->host_pending) return fail(YACRD_EINVAL, "nothing was submitted");
    e->host_pending = false;
    DeviceGuard guard(e->device);
    int rc = finish_pending(e);
    if (rc) return rc;
    e->timing.h2d_ms = ev_ms(e->ev_h2d0, e->ev_h2d1);
    e->timing_sum.h2d_ms += e->timing.h2d_ms;
    rc = fetch_result(e, out);
    if (!rc) e->timing_sum.d2h_ms += e->timing.d2h_ms;
    return rc;
}

void *yacrd_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        yke::err_slot() = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void yacrd_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int yacrd_engine_fetch(yacrd_engine *e, yacrd_result *out)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    DeviceGuard guard(e->device);
    return fetch_result(e, out);
}

void yacrd_result_free(yacrd_result *r)
{
    if (!r) return;
    std::free(r->bad_offsets);
    std::free(r->bad_regions);
    std::free(r->read_type);
    std::memset(r, 0, sizeof(*r));
}

int yacrd_engine_last_timing(const yacrd_engine *e, yacrd_timing *t)
{
    if (!e || !t) return fail(YACRD_EINVAL, "null argument");
    *t = e->timing;
    return YACRD_OK;
}

int yacrd_engine_timing_total(yacrd_engine *e, yacrd_timing *sum, uint64_t *n_runs, int reset)
{
    if (!e) return fail(YACRD_EINVAL, "null argument");
    if (sum) *sum = e->timing_sum;
    if (n_runs) *n_runs = e->timing_runs;
    if (reset) {
        e->timing_sum = yacrd_timing{};
        e->timing_runs = 0;
        e->run_seq = 0; // YACRD_F_TIMING_SAMPLED: the next run is a timed one
    }
    return YACRD_OK;
}

int yacrd_engine_event_overhead(yacrd_engine *e, float *ms)
{
    if (!e || !ms) return fail(YACRD_EINVAL, "null argument");
    DeviceGuard guard(e->device);
    // an empty bracket: what two event records add to the span they enclose
    constexpr int kPairs = 32;
    double sum = 0;
    for (int i = 0; i < kPairs + 4; i++) {
        HIP_TRY(hipEventRecord(e->ev_cls[22], e->stream));
        HIP_TRY(hipEventRecord(e->ev_cls[23], e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (i >= 4) sum += ev_ms(e->ev_cls[22], e->ev_cls[23]); // the first few warm the path up
    }
    *ms = (float)(sum / kPairs);
    return YACRD_OK;
}

int yacrd_partition_reads(const uint64_t *offsets, uint64_t n_reads, uint32_t n_parts,
                          uint64_t *cuts)
{
    if (!cuts || n_parts == 0 || (n_reads && !offsets)) return fail(YACRD_EINVAL, "bad argument");
    // Balance by work ~ intervals + a per-read constant (a read costs a wavefront even when tiny).
    const uint64_t per_read = 8;
    const uint64_t total = (n_reads ? offsets[n_reads] : 0) + per_read * n_reads;
    cuts[0] = 0;
    uint64_t r = 0;
    for (uint32_t p = 1; p < n_parts; p++) {
        const uint64_t target = total / n_parts * p + (total % n_parts) * p / n_parts;
        uint64_t lo = r, hi = n_reads; // first read index whose prefix work >= target
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (offsets[mid] + per_read * mid >= target) hi = mid;
            else lo = mid + 1;
        }
        r = lo;
        cuts[p] = r;
    }
    cuts[n_parts] = n_reads;
    return YACRD_OK;
}

int yacrd_engines_run_partitioned(yacrd_engine *const *engines, uint32_t n_engines,
                                  const uint64_t *offsets, const uint32_t *intervals,
                                  const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                                  double not_coverage, yacrd_result *out)
{
    if (!engines || n_engines == 0 || !out) return fail(YACRD_EINVAL, "bad argument");
    std::memset(out, 0, sizeof(*out));
    for (uint32_t p = 0; p < n_engines; p++)
        if (!engines[p]) return fail(YACRD_EINVAL, "null engine");
    std::vector<uint64_t> cuts(n_engines + 1);
    int rc = yacrd_partition_reads(offsets, n_reads, n_engines, cuts.data());
    if (rc) return rc;

    std::vector<yacrd_result> parts(n_engines);
    std::vector<int> codes(n_engines, YACRD_OK);
    std::vector<std::string> errs(n_engines);
    auto work = [&](uint32_t p) {
        const uint64_t r0 = cuts[p], r1 = cuts[p + 1];
        const uint64_t base = n_reads ? offsets[r0] : 0;
        std::vector<uint64_t> loc(r1 - r0 + 1);
        for (uint64_t r = r0; r <= r1; r++) loc[r - r0] = offsets[r] - base;
        codes[p] = yacrd_engine_run(engines[p], loc.data(), intervals ? intervals + 2 * base : nullptr,
                                    lengths ? lengths + r0 : nullptr, r1 - r0, coverage,
                                    not_coverage, &parts[p]);
        if (codes[p]) errs[p] = yke::err_slot(); // thread-local message of this worker
    };
    {
        std::vector<std::thread> th;
        for (uint32_t p = 1; p < n_engines; p++) th.emplace_back(work, p);
        work(0);
        for (auto &t : th) t.join();
    }
    for (uint32_t p = 0; p < n_engines; p++)
        if (codes[p]) {
            const int code = codes[p];
            const std::string msg = "partition " + std::to_string(p) + ": " + errs[p];
            for (auto &r : parts) yacrd_result_free(&r);
            return fail(code, msg);
        }
    uint64_t G = 0;
    for (auto &r : parts) G += r.n_regions;
    out->bad_offsets = (uint64_t *)std::malloc((size_t)(n_reads + 1) * sizeof(uint64_t));
    out->bad_regions = (uint32_t *)std::malloc((size_t)(2 * G + 2) * sizeof(uint32_t));
    out->read_type = (uint8_t *)std::malloc((size_t)n_reads + 1);
    if (!out->bad_offsets || !out->bad_regions || !out->read_type) {
        for (auto &r : parts) yacrd_result_free(&r);
        yacrd_result_free(out);
        return fail(YACRD_ENOMEM, "host allocation failed");
    }
    uint64_t g0 = 0;
    for (uint32_t p = 0; p < n_engines; p++) {
        const uint64_t r0 = cuts[p], nr = cuts[p + 1] - cuts[p];
        for (uint64_t r = 0; r < nr; r++) out->bad_offsets[r0 + r] = g0 + parts[p].bad_offsets[r];
        if (parts[p].n_regions)
            std::memcpy(out->bad_regions + 2 * g0, parts[p].bad_regions,
                        (size_t)parts[p].n_regions * 2 * sizeof(uint32_t));
        if (nr) std::memcpy(out->read_type + r0, parts[p].read_type, (size_t)nr);
        g0 += parts[p].n_regions;
        yacrd_result_free(&parts[p]);
    }
    out->bad_offsets[n_reads] = G;
    out->n_reads = n_reads;
    out->n_regions = G;
    return YACRD_OK;
}

int yacrd_engine_classify(yacrd_engine *e, const uint64_t *bad_offsets, const uint32_t *bad_regions,
                          const uint32_t *lengths, uint64_t n_reads, double not_coverage,
                          uint8_t *read_type)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (n_reads == 0) return YACRD_OK;
    if (!bad_offsets || !lengths || !read_type) return fail(YACRD_EINVAL, "null argument");
    if (n_reads >= 0xFFFFFFFFull) return fail(YACRD_EINVAL, "n_reads must be < 2^32 - 1");
    const uint64_t G = bad_offsets[n_reads];
    if (G && !bad_regions) return fail(YACRD_EINVAL, "null regions");
    DeviceGuard guard(e->device);
    DevBuf d_off, d_reg, d_len, d_type;
    int rc = YACRD_OK;
    auto body = [&]() -> int {
        HIP_TRY(d_off.reserve((size_t)(n_reads + 1) * sizeof(u64)));
        HIP_TRY(d_reg.reserve((size_t)(G + 1) * sizeof(uint2)));
        HIP_TRY(d_len.reserve((size_t)n_reads * sizeof(u32)));
        HIP_TRY(d_type.reserve((size_t)n_reads));
        HIP_TRY(hipMemcpyAsync(d_off.p, bad_offsets, (size_t)(n_reads + 1) * sizeof(u64),
                               hipMemcpyHostToDevice, e->stream));
        if (G)
            HIP_TRY(hipMemcpyAsync(d_reg.p, bad_regions, (size_t)G * sizeof(uint2),
                                   hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(d_len.p, lengths, (size_t)n_reads * sizeof(u32),
                               hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(yk::classify_csr_kernel, dim3((u32)((n_reads + 255) / 256)), dim3(256), 0,
                           e->stream, d_off.as<u64>(), d_reg.as<uint2>(), d_len.as<u32>(),
                           (u32)n_reads, not_coverage, d_type.as<uint8_t>());
        HIP_TRY(hipMemcpyAsync(read_type, d_type.p, (size_t)n_reads, hipMemcpyDeviceToHost,
                               e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        return YACRD_OK;
    };
    rc = body();
    d_off.release();
    d_reg.release();
    d_len.release();
    d_type.release();
    return rc;
}

} // extern "C"
