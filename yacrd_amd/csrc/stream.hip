// stream.hip — streaming ingest (include/yacrd_engine.h, yacrd_stream_*): overlap records cross PCIe
// from pinned buffers while the parser is still running, the CSR is built on the GPU (csr_build.h)
// and handed to the engine's launch sequence (engine.hip).
//
// Reference shape: Reads2Ovl::get_overlaps fills a batch, FromOverlap::compute_all_bad_part consumes
// it (src/stack.rs:143-162); FullMemory only has one batch, after the last line
// (src/reads2ovl/fullmemory.rs:46-50).  Here the batch boundary stays where it is — reads are only
// complete when the file ends — but the bytes do not wait for it.
#include "engine_internal.h"

#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "csr_build.h"

using namespace yke;

namespace {

unsigned usable_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long long period = 0;
        if (std::fscanf(f, "%31s %llu", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
            const unsigned long long q = std::strtoull(quota, nullptr, 10);
            if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long long>(1, (q + period - 1) / period));
        }
        std::fclose(f);
    }
    return n;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

enum { BUF_FREE = 0, BUF_HELD = 1, BUF_FLYING = 2 };

} // namespace

struct yacrd_stream {
    yacrd_engine *e = nullptr;
    hipStream_t copy = nullptr;
    uint64_t chunk_records = 0;
    uint32_t n_buffers = 0;
    char *arena = nullptr; // one pinned allocation, n_buffers * chunk bytes
    std::vector<int> state;
    std::vector<hipEvent_t> ev0, ev1; // around each buffer's copy
    std::mutex mu;
    // device side: records land in slabs, back to back
    struct Slab {
        yke::DevBuf buf;
        uint64_t cap = 0, used = 0; // records
    };
    std::vector<Slab> slabs;
    size_t cur_slab = 0;
    uint64_t n_records = 0;
    double busy_ms = 0;
    yke::DevBuf cnt, part, map, err;
    hipEvent_t evb0 = nullptr, evb1 = nullptr;
    yacrd_stream_stats stats{};
};

namespace {

// a buffer's DMA is over: add its duration, mark it free (mutex held)
void retire(yacrd_stream *s, uint32_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev0[b], s->ev1[b]) == hipSuccess) s->busy_ms += ms;
    else (void)hipGetLastError();
    s->state[b] = BUF_FREE;
}

int sink_acquire(void *ctx, yacrd_ovl_rec **buf, uint64_t *cap)
{
    return yacrd_stream_acquire((yacrd_stream *)ctx, buf, cap);
}
int sink_commit(void *ctx, yacrd_ovl_rec *buf, uint64_t n)
{
    return yacrd_stream_commit((yacrd_stream *)ctx, buf, n);
}

} // namespace

namespace yke {
// Overlap records in HBM -> the engine's input CSR (in_off, in_iv; in_len holds the lengths already): count,
// scan, scatter (csr_build.h).  `map` (or null) translates the records' handles to read ids.  Blocking.
int csr_from_records(yacrd_engine *e, const RecSlab *slabs, size_t n_slabs, const u32 *d_map, u64 n_handles, u64 n_reads,
                     DevBuf &cnt, DevBuf &part, DevBuf &err, hipEvent_t done, u64 *n_intervals, bool counted, u64 iv_bound)
{
    u64 n = 0;
    for (size_t i = 0; i < n_slabs; i++) n += slabs[i].n;
    // two intervals per record — unless the caller knows how many of them name THIS engine's reads (iv_bound: the N-engine
    // device parser hands every engine every range's records and the engine keeps 1/N of the halves; sized by the records,
    // in_iv was N times what the engine's CSR holds)
    const u64 n_iv = iv_bound ? std::min<u64>(iv_bound, 2 * n) : 2 * n;
    const u64 nb = (n_reads + yk::kScanTile - 1) / yk::kScanTile;
    HIP_TRY(e->in_off.reserve((size_t)(n_reads + 1) * sizeof(u64)));
    HIP_TRY(e->in_iv.reserve((size_t)(n_iv + 1) * sizeof(uint2)));
    HIP_TRY(cnt.reserve((size_t)(n_reads + 4) * sizeof(u32)));
    HIP_TRY(part.reserve((size_t)(nb + 1) * sizeof(u64)));
    HIP_TRY(err.reserve(64));
    // (`counted`: cnt[] already holds every read's number of intervals — the device parser counts while it parses)
    if (!counted) HIP_TRY(hipMemsetAsync(cnt.p, 0, (size_t)(n_reads + 4) * sizeof(u32), e->stream));
    HIP_TRY(hipMemsetAsync(err.p, 0, 64, e->stream));
    const u32 R32 = (u32)n_reads;
    auto grid_for = [&](uint64_t recs) {
        return (u32)std::min<uint64_t>((recs + yk::kCsrThreads - 1) / yk::kCsrThreads, (uint64_t)e->num_cu * 16);
    };
    for (size_t i = 0; i < n_slabs && !counted; i++)
        if (slabs[i].n)
            hipLaunchKernelGGL(yk::csr_count_kernel, dim3(grid_for(slabs[i].n)), dim3(yk::kCsrThreads), 0, e->stream,
                               slabs[i].recs, (u64)slabs[i].n, d_map, (u64)n_handles, R32, cnt.as<u32>(), err.as<u32>());
    if (n_reads) {
        hipLaunchKernelGGL(yk::scan_tile_sums_kernel, dim3((u32)nb), dim3(yk::kScanT), 0, e->stream, cnt.as<u32>(),
                           (u64)n_reads, part.as<u64>());
        hipLaunchKernelGGL(yk::scan_parts_kernel, dim3(1), dim3(yk::kScanT), 0, e->stream, part.as<u64>(), nb,
                           e->in_off.as<u64>() + n_reads);
        hipLaunchKernelGGL(yk::scan_tiles_kernel, dim3((u32)nb), dim3(yk::kScanT), 0, e->stream, cnt.as<u32>(),
                           (u64)n_reads, part.as<u64>(), e->in_off.as<u64>());
    } else {
        HIP_TRY(hipMemsetAsync(e->in_off.p, 0, sizeof(u64), e->stream));
    }
    for (size_t i = 0; i < n_slabs; i++)
        if (slabs[i].n)
            hipLaunchKernelGGL(yk::csr_scatter_kernel, dim3(grid_for(slabs[i].n)), dim3(yk::kCsrThreads), 0, e->stream,
                               slabs[i].recs, (u64)slabs[i].n, d_map, (u64)n_handles, R32, e->in_off.as<u64>(),
                               cnt.as<u32>(), e->in_iv.as<uint2>());
    if (done) HIP_TRY(hipEventRecord(done, e->stream));
    u32 h_err = 0;
    u64 h_total = 0; // (fewer than two per record when reads live elsewhere: kSkipRead)
    HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(&h_total, e->in_off.as<u64>() + n_reads, sizeof(u64), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    if (h_err) return fail(YACRD_EINVAL, "a record names a read outside handle_map / n_reads");
    if (n_intervals) *n_intervals = h_total;
    return YACRD_OK;
}
// exclusive prefix sums u32[n] -> u64[n + 1] on the engine's stream (asynchronous)
int scan_u32_to_u64(yacrd_engine *e, const u32 *in, u64 n, u64 *out, DevBuf &part)
{
    const u64 nb = (n + yk::kScanTile - 1) / yk::kScanTile;
    HIP_TRY(part.reserve((size_t)(nb + 1) * sizeof(u64)));
    if (!n) {
        HIP_TRY(hipMemsetAsync(out, 0, sizeof(u64), e->stream));
        return YACRD_OK;
    }
    hipLaunchKernelGGL(yk::scan_tile_sums_kernel, dim3((u32)nb), dim3(yk::kScanT), 0, e->stream, in, n, part.as<u64>());
    hipLaunchKernelGGL(yk::scan_parts_kernel, dim3(1), dim3(yk::kScanT), 0, e->stream, part.as<u64>(), nb, out + n);
    hipLaunchKernelGGL(yk::scan_tiles_kernel, dim3((u32)nb), dim3(yk::kScanT), 0, e->stream, in, n, part.as<u64>(), out);
    return YACRD_OK;
}
} // namespace yke

extern "C" {

int yacrd_stream_open(yacrd_engine *e, uint64_t chunk_records, uint32_t n_buffers, yacrd_stream **out)
{
    if (!e || !out) return fail(YACRD_EINVAL, "null argument");
    *out = nullptr;
    if (chunk_records == 0) chunk_records = 131072;
    if (n_buffers == 0) n_buffers = 2 * std::min(64u, usable_cpus()) + 2;
    if (n_buffers < 2) n_buffers = 2;
    if (chunk_records > (1ull << 28)) return fail(YACRD_EINVAL, "chunk_records too large");
    DeviceGuard guard(e->device);
    yacrd_stream *s = new (std::nothrow) yacrd_stream();
    if (!s) return fail(YACRD_ENOMEM, "host allocation failed");
    s->e = e;
    s->chunk_records = chunk_records;
    s->n_buffers = n_buffers;
    s->state.assign(n_buffers, BUF_FREE);
    s->ev0.assign(n_buffers, nullptr);
    s->ev1.assign(n_buffers, nullptr);
    hipError_t err = hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking);
    if (err == hipSuccess)
        err = hipHostMalloc((void **)&s->arena, (size_t)n_buffers * chunk_records * sizeof(yacrd_ovl_rec));
    for (uint32_t b = 0; b < n_buffers && err == hipSuccess; b++) {
        err = hipEventCreate(&s->ev0[b]);
        if (err == hipSuccess) err = hipEventCreate(&s->ev1[b]);
    }
    if (err == hipSuccess) err = hipEventCreate(&s->evb0);
    if (err == hipSuccess) err = hipEventCreate(&s->evb1);
    if (err != hipSuccess) {
        yacrd_stream_close(s);
        return fail(err == hipErrorOutOfMemory ? YACRD_ENOMEM : YACRD_ENODEV,
                    std::string("stream setup: ") + hipGetErrorString(err));
    }
    *out = s;
    return YACRD_OK;
}

int yacrd_stream_sink(yacrd_stream *s, yacrd_rec_sink *sink)
{
    if (!s || !sink) return fail(YACRD_EINVAL, "null argument");
    sink->ctx = s;
    sink->acquire = sink_acquire;
    sink->commit = sink_commit;
    return YACRD_OK;
}

int yacrd_stream_acquire(yacrd_stream *s, yacrd_ovl_rec **buf, uint64_t *capacity)
{
    if (!s || !buf) return fail(YACRD_EINVAL, "null argument");
    *buf = nullptr;
    DeviceGuard guard(s->e->device);
    for (;;) {
        {
            std::lock_guard<std::mutex> g(s->mu);
            int pick = -1;
            for (uint32_t b = 0; b < s->n_buffers && pick < 0; b++)
                if (s->state[b] == BUF_FREE) pick = (int)b;
            for (uint32_t b = 0; b < s->n_buffers && pick < 0; b++) {
                if (s->state[b] != BUF_FLYING) continue;
                const hipError_t q = hipEventQuery(s->ev1[b]);
                if (q == hipSuccess) {
                    retire(s, b);
                    pick = (int)b;
                } else if (q != hipErrorNotReady) {
                    return fail(YACRD_ENODEV, std::string("stream copy: ") + hipGetErrorString(q));
                } else {
                    (void)hipGetLastError();
                }
            }
            if (pick >= 0) {
                s->state[pick] = BUF_HELD;
                *buf = reinterpret_cast<yacrd_ovl_rec *>(s->arena) + (size_t)pick * s->chunk_records;
                if (capacity) *capacity = s->chunk_records;
                return YACRD_OK;
            }
        }
        struct timespec ts = {0, 20000}; // every buffer is in flight: PCIe is the bottleneck right now
        nanosleep(&ts, nullptr);
    }
}

int yacrd_stream_commit(yacrd_stream *s, yacrd_ovl_rec *buf, uint64_t n)
{
    if (!s || !buf) return fail(YACRD_EINVAL, "null argument");
    const size_t at = (size_t)(buf - reinterpret_cast<yacrd_ovl_rec *>(s->arena));
    if (at % s->chunk_records != 0 || at / s->chunk_records >= s->n_buffers)
        return fail(YACRD_EINVAL, "not a buffer of this stream");
    const uint32_t b = (uint32_t)(at / s->chunk_records);
    if (n > s->chunk_records) return fail(YACRD_EINVAL, "more records than the buffer holds");
    DeviceGuard guard(s->e->device);
    std::lock_guard<std::mutex> g(s->mu);
    if (s->state[b] != BUF_HELD) return fail(YACRD_EINVAL, "buffer was not acquired");
    if (n == 0) {
        s->state[b] = BUF_FREE;
        return YACRD_OK;
    }
    // room in the current slab, or the next one (256 MiB of records first, doubling up to 4 GiB;
    // the tail of a slab that cannot take a whole buffer stays unused)
    auto fits = [&](size_t i) { return i < s->slabs.size() && s->slabs[i].cap - s->slabs[i].used >= n; };
    while (!fits(s->cur_slab)) {
        if (s->cur_slab < s->slabs.size()) {
            s->cur_slab++;
            continue;
        }
        uint64_t cap = ((uint64_t)256 << 20) / sizeof(yacrd_ovl_rec);
        if (!s->slabs.empty())
            cap = std::min<uint64_t>(s->slabs.back().cap * 2, ((uint64_t)4 << 30) / sizeof(yacrd_ovl_rec));
        cap = std::max<uint64_t>(cap, n);
        s->slabs.emplace_back();
        yacrd_stream::Slab &fresh = s->slabs.back();
        const hipError_t er = hipMalloc(&fresh.buf.p, (size_t)cap * sizeof(yacrd_ovl_rec));
        if (er != hipSuccess) {
            s->slabs.pop_back();
            s->state[b] = BUF_FREE;
            return fail(YACRD_ENOMEM, std::string("stream slab: ") + hipGetErrorString(er));
        }
        fresh.buf.cap = (size_t)cap * sizeof(yacrd_ovl_rec);
        fresh.cap = cap;
    }
    yacrd_stream::Slab &sl = s->slabs[s->cur_slab];
    char *dst = sl.buf.as<char>() + (size_t)sl.used * sizeof(yacrd_ovl_rec);
    HIP_TRY(hipEventRecord(s->ev0[b], s->copy));
    HIP_TRY(hipMemcpyAsync(dst, buf, (size_t)n * sizeof(yacrd_ovl_rec), hipMemcpyHostToDevice, s->copy));
    HIP_TRY(hipEventRecord(s->ev1[b], s->copy));
    sl.used += n;
    s->n_records += n;
    s->state[b] = BUF_FLYING;
    return YACRD_OK;
}

// forget every record the stream holds (s->mu held by the caller)
static void stream_reset_locked(yacrd_stream *s)
{
    for (auto &sl : s->slabs) sl.used = 0;
    s->cur_slab = 0;
    s->n_records = 0;
    s->busy_ms = 0;
}

int yacrd_stream_reset(yacrd_stream *s)
{
    if (!s) return fail(YACRD_EINVAL, "null argument");
    DeviceGuard guard(s->e->device);
    std::lock_guard<std::mutex> g(s->mu);
    for (uint32_t b = 0; b < s->n_buffers; b++)
        if (s->state[b] == BUF_HELD) return fail(YACRD_EINVAL, "a buffer is still held by a parser");
    HIP_TRY(hipStreamSynchronize(s->copy)); // copies in flight land in slabs that are about to be reused
    for (uint32_t b = 0; b < s->n_buffers; b++)
        if (s->state[b] == BUF_FLYING) retire(s, b);
    stream_reset_locked(s);
    return YACRD_OK;
}

int yacrd_stream_finish(yacrd_stream *s, const uint32_t *handle_map, uint64_t n_handles,
                        const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                        double not_coverage, yacrd_result *out)
{
    if (!s || !out) return fail(YACRD_EINVAL, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (n_reads && !lengths) return fail(YACRD_EINVAL, "null lengths");
    if (n_reads >= 0xFFFFFFFFull) return fail(YACRD_EINVAL, "n_reads must be < 2^32 - 1");
    yacrd_engine *e = s->e;
    if (e->pending.active || e->host_pending) return fail(YACRD_EINVAL, "the engine has a submitted batch pending");
    DeviceGuard guard(e->device);
    std::lock_guard<std::mutex> g(s->mu);
    for (uint32_t b = 0; b < s->n_buffers; b++)
        if (s->state[b] == BUF_HELD) return fail(YACRD_EINVAL, "a buffer is still held by a parser");
    // Whatever happens from here on, the stream is empty afterwards: records of a failed finish must not mix
    // with the next file's (their handles come from another id table).  (The two misuse returns above — a batch
    // pending on the engine, a buffer a parser still holds — touch nothing: yacrd_stream_reset after fixing them.)
    struct ResetOnExit {
        yacrd_stream *s;
        ~ResetOnExit()
        {
            (void)hipStreamSynchronize(s->copy); // (copies in flight land in slabs about to be reused)
            for (uint32_t b = 0; b < s->n_buffers; b++)
                if (s->state[b] == BUF_FLYING) retire(s, b);
            stream_reset_locked(s);
        }
    } reset_on_exit{s};
    HIP_TRY(hipStreamSynchronize(s->copy)); // every record is in HBM
    for (uint32_t b = 0; b < s->n_buffers; b++)
        if (s->state[b] == BUF_FLYING) retire(s, b);
    const uint64_t n = s->n_records, n_iv = 2 * n;
    s->stats = yacrd_stream_stats{};
    s->stats.n_records = n;
    s->stats.h2d_bytes = n * sizeof(yacrd_ovl_rec);
    s->stats.h2d_busy_ms = (float)s->busy_ms;
    if (n && n_reads == 0) return fail(YACRD_EINVAL, "records without reads");

    HIP_TRY(e->in_len.reserve((size_t)(n_reads + 1) * sizeof(u32)));
    const u32 *d_map = nullptr;
    if (handle_map && n_handles) {
        HIP_TRY(s->map.reserve((size_t)n_handles * sizeof(u32)));
        d_map = s->map.as<u32>();
    }
    HIP_TRY(hipEventRecord(s->evb0, e->stream));
    int rc = YACRD_OK;
    if (d_map) rc = h2d(e, s->map.p, handle_map, (size_t)n_handles * sizeof(u32));
    if (!rc && n_reads) rc = h2d(e, e->in_len.p, lengths, (size_t)n_reads * sizeof(u32));
    if (rc) return rc;
    std::vector<RecSlab> rs;
    for (auto &sl : s->slabs)
        if (sl.used) rs.push_back(RecSlab{sl.buf.as<yk::OvlRec>(), sl.used});
    u64 n_iv_here = n_iv;
    rc = csr_from_records(e, rs.data(), rs.size(), d_map, n_handles, n_reads, s->cnt, s->part, s->err, s->evb1, &n_iv_here);
    if (rc) return rc;
    s->stats.build_ms = ev_ms(s->evb0, s->evb1);
    const double t0 = now_ms();
    rc = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), n_reads, n_iv_here,
                       coverage, not_coverage);
    if (rc) return rc;
    s->stats.run_ms = (float)(now_ms() - t0);
    rc = fetch_result(e, out);
    s->stats.d2h_ms = e->timing.d2h_ms;
    return rc;
}

// ---- the group: one stream per device, records routed by handle mod N while the parser runs ----------
struct yacrd_stream_group {
    std::vector<yacrd_stream *> st;
    uint64_t chunk = 0; // records per staging buffer (what a parser thread fills)
    uint64_t piece = 0; // records per routed piece (what leaves for a device in one copy)
    // What a parser thread fills is a plain host buffer (a lane's staging); commit sorts its records into the
    // lane's per-device pieces, and a full piece goes through a pinned buffer of that device's stream — held
    // only for the memcpy, never between calls: any number of parser threads works with any number of buffers.
    struct Lane {
        yacrd_ovl_rec *staging = nullptr;
        bool held = false;
        std::vector<yacrd_ovl_rec *> out; // per device: piece records
        std::vector<uint64_t> fill;
        ~Lane()
        {
            std::free(staging);
            for (yacrd_ovl_rec *p : out) std::free(p);
        }
    };
    std::mutex mu;
    std::vector<std::unique_ptr<Lane>> lanes;
    std::vector<uint64_t> owned; // reads per device in the last finish
};

namespace {

int group_acquire(void *ctx, yacrd_ovl_rec **buf, uint64_t *cap)
{
    yacrd_stream_group *g = (yacrd_stream_group *)ctx;
    if (!g || !buf) return fail(YACRD_EINVAL, "null argument");
    *buf = nullptr;
    std::lock_guard<std::mutex> lock(g->mu);
    yacrd_stream_group::Lane *lane = nullptr;
    for (auto &l : g->lanes)
        if (!l->held) {
            lane = l.get();
            break;
        }
    if (!lane) {
        std::unique_ptr<yacrd_stream_group::Lane> fresh(new (std::nothrow) yacrd_stream_group::Lane());
        if (!fresh) return fail(YACRD_ENOMEM, "host allocation failed");
        fresh->staging = (yacrd_ovl_rec *)std::malloc((size_t)g->chunk * sizeof(yacrd_ovl_rec));
        if (!fresh->staging) return fail(YACRD_ENOMEM, "host allocation failed");
        fresh->out.assign(g->st.size(), nullptr);
        fresh->fill.assign(g->st.size(), 0);
        for (size_t d = 0; d < g->st.size(); d++) {
            fresh->out[d] = (yacrd_ovl_rec *)std::malloc((size_t)g->piece * sizeof(yacrd_ovl_rec));
            if (!fresh->out[d]) return fail(YACRD_ENOMEM, "host allocation failed");
        }
        lane = fresh.get();
        g->lanes.push_back(std::move(fresh));
    }
    lane->held = true;
    *buf = lane->staging;
    if (cap) *cap = g->chunk;
    return YACRD_OK;
}

// a lane's piece for device d leaves: pinned buffer, memcpy, commit (the DMA starts)
int lane_send(yacrd_stream_group *g, yacrd_stream_group::Lane *lane, uint32_t d)
{
    const uint64_t n = lane->fill[d];
    if (!n) return YACRD_OK;
    lane->fill[d] = 0;
    yacrd_ovl_rec *pin = nullptr;
    uint64_t cap = 0;
    int rc = yacrd_stream_acquire(g->st[d], &pin, &cap);
    if (rc) return rc;
    std::memcpy(pin, lane->out[d], (size_t)n * sizeof(yacrd_ovl_rec)); // (piece <= the stream's chunk)
    return yacrd_stream_commit(g->st[d], pin, n);
}

int group_commit(void *ctx, yacrd_ovl_rec *buf, uint64_t n)
{
    yacrd_stream_group *g = (yacrd_stream_group *)ctx;
    if (!g || !buf) return fail(YACRD_EINVAL, "null argument");
    yacrd_stream_group::Lane *lane = nullptr;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        for (auto &l : g->lanes)
            if (l->staging == buf && l->held) lane = l.get();
    }
    if (!lane) return fail(YACRD_EINVAL, "not a held buffer of this stream group");
    const uint32_t N = (uint32_t)g->st.size();
    int rc = n > g->chunk ? fail(YACRD_EINVAL, "more records than the buffer holds") : YACRD_OK;
    const uint64_t piece = g->piece;
    for (uint64_t i = 0; i < n && !rc; i++) {
        const yacrd_ovl_rec &r = buf[i];
        const uint32_t da = r.a % N, db = r.b % N; // == yacrd_stream_device_of
        lane->out[da][lane->fill[da]++] = r;
        if (lane->fill[da] == piece) rc = lane_send(g, lane, da);
        if (db != da && !rc) {
            lane->out[db][lane->fill[db]++] = r;
            if (lane->fill[db] == piece) rc = lane_send(g, lane, db);
        }
    }
    std::lock_guard<std::mutex> lock(g->mu);
    lane->held = false;
    return rc;
}

// every lane's pieces leave for their devices, or are dropped (g->mu held; no lane may be held)
int group_flush_locked(yacrd_stream_group *g, bool keep)
{
    int rc = YACRD_OK;
    for (auto &l : g->lanes) {
        if (l->held) return fail(YACRD_EINVAL, "a buffer is still held by a parser");
        for (size_t d = 0; d < g->st.size(); d++) {
            if (keep) {
                const int r1 = lane_send(g, l.get(), (uint32_t)d);
                if (r1 && !rc) rc = r1;
            }
            l->fill[d] = 0;
        }
    }
    return rc;
}

} // namespace

uint32_t yacrd_stream_device_of(uint32_t handle, uint32_t n_devices)
{
    return n_devices ? handle % n_devices : 0u;
}

int yacrd_stream_group_open(yacrd_engine *const *engines, uint32_t n_engines, uint64_t chunk_records,
                            uint32_t n_buffers, yacrd_stream_group **out)
{
    if (!engines || n_engines == 0 || !out) return fail(YACRD_EINVAL, "bad argument");
    *out = nullptr;
    for (uint32_t d = 0; d < n_engines; d++)
        if (!engines[d]) return fail(YACRD_EINVAL, "null engine");
    yacrd_stream_group *g = new (std::nothrow) yacrd_stream_group();
    if (!g) return fail(YACRD_ENOMEM, "host allocation failed");
    if (chunk_records == 0) chunk_records = 131072;
    g->chunk = chunk_records;
    g->piece = std::min<uint64_t>(chunk_records, 32768); // 768 KiB per copy: PCIe is near its peak from 256 KiB on
    g->owned.assign(n_engines, 0);
    for (uint32_t d = 0; d < n_engines; d++) {
        yacrd_stream *s = nullptr;
        const int rc = yacrd_stream_open(engines[d], chunk_records, n_buffers, &s);
        if (rc) {
            yacrd_stream_group_close(g);
            return rc;
        }
        g->st.push_back(s);
    }
    *out = g;
    return YACRD_OK;
}

int yacrd_stream_group_sink(yacrd_stream_group *g, yacrd_rec_sink *sink)
{
    if (!g || !sink) return fail(YACRD_EINVAL, "null argument");
    if (g->st.size() == 1) return yacrd_stream_sink(g->st[0], sink); // nothing to route
    sink->ctx = g;
    sink->acquire = group_acquire;
    sink->commit = group_commit;
    return YACRD_OK;
}

int yacrd_stream_group_reset(yacrd_stream_group *g)
{
    if (!g) return fail(YACRD_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(g->mu);
    int rc = group_flush_locked(g, false);
    for (yacrd_stream *s : g->st) {
        const int r1 = yacrd_stream_reset(s);
        if (r1 && !rc) rc = r1;
    }
    return rc;
}

int yacrd_stream_group_finish(yacrd_stream_group *g, const uint32_t *handle_map, uint64_t n_handles,
                              const uint32_t *lengths, uint64_t n_reads, uint32_t coverage,
                              double not_coverage, yacrd_result *out)
{
    if (!g || !out) return fail(YACRD_EINVAL, "null argument");
    const uint32_t N = (uint32_t)g->st.size();
    if (N == 1) {
        g->owned[0] = n_reads;
        return yacrd_stream_finish(g->st[0], handle_map, n_handles, lengths, n_reads, coverage, not_coverage, out);
    }
    std::memset(out, 0, sizeof(*out));
    if (n_reads && !lengths) return fail(YACRD_EINVAL, "null lengths");
    if (n_reads >= 0xFFFFFFFEull) return fail(YACRD_EINVAL, "n_reads must be < 2^32 - 2");
    std::lock_guard<std::mutex> lock(g->mu);
    // whatever happens below, no device keeps records of this file
    struct ResetOnExit {
        yacrd_stream_group *g;
        bool armed = true;
        ~ResetOnExit()
        {
            if (!armed) return;
            (void)group_flush_locked(g, false);
            for (yacrd_stream *s : g->st) (void)yacrd_stream_reset(s);
        }
    } reset_on_exit{g};
    int rc = group_flush_locked(g, true);
    if (rc) return rc;

    // read id -> handle (the identity without a map)
    if (!handle_map) n_handles = n_reads;
    std::vector<uint32_t> inv((size_t)n_reads, 0xFFFFFFFFu);
    for (uint64_t h = 0; h < n_handles; h++) {
        const uint32_t r = handle_map ? handle_map[h] : (uint32_t)h;
        if (r == 0xFFFFFFFFu) continue; // (a handle no record uses)
        if (r >= n_reads || inv[r] != 0xFFFFFFFFu) return fail(YACRD_EINVAL, "handle_map is not one-to-one onto the reads");
        inv[r] = (uint32_t)h;
    }
    for (uint64_t r = 0; r < n_reads; r++)
        if (inv[r] == 0xFFFFFFFFu) return fail(YACRD_EINVAL, "handle_map misses a read");

    std::vector<yacrd_result> parts(N);
    std::vector<int> codes(N, YACRD_OK);
    std::vector<std::string> errs(N);
    auto work = [&](uint32_t d) {
        std::vector<uint32_t> local((size_t)n_handles, YACRD_HANDLE_ELSEWHERE), len_d;
        len_d.reserve((size_t)(n_reads / N + 1024));
        for (uint64_t r = 0; r < n_reads; r++) {
            const uint32_t h = inv[r];
            if (h % N != d) continue;
            local[h] = (uint32_t)len_d.size();
            len_d.push_back(lengths[r]);
        }
        g->owned[d] = len_d.size();
        codes[d] = yacrd_stream_finish(g->st[d], local.data(), n_handles, len_d.data(), len_d.size(), coverage,
                                       not_coverage, &parts[d]);
        if (codes[d]) errs[d] = yke::err_slot();
    };
    {
        std::vector<std::thread> th;
        for (uint32_t d = 1; d < N; d++) th.emplace_back(work, d);
        work(0);
        for (auto &t : th) t.join();
    }
    // (a finish that got as far as its records has emptied its stream; one that failed before that — the engine had a
    // batch pending, a copy failed — has not: then the group's reset stays armed and clears every device)
    bool any_failed = false;
    for (uint32_t d = 0; d < N; d++) any_failed |= codes[d] != YACRD_OK;
    reset_on_exit.armed = any_failed;
    for (uint32_t d = 0; d < N; d++)
        if (codes[d]) {
            const int code = codes[d];
            const std::string msg = "device " + std::to_string(d) + ": " + errs[d];
            for (auto &r : parts) yacrd_result_free(&r);
            return fail(code, msg);
        }
    // merge: first-appearance order again
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
    std::vector<uint64_t> next(N, 0);
    uint64_t g0 = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint32_t d = inv[r] % N;
        const uint64_t l = next[d]++;
        const uint64_t b0 = parts[d].bad_offsets[l], b1 = parts[d].bad_offsets[l + 1];
        out->bad_offsets[r] = g0;
        if (b1 > b0) std::memcpy(out->bad_regions + 2 * g0, parts[d].bad_regions + 2 * b0, (size_t)(b1 - b0) * 2 * sizeof(uint32_t));
        g0 += b1 - b0;
        out->read_type[r] = parts[d].read_type[l];
    }
    out->bad_offsets[n_reads] = g0;
    out->n_reads = n_reads;
    out->n_regions = g0;
    for (auto &r : parts) yacrd_result_free(&r);
    return YACRD_OK;
}

int yacrd_stream_group_last_stats(const yacrd_stream_group *g, uint32_t index, yacrd_stream_stats *st,
                                  uint64_t *n_reads_owned)
{
    if (!g || index >= g->st.size()) return fail(YACRD_EINVAL, "bad argument");
    if (n_reads_owned) *n_reads_owned = g->owned[index];
    return st ? yacrd_stream_last_stats(g->st[index], st) : YACRD_OK;
}

void yacrd_stream_group_close(yacrd_stream_group *g)
{
    if (!g) return;
    for (yacrd_stream *s : g->st) yacrd_stream_close(s);
    delete g;
}

int yacrd_stream_last_stats(const yacrd_stream *s, yacrd_stream_stats *st)
{
    if (!s || !st) return fail(YACRD_EINVAL, "null argument");
    *st = s->stats;
    return YACRD_OK;
}

void yacrd_stream_close(yacrd_stream *s)
{
    if (!s) return;
    DeviceGuard guard(s->e->device);
    if (s->copy) (void)hipStreamSynchronize(s->copy);
    for (auto &sl : s->slabs) sl.buf.release();
    s->cnt.release();
    s->part.release();
    s->map.release();
    s->err.release();
    if (s->arena) (void)hipHostFree(s->arena);
    for (hipEvent_t x : s->ev0)
        if (x) (void)hipEventDestroy(x);
    for (hipEvent_t x : s->ev1)
        if (x) (void)hipEventDestroy(x);
    if (s->evb0) (void)hipEventDestroy(s->evb0);
    if (s->evb1) (void)hipEventDestroy(s->evb1);
    if (s->copy) (void)hipStreamDestroy(s->copy);
    delete s;
}

} // extern "C"
