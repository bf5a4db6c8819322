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
                     DevBuf &cnt, DevBuf &part, DevBuf &err, hipEvent_t done)
{
    u64 n = 0;
    for (size_t i = 0; i < n_slabs; i++) n += slabs[i].n;
    const u64 n_iv = 2 * n;
    const u64 nb = (n_reads + yk::kScanTile - 1) / yk::kScanTile;
    HIP_TRY(e->in_off.reserve((size_t)(n_reads + 1) * sizeof(u64)));
    HIP_TRY(e->in_iv.reserve((size_t)(n_iv + 1) * sizeof(uint2)));
    HIP_TRY(cnt.reserve((size_t)(n_reads + 4) * sizeof(u32)));
    HIP_TRY(part.reserve((size_t)(nb + 1) * sizeof(u64)));
    HIP_TRY(err.reserve(64));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, (size_t)(n_reads + 4) * sizeof(u32), e->stream));
    HIP_TRY(hipMemsetAsync(err.p, 0, 64, e->stream));
    const u32 R32 = (u32)n_reads;
    auto grid_for = [&](uint64_t recs) {
        return (u32)std::min<uint64_t>((recs + yk::kCsrThreads - 1) / yk::kCsrThreads, (uint64_t)e->num_cu * 16);
    };
    for (size_t i = 0; i < n_slabs; i++)
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
    HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    if (h_err) return fail(YACRD_EINVAL, "a record names a read outside handle_map / n_reads");
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
    HIP_TRY(hipStreamSynchronize(s->copy)); // every record is in HBM
    for (uint32_t b = 0; b < s->n_buffers; b++)
        if (s->state[b] == BUF_FLYING) retire(s, b);
    const uint64_t n = s->n_records, n_iv = 2 * n;
    s->stats = yacrd_stream_stats{};
    s->stats.n_records = n;
    s->stats.h2d_bytes = n * sizeof(yacrd_ovl_rec);
    s->stats.h2d_busy_ms = (float)s->busy_ms;

    // Whatever happens below, the stream is empty afterwards: records of a failed finish must not mix
    // with the next file's (their handles come from another id table).
    struct ResetOnExit {
        yacrd_stream *s;
        ~ResetOnExit() { stream_reset_locked(s); }
    } reset_on_exit{s};
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
    rc = csr_from_records(e, rs.data(), rs.size(), d_map, n_handles, n_reads, s->cnt, s->part, s->err, s->evb1);
    if (rc) return rc;
    s->stats.build_ms = ev_ms(s->evb0, s->evb1);
    const double t0 = now_ms();
    rc = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), n_reads, n_iv,
                       coverage, not_coverage);
    if (rc) return rc;
    s->stats.run_ms = (float)(now_ms() - t0);
    rc = fetch_result(e, out);
    s->stats.d2h_ms = e->timing.d2h_ms;
    return rc;
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
