// tools/ingest_null.cc — the streaming ingest (yacrd_ingest_stream) into a sink that throws the records away:
// the parser's own rate, without a GPU.  g++ -O2 -o /tmp/ingest_null tools/ingest_null.cc -Lyacrd_amd/lib -lyacrd_host -Wl,-rpath,$PWD/yacrd_amd/lib -pthread
// usage: ingest_null file.paf threads [reps]
#include "../include/yacrd_host.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

struct Pool {
    std::mutex mu;
    std::vector<yacrd_ovl_rec *> free_bufs;
    uint64_t cap = 131072;
    std::atomic<uint64_t> recs{0};
};
static int acq(void *ctx, yacrd_ovl_rec **buf, uint64_t *cap)
{
    Pool *p = (Pool *)ctx;
    std::lock_guard<std::mutex> g(p->mu);
    if (p->free_bufs.empty()) *buf = (yacrd_ovl_rec *)std::malloc(p->cap * sizeof(yacrd_ovl_rec));
    else {
        *buf = p->free_bufs.back();
        p->free_bufs.pop_back();
    }
    *cap = p->cap;
    return 0;
}
static int com(void *ctx, yacrd_ovl_rec *buf, uint64_t n)
{
    Pool *p = (Pool *)ctx;
    p->recs += n;
    std::lock_guard<std::mutex> g(p->mu);
    p->free_bufs.push_back(buf);
    return 0;
}
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int th = std::atoi(argv[2]), reps = argc > 3 ? std::atoi(argv[3]) : 3;
    Pool pool;
    yacrd_rec_sink sink = {&pool, acq, com};
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        pool.recs = 0;
        yacrd_csr *h = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        if (yacrd_ingest_stream(argv[1], 0, th, &sink, &h)) {
            std::fprintf(stderr, "error: %s\n", yacrd_host_last_error());
            return 1;
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        yacrd_csr_free(h);
        if (dt < best) best = dt;
    }
    std::printf("threads %d: %.1f ms, %.2f M overlaps/s (%llu records), %.1f ns per line per thread\n", th, best * 1e3,
                pool.recs.load() / best / 1e6, (unsigned long long)pool.recs.load(), best * 1e9 / pool.recs.load() * th);
    return 0;
}
