// paf_csr.cc — overlap ingest: PAF / M4 text -> the CSR the engine consumes.
//
// Replaces reference Reads2Ovl::init_paf / init_m4 (src/reads2ovl/mod.rs:83-145; column contract
// src/io.rs:23-50) and FullMemory::add_overlap_and_length (src/reads2ovl/fullmemory.rs:82-90):
//   * both reads of every record get an interval (mod.rs:108-109 / :140-141)
//   * a read's length is the FIRST length seen for its id (fullmemory.rs:82-90)
//   * records may carry extra columns (csv `flexible(true)`), empty lines are skipped
//   * a short or non-numeric record is an error (the reference bails, mod.rs:93-97)
// Reads are numbered in first-appearance order.  Parsing is chunk-parallel over one shared id
// table whose entries remember where in the file their id came first, so numbering and the
// first-length rule do not depend on the thread count.  csv-crate quoting ("...") is not interpreted (unpinned by
// the reference's tests): a '"' is an ordinary byte here.
#include "../../../include/yacrd_host.h"
#include "host_common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <memory>
#include <mutex>
#include <new>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace yh {
std::string &err_slot()
{
    thread_local std::string s;
    return s;
}
} // namespace yh

struct yacrd_csr {
    std::vector<uint64_t> offsets;
    std::unique_ptr<uint32_t[]> intervals; // 2 * n_intervals, deliberately not zero-filled
    std::vector<uint32_t> lengths;
    std::vector<uint64_t> name_off;
    std::vector<char> names;
    uint64_t n_records = 0;
    std::vector<uint32_t> table; // open addressing over read ids (value = id + 1), lazy
    uint64_t mask = 0;
    std::once_flag table_once;
};

namespace {

enum { FMT_AUTO = 0, FMT_PAF = 1, FMT_M4 = 2 };

// Allocator of the ingest's big arrays: anonymous mappings with MADV_HUGEPAGE from 1 MiB up.
// First-touch page faults on 4 KiB pages top out at ~12 GB/s on this class of host however many
// threads fault (tools/io_probe.cc: 120 ms for 1.45 GB at 16 and at 64 threads, 11 ms with 2 MiB
// pages), and the parse touches ~1 GB of fresh memory per 20 M overlaps.
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U>
    HugeAlloc(const HugeAlloc<U> &) {}
    static constexpr size_t kBig = 1u << 20, kHuge = 2u << 20;
    static size_t rounded(size_t bytes) { return (bytes + kHuge - 1) & ~(kHuge - 1); }
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kBig) {
            void *p = std::malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
            return static_cast<T *>(p);
        }
        void *p = mmap(nullptr, rounded(bytes), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(p, rounded(bytes), MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kBig) std::free(p);
        else munmap(p, rounded(bytes));
    }
    template <class U>
    bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T>
using big_vector = std::vector<T, HugeAlloc<T>>;

struct Rec {
    uint32_t a, b, sa, ea, sb, eb;
};

// Zero-filled memory for the id table, carved out of 64 MiB anonymous regions with MADV_HUGEPAGE:
// the table is probed at random, and on 4 KiB pages every probe also misses the TLB.
struct HugePool {
    std::mutex mu;
    char *cur = nullptr;
    size_t left = 0;
    std::vector<std::pair<void *, size_t>> regions;
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 63) & ~(size_t)63;
        std::lock_guard<std::mutex> g(mu);
        if (bytes > left) {
            const size_t sz = std::max<size_t>((size_t)64 << 20, (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
            void *p = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (p == MAP_FAILED) throw std::bad_alloc();
            (void)madvise(p, sz, MADV_HUGEPAGE);
            regions.emplace_back(p, sz);
            cur = (char *)p;
            left = sz;
        }
        void *r = cur;
        cur += bytes;
        left -= bytes;
        return r;
    }
    ~HugePool()
    {
        for (auto &r : regions) munmap(r.first, r.second);
    }
};

// Read ids are interned into ONE table shared by the parse threads (per-chunk tables made every
// chunk re-intern nearly every id: the parse stopped scaling at 16 threads).  The table is
// sharded by the top hash bits.  Lookups — 99 % of the calls — take no lock and write nothing
// shared: a shard publishes an immutable open-addressing index (slot = hash, name length, entry
// number; the entry number is release-stored last) over entries and name bytes that never move
// (geometrically growing blocks).  A miss, or a hit at an earlier file position than the entry
// knows, takes the shard's spin lock: inserts re-probe the current index, a full index is replaced
// by one twice the size (the old one stays allocated until the end: readers may still walk it,
// and fall back to the locked path when it misses).  An entry remembers the smallest position
// (byte offset of the line * 2 + 0/1 for the first / second id) at which its id was seen and the
// length given there: global numbering = order of those positions (first appearance in the
// file) and a read's length = the first length seen (fullmemory.rs:82-90), whatever the timing.
struct IdTable {
    static constexpr uint32_t kIdxBits = 22, kBlock0 = 64, kBlocks = 17; // 64 * (2^17 - 1) > 2^22
    // What a lookup touches is kept small (16 B per id + the name bytes + a 16 B slot: for 400 k ids
    // 27 MB, an L3 slice; with hash, length and name length in the same record it was 37 MB and
    // a single thread parsed 1.6x slower).
    struct Hot {
        const char *name;
        std::atomic<uint64_t> first_pos; // smallest position at which the id was seen
    };
    struct Cold {
        uint64_t hash;
        uint64_t first_len; // length given at first_pos (written under the lock)
        uint32_t nlen;
    };
    struct Slot {
        uint64_t hash;
        uint32_t nlen;
        std::atomic<uint32_t> idx1; // entry number + 1, 0 = empty; stored last (release)
    };
    struct Index {
        uint32_t mask;
        Slot *slots;
    };
    struct alignas(64) Shard {
        alignas(64) std::atomic<bool> lock{false};
        alignas(64) std::atomic<Index *> index{nullptr};
        Hot *hot[kBlocks] = {};
        Cold *cold[kBlocks] = {};
        uint32_t n_entries = 0;
        char *name_cur = nullptr;
        size_t name_left = 0;
    };
    size_t n_shards = 1;
    int shard_shift = 63; // shard = (hash >> shard_shift) & (n_shards - 1)
    std::unique_ptr<Shard[]> shards;
    HugePool pool; // indexes, entries and name bytes; released as a whole
    std::atomic<bool> overflow{false};

    explicit IdTable(size_t want_shards)
    {
        while (n_shards < want_shards) n_shards <<= 1;
        shard_shift = n_shards == 1 ? 63 : 64 - __builtin_ctzll((unsigned long long)n_shards);
        shards.reset(new Shard[n_shards]);
    }
    IdTable(const IdTable &) = delete;
    IdTable &operator=(const IdTable &) = delete;

    static void locate(uint32_t idx, uint32_t &block, uint32_t &at)
    {
        const uint32_t v = idx + kBlock0;
        block = 31u - (uint32_t)__builtin_clz(v) - 6u;
        at = v - (kBlock0 << block);
    }
    static Hot &hot(const Shard &sh, uint32_t idx)
    {
        uint32_t k, at;
        locate(idx, k, at);
        return sh.hot[k][at];
    }
    static Cold &cold(const Shard &sh, uint32_t idx)
    {
        uint32_t k, at;
        locate(idx, k, at);
        return sh.cold[k][at];
    }
    Index *new_index(uint32_t cap)
    {
        Index *ix = (Index *)pool.alloc(sizeof(Index));
        Slot *sl = (Slot *)pool.alloc((size_t)cap * sizeof(Slot)); // all-zero = empty slots
        ix->mask = cap - 1;
        ix->slots = sl;
        return ix;
    }
    // lock held.  Puts entry idx into ix (no duplicates possible).
    static void place(Index *ix, uint64_t hash, uint32_t nlen, uint32_t idx)
    {
        uint32_t s2 = (uint32_t)hash & ix->mask;
        while (ix->slots[s2].idx1.load(std::memory_order_relaxed)) s2 = (s2 + 1) & ix->mask;
        ix->slots[s2].hash = hash;
        ix->slots[s2].nlen = nlen;
        ix->slots[s2].idx1.store(idx + 1, std::memory_order_release);
    }
    // returns the entry number, or ~0u when absent
    static uint32_t probe(const Shard &sh, const Index *ix, const char *p, size_t n, uint64_t h)
    {
        uint32_t s2 = (uint32_t)h & ix->mask;
        for (;;) {
            const Slot &sl = ix->slots[s2];
            const uint32_t v = sl.idx1.load(std::memory_order_acquire);
            if (!v) return ~0u;
            if (sl.hash == h && sl.nlen == n && std::memcmp(hot(sh, v - 1).name, p, n) == 0) return v - 1;
            s2 = (s2 + 1) & ix->mask;
        }
    }
    // returns shard << kIdxBits | entry number
    uint32_t intern(const char *p, size_t n, uint64_t h, uint64_t length, uint64_t pos)
    {
        const uint32_t si = (uint32_t)((h >> shard_shift) & (n_shards - 1));
        Shard &sh = shards[si];
        const Index *ix = sh.index.load(std::memory_order_acquire);
        uint32_t idx = ix ? probe(sh, ix, p, n, h) : ~0u;
        if (idx != ~0u && pos >= hot(sh, idx).first_pos.load(std::memory_order_relaxed))
            return (si << kIdxBits) | idx; // the common case: nothing shared is written

        while (sh.lock.exchange(true, std::memory_order_acquire))
            while (sh.lock.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        Index *cur = sh.index.load(std::memory_order_relaxed);
        if (!cur) {
            cur = new_index(64);
            sh.index.store(cur, std::memory_order_release);
        }
        if (idx == ~0u) idx = probe(sh, cur, p, n, h); // somebody else may have inserted it
        if (idx == ~0u) {
            idx = sh.n_entries;
            if (idx >= (1u << kIdxBits)) {
                overflow.store(true, std::memory_order_relaxed);
                sh.lock.store(false, std::memory_order_release);
                return si << kIdxBits;
            }
            uint32_t k, at;
            locate(idx, k, at);
            if (!sh.hot[k]) {
                sh.hot[k] = (Hot *)pool.alloc(sizeof(Hot) * ((size_t)kBlock0 << k));
                sh.cold[k] = (Cold *)pool.alloc(sizeof(Cold) * ((size_t)kBlock0 << k));
            }
            if (sh.name_left < n) {
                const size_t sz = std::max<size_t>(n, 4u << 10);
                sh.name_cur = (char *)pool.alloc(sz);
                sh.name_left = sz;
            }
            std::memcpy(sh.name_cur, p, n);
            Hot *e = new (&sh.hot[k][at]) Hot();
            e->name = sh.name_cur;
            e->first_pos.store(pos, std::memory_order_relaxed);
            sh.cold[k][at] = Cold{h, length, (uint32_t)n};
            sh.name_cur += n;
            sh.name_left -= n;
            sh.n_entries = idx + 1;
            if ((uint64_t)(idx + 2) * 2 > (uint64_t)cur->mask + 1) { // keep the load below 1/2
                Index *bigger = new_index((cur->mask + 1) * 2);
                for (uint32_t k2 = 0; k2 <= idx; k2++) {
                    const Cold &c2 = cold(sh, k2);
                    place(bigger, c2.hash, c2.nlen, k2);
                }
                sh.index.store(bigger, std::memory_order_release);
            } else {
                place(cur, h, (uint32_t)n, idx);
            }
        } else {
            Hot &e = hot(sh, idx);
            if (pos < e.first_pos.load(std::memory_order_relaxed)) {
                cold(sh, idx).first_len = length;
                e.first_pos.store(pos, std::memory_order_relaxed);
            }
        }
        sh.lock.store(false, std::memory_order_release);
        return (si << kIdxBits) | idx;
    }
};

// One per parse thread.  Aligned and padded to its own cache lines: the threads update their
// chunk's counters and vector ends on every line, and neighbours sharing a line cost 10x.
struct alignas(256) Chunk {
    const char *begin = nullptr, *end = nullptr;
    IdTable *ids = nullptr;
    const char *text = nullptr; // start of the whole input (positions are relative to it)
    big_vector<Rec> recs; // a / b hold IdTable handles until the global numbering exists
    std::string error;
    uint64_t error_line = 0; // 1-based within chunk
    uint64_t lines = 0;
};

inline bool parse_u64(const char *p, const char *e, uint64_t &out)
{
    if (p < e && *p == '+') p++;
    if (p == e) return false;
    uint64_t v = 0;
    for (; p < e; p++) {
        const unsigned d = (unsigned char)*p - '0';
        if (d > 9) return false;
        if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}
inline bool parse_u32(const char *p, const char *e, uint32_t &out)
{
    uint64_t v;
    if (!parse_u64(p, e, v) || v > 0xFFFFFFFFull) return false;
    out = (uint32_t)v;
    return true;
}
inline bool is_one_char(const char *p, const char *e)
{ // serde `char`: exactly one UTF-8 scalar
    if (p == e) return false;
    const unsigned char c = (unsigned char)*p;
    const int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
    return n != 0 && e - p == n;
}

// ---- PAF fast path: one forward scan per line; ids are hashed while they are scanned ----------
// Same acceptance as the generic path below (src/io.rs:23-34): 9 leading tab-separated fields,
// u64 lengths, u32 positions (optional '+'), one-character strand, anything after ignored.
inline bool scan_id(const char *&p, const char *le, uint64_t &h, const char *&b, size_t &n)
{
    b = p;
    uint64_t x = 0xcbf29ce484222325ull;
    while (p < le && *p != '\t') {
        x ^= (unsigned char)*p++;
        x *= 0x100000001b3ull;
    }
    if (p >= le) return false; // an id must be followed by more fields
    n = (size_t)(p - b);
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    h = x;
    p++;
    return true;
}
inline bool scan_uint(const char *&p, const char *le, uint64_t limit, uint64_t &out, bool last)
{
    if (p < le && *p == '+') p++;
    const char *b = p;
    uint64_t v = 0;
    while (p < le) {
        const unsigned d = (unsigned char)*p - '0';
        if (d > 9) break;
        if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
        v = v * 10 + d;
        p++;
    }
    if (p == b || v > limit) return false;
    if (p < le) {
        if (*p != '\t') return false;
        p++;
    } else if (!last) {
        return false;
    }
    out = v;
    return true;
}

void parse_chunk_paf(Chunk &c)
{
    c.recs.reserve((size_t)(c.end - c.begin) / 48 + 16); // untouched pages cost nothing
    const char *p = c.begin;
    while (p < c.end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(c.end - p));
        if (!eol) eol = c.end;
        const char *le = eol;
        if (le > p && le[-1] == '\r') le--;
        c.lines++;
        if (le == p) {
            p = eol + 1;
            continue;
        }
        const char *q = p, *ida, *idb;
        size_t na, nb;
        uint64_t ha, hb, la, lb, sa, ea, sb, eb;
        bool ok = scan_id(q, le, ha, ida, na) && scan_uint(q, le, ~0ull, la, false) &&
                  scan_uint(q, le, 0xFFFFFFFFull, sa, false) &&
                  scan_uint(q, le, 0xFFFFFFFFull, ea, false);
        if (ok) { // strand: exactly one UTF-8 scalar, then a tab
            const char *t = (const char *)std::memchr(q, '\t', (size_t)(le - q));
            ok = t && is_one_char(q, t);
            q = t ? t + 1 : le;
        }
        ok = ok && scan_id(q, le, hb, idb, nb) && scan_uint(q, le, ~0ull, lb, false) &&
             scan_uint(q, le, 0xFFFFFFFFull, sb, false) && scan_uint(q, le, 0xFFFFFFFFull, eb, true);
        if (ok && (la > 0xFFFFFFFFull || lb > 0xFFFFFFFFull)) {
            c.error = "read length >= 2^32 is not supported by the engine";
            c.error_line = c.lines;
            return;
        }
        if (!ok) {
            c.error = "Reading of the file in paf format failed";
            c.error_line = c.lines;
            return;
        }
        Rec r;
        r.sa = (uint32_t)sa;
        r.ea = (uint32_t)ea;
        r.sb = (uint32_t)sb;
        r.eb = (uint32_t)eb;
        const uint64_t pos = (uint64_t)(p - c.text) * 2;
        r.a = c.ids->intern(ida, na, ha, la, pos);
        r.b = c.ids->intern(idb, nb, hb, lb, pos + 1);
        c.recs.push_back(r);
        p = eol + 1;
    }
}

void parse_chunk(Chunk &c, int format)
{
    const char delim = format == FMT_PAF ? '\t' : ' ';
    const int need = format == FMT_PAF ? 9 : 12;
    c.recs.reserve((size_t)(c.end - c.begin) / 48 + 16); // untouched pages cost nothing
    const char *p = c.begin;
    const char *fb[12], *fe[12];
    while (p < c.end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(c.end - p));
        if (!eol) eol = c.end;
        const char *le = eol;
        if (le > p && le[-1] == '\r') le--;
        c.lines++;
        if (le == p) { // csv skips empty lines
            p = eol + 1;
            continue;
        }
        int nf = 0;
        const char *q = p;
        while (nf < need) {
            const char *d = (const char *)std::memchr(q, delim, (size_t)(le - q));
            fb[nf] = q;
            fe[nf] = d ? d : le;
            nf++;
            if (!d) break;
            q = d + 1;
        }
        bool ok = nf == need;
        Rec r{};
        uint64_t la = 0, lb = 0;
        int ia = 0, ib = 0;
        if (ok && format == FMT_PAF) { // src/io.rs:23-34
            ia = 0;
            ib = 5;
            ok = parse_u64(fb[1], fe[1], la) && parse_u32(fb[2], fe[2], r.sa) &&
                 parse_u32(fb[3], fe[3], r.ea) && is_one_char(fb[4], fe[4]) &&
                 parse_u64(fb[6], fe[6], lb) && parse_u32(fb[7], fe[7], r.sb) &&
                 parse_u32(fb[8], fe[8], r.eb);
        } else if (ok) { // src/io.rs:36-50: a b err shared sa ba ea la sb bb eb lb
            ia = 0;
            ib = 1;
            uint64_t shared;
            char *endp = nullptr;
            std::string errf(fb[2], fe[2]);
            errno = 0;
            (void)std::strtod(errf.c_str(), &endp);
            ok = !errf.empty() && endp && *endp == '\0' && parse_u64(fb[3], fe[3], shared) &&
                 is_one_char(fb[4], fe[4]) && parse_u32(fb[5], fe[5], r.sa) &&
                 parse_u32(fb[6], fe[6], r.ea) && parse_u64(fb[7], fe[7], la) &&
                 is_one_char(fb[8], fe[8]) && parse_u32(fb[9], fe[9], r.sb) &&
                 parse_u32(fb[10], fe[10], r.eb) && parse_u64(fb[11], fe[11], lb);
        }
        if (ok && (la > 0xFFFFFFFFull || lb > 0xFFFFFFFFull)) {
            c.error = "read length >= 2^32 is not supported by the engine";
            c.error_line = c.lines;
            return;
        }
        if (!ok) {
            c.error = format == FMT_PAF ? "Reading of the file in paf format failed"
                                        : "Reading of the file in m4 format failed";
            c.error_line = c.lines;
            return;
        }
        const size_t na = (size_t)(fe[ia] - fb[ia]), nb = (size_t)(fe[ib] - fb[ib]);
        const uint64_t pos = (uint64_t)(p - c.text) * 2;
        r.a = c.ids->intern(fb[ia], na, yh::hash_bytes(fb[ia], na), la, pos);
        r.b = c.ids->intern(fb[ib], nb, yh::hash_bytes(fb[ib], nb), lb, pos + 1);
        c.recs.push_back(r);
        p = eol + 1;
    }
}

// run fn(task) for task in [0, n_tasks) on up to n_threads threads (dynamic hand-out)
template <class F>
void parallel_for(size_t n_tasks, size_t n_threads, F fn)
{
    if (n_tasks == 0) return;
    n_threads = std::max<size_t>(1, std::min(n_threads, n_tasks));
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= n_tasks) return;
            fn(t);
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < n_threads; i++) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
}

// CPUs this process may actually use: hardware threads, capped by the cgroup CPU quota (a
// container with cpu.max = "1600000 100000" gets 16 however many the machine has; threads beyond
// the quota only get throttled).
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

struct Phase {
    const bool on = std::getenv("YACRD_INGEST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ingest] %-10s %8.2f ms\n", what,
                     std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int build(const char *text, size_t len, int format, int n_threads, yacrd_csr **out)
{
    if (format != FMT_PAF && format != FMT_M4) return yh::fail("unknown overlap format");
    // auto: every usable CPU up to 64
    if (n_threads <= 0) n_threads = (int)std::min(64u, usable_cpus());
    if (n_threads <= 0) n_threads = 1;
    const size_t NT = (size_t)n_threads;
    // one chunk per thread, >= 2 MiB of text each
    const size_t T = std::max<size_t>(1, std::min<size_t>(NT, len / (2u << 20) + 1));
    Phase ph;

    IdTable ids(T == 1 ? 1 : 1024);
    std::vector<Chunk> chunks(T);
    {
        const char *p = text, *end = text + len;
        for (size_t t = 0; t < T; t++) {
            chunks[t].ids = &ids;
            chunks[t].text = text;
            chunks[t].begin = p;
            const char *q = (t + 1 == T) ? end : text + len / T * (t + 1);
            if (q < p) q = p;
            if (t + 1 != T) {
                const char *nl = (const char *)std::memchr(q, '\n', (size_t)(end - q));
                q = nl ? nl + 1 : end;
            }
            chunks[t].end = q;
            p = q;
        }
    }
    parallel_for(T, NT, [&](size_t t) {
        if (format == FMT_PAF) parse_chunk_paf(chunks[t]);
        else parse_chunk(chunks[t], format);
    });
    uint64_t line0 = 0;
    for (size_t t = 0; t < T; t++) {
        if (!chunks[t].error.empty())
            return yh::fail(chunks[t].error + " (line " +
                            std::to_string(line0 + chunks[t].error_line) + ")");
        line0 += chunks[t].lines;
    }
    if (ids.overflow.load()) return yh::fail("more than 2^32 - 2 reads");
    ph.mark("parse");

    // ---- global numbering = first appearance in the file = ascending first_pos.  Entries are
    // bucketed by the chunk their first_pos lies in, buckets sorted in parallel.
    const size_t S = ids.n_shards;
    std::vector<uint64_t> shard_base(S + 1, 0);
    for (size_t i = 0; i < S; i++) shard_base[i + 1] = shard_base[i] + ids.shards[i].n_entries;
    const uint64_t R = shard_base[S];
    if (R >= 0xFFFFFFFFull) return yh::fail("more than 2^32 - 2 reads");
    struct Ord {
        uint64_t pos;
        uint32_t flat; // shard_base[shard] + index
    };
    std::vector<uint64_t> chunk_pos(T); // first position of each chunk
    for (size_t t = 0; t < T; t++) chunk_pos[t] = (uint64_t)(chunks[t].begin - text) * 2;
    auto bucket_of = [&](uint64_t pos) {
        return (size_t)(std::upper_bound(chunk_pos.begin(), chunk_pos.end(), pos) - chunk_pos.begin()) - 1;
    };
    // per (shard, bucket) counts -> bucket-major offsets, then scatter
    std::vector<uint32_t> cnt((size_t)S * T, 0);
    parallel_for(S, NT, [&](size_t i) {
        const IdTable::Shard &sh = ids.shards[i];
        for (uint32_t k = 0; k < sh.n_entries; k++)
            cnt[i * T + bucket_of(IdTable::hot(sh, k).first_pos.load(std::memory_order_relaxed))]++;
    });
    std::vector<uint64_t> bucket_off(T + 1, 0);
    std::vector<uint64_t> cell_off((size_t)S * T);
    {
        uint64_t acc = 0;
        for (size_t t = 0; t < T; t++) {
            bucket_off[t] = acc;
            for (size_t i = 0; i < S; i++) {
                cell_off[i * T + t] = acc;
                acc += cnt[i * T + t];
            }
        }
        bucket_off[T] = acc;
    }
    big_vector<Ord> order(R);
    parallel_for(S, NT, [&](size_t i) {
        std::vector<uint64_t> cur(cell_off.begin() + i * T, cell_off.begin() + (i + 1) * T);
        const IdTable::Shard &sh = ids.shards[i];
        for (uint32_t k = 0; k < sh.n_entries; k++) {
            const uint64_t fp = IdTable::hot(sh, k).first_pos.load(std::memory_order_relaxed);
            order[cur[bucket_of(fp)]++] = Ord{fp, (uint32_t)(shard_base[i] + k)};
        }
    });
    parallel_for(T, NT, [&](size_t t) {
        std::sort(order.begin() + bucket_off[t], order.begin() + bucket_off[t + 1],
                  [](const Ord &x, const Ord &y) { return x.pos < y.pos; });
    });
    // flat entry index -> (shard, index) needs the shard: walk the shards' flat ranges
    big_vector<uint32_t> dense(R); // flat entry index -> global read id
    yacrd_csr *c = new yacrd_csr();
    c->lengths.resize(R);
    c->name_off.assign(R + 1, 0);
    auto shard_of_flat = [&](uint32_t flat) {
        return (size_t)(std::upper_bound(shard_base.begin(), shard_base.end(), (uint64_t)flat) -
                        shard_base.begin()) - 1;
    };
    parallel_for(NT, NT, [&](size_t w) {
        for (uint64_t g = R * w / NT; g < R * (w + 1) / NT; g++) {
            const uint32_t flat = order[g].flat;
            const size_t i = shard_of_flat(flat);
            const IdTable::Cold &e = IdTable::cold(ids.shards[i], (uint32_t)(flat - shard_base[i]));
            dense[flat] = (uint32_t)g;
            c->lengths[g] = (uint32_t)e.first_len; // first length seen
            c->name_off[g + 1] = e.nlen;
        }
    });
    for (uint64_t g = 0; g < R; g++) c->name_off[g + 1] += c->name_off[g];
    c->names.resize(c->name_off[R]);
    parallel_for(NT, NT, [&](size_t w) {
        for (uint64_t g = R * w / NT; g < R * (w + 1) / NT; g++) {
            const uint32_t flat = order[g].flat;
            const size_t i = shard_of_flat(flat);
            const uint32_t k = (uint32_t)(flat - shard_base[i]);
            std::memcpy(c->names.data() + c->name_off[g], IdTable::hot(ids.shards[i], k).name,
                        IdTable::cold(ids.shards[i], k).nlen);
        }
    });
    auto global_id = [&](uint32_t handle) {
        return dense[shard_base[handle >> IdTable::kIdxBits] + (handle & ((1u << IdTable::kIdxBits) - 1))];
    };
    ph.mark("number ids");

    // ---- counts -> offsets -> fill ------------------------------------------------------------
    // Each chunk counts the intervals it holds per read in a private array, a pass over the reads
    // turns the counts into every chunk's first write position inside every read (chunk order =
    // file order), and the chunks fill their slices: no atomics, and the intervals of a read come
    // out in line order whatever the thread count.  When T private arrays of R counters would
    // be too big (> 1 GiB), fall back to shared atomic cursors (order inside a read then depends
    // on thread timing; results do not: the sweep sorts).
    parallel_for(T, NT, [&](size_t t) {
        for (Rec &r : chunks[t].recs) {
            r.a = global_id(r.a);
            r.b = global_id(r.b);
        }
    });
    c->offsets.resize(R + 1);
    const bool private_counts = (uint64_t)T * R * sizeof(uint32_t) <= (1ull << 30);
    if (private_counts) {
        std::vector<big_vector<uint32_t>> cnt2(T);
        parallel_for(T, NT, [&](size_t t) {
            cnt2[t].assign(R, 0u);
            uint32_t *k = cnt2[t].data();
            for (const Rec &r : chunks[t].recs) {
                k[r.a]++;
                k[r.b]++;
            }
        });
        // per read: total, and the exclusive prefix over chunks (in place)
        big_vector<uint64_t> tot(R);
        parallel_for(NT, NT, [&](size_t w) {
            for (uint64_t r = R * w / NT; r < R * (w + 1) / NT; r++) {
                uint64_t acc2 = 0;
                for (size_t t = 0; t < T; t++) {
                    const uint32_t n = cnt2[t][r];
                    cnt2[t][r] = (uint32_t)acc2;
                    acc2 += n;
                }
                tot[r] = acc2;
            }
        });
        uint64_t acc = 0;
        for (uint64_t r = 0; r < R; r++) {
            c->offsets[r] = acc;
            acc += tot[r];
        }
        c->offsets[R] = acc;
        // new[] without () leaves the 8 B/interval buffer untouched: a vector would zero it on
        // one thread (tens of ms for 10^7 intervals) before the parallel fill overwrites every word
        c->intervals.reset(new uint32_t[2 * acc + 2]);
        uint32_t *iv = c->intervals.get();
        parallel_for(T, NT, [&](size_t t) {
            uint32_t *k = cnt2[t].data(); // this chunk's next slot inside each read
            const uint64_t *off = c->offsets.data();
            for (const Rec &r : chunks[t].recs) {
                uint64_t p = off[r.a] + k[r.a]++;
                iv[2 * p] = r.sa;
                iv[2 * p + 1] = r.ea;
                p = off[r.b] + k[r.b]++;
                iv[2 * p] = r.sb;
                iv[2 * p + 1] = r.eb;
            }
        });
    } else {
        std::vector<std::atomic<uint64_t>> cur(R + 1);
        parallel_for(NT, NT, [&](size_t w) {
            for (uint64_t r = R * w / NT; r < R * (w + 1) / NT; r++) cur[r].store(0, std::memory_order_relaxed);
        });
        parallel_for(T, NT, [&](size_t t) {
            for (const Rec &r : chunks[t].recs) {
                cur[r.a].fetch_add(1, std::memory_order_relaxed);
                cur[r.b].fetch_add(1, std::memory_order_relaxed);
            }
        });
        uint64_t acc = 0;
        for (uint64_t r = 0; r < R; r++) {
            c->offsets[r] = acc;
            const uint64_t n = cur[r].load(std::memory_order_relaxed);
            cur[r].store(acc, std::memory_order_relaxed);
            acc += n;
        }
        c->offsets[R] = acc;
        c->intervals.reset(new uint32_t[2 * acc + 2]);
        uint32_t *iv = c->intervals.get();
        parallel_for(T, NT, [&](size_t t) {
            for (const Rec &r : chunks[t].recs) {
                uint64_t p = cur[r.a].fetch_add(1, std::memory_order_relaxed);
                iv[2 * p] = r.sa;
                iv[2 * p + 1] = r.ea;
                p = cur[r.b].fetch_add(1, std::memory_order_relaxed);
                iv[2 * p] = r.sb;
                iv[2 * p + 1] = r.eb;
            }
        });
    }
    for (auto &ch : chunks) c->n_records += ch.recs.size();
    ph.mark("csr fill");
    // give the per-chunk arrays back in parallel (hundreds of MB of mappings)
    parallel_for(T, NT, [&](size_t t) { big_vector<Rec>().swap(chunks[t].recs); });
    ph.mark("teardown");
    *out = c;
    return 0;
}

// name -> id index for yacrd_csr_find, built on first use
void build_find_index(yacrd_csr *c)
{
    const size_t R = c->lengths.size();
    size_t cap = 16;
    while (cap < R * 2) cap <<= 1;
    c->table.assign(cap, 0);
    c->mask = cap - 1;
    for (uint32_t id = 0; id < R; id++) {
        const char *p = c->names.data() + c->name_off[id];
        const size_t n = (size_t)(c->name_off[id + 1] - c->name_off[id]);
        uint64_t s2 = yh::hash_bytes(p, n) & c->mask;
        while (c->table[s2]) s2 = (s2 + 1) & c->mask;
        c->table[s2] = id + 1;
    }
}

// src/util.rs:39-55 get_file_type: substring match, .m4/.mhap before .paf
int sniff_format(const std::string &name)
{
    if (name.find(".m4") != std::string::npos || name.find(".mhap") != std::string::npos)
        return FMT_M4;
    if (name.find(".paf") != std::string::npos) return FMT_PAF;
    return FMT_AUTO;
}

} // namespace

extern "C" {

const char *yacrd_host_last_error(void) { return yh::err_slot().c_str(); }

int yacrd_csr_from_memory(const char *text, size_t len, int format, int n_threads, yacrd_csr **out)
{
    if (!out || (!text && len)) return yh::fail("null argument");
    *out = nullptr;
    return build(text, len, format, n_threads, out);
}

int yacrd_csr_from_file(const char *path, int format, int n_threads, yacrd_csr **out)
{
    if (!out || !path) return yh::fail("null argument");
    *out = nullptr;
    if (format == FMT_AUTO) format = sniff_format(path);
    if (format == FMT_AUTO)
        return yh::fail(std::string("Format detection of file ") + path + " failed");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return yh::fail(std::string("Can't open file ") + path + " to read");
    struct stat st;
    if (fstat(fd, &st) != 0) {
        ::close(fd);
        return yh::fail(std::string("Can't stat ") + path);
    }
    unsigned char magic[6] = {0};
    const ssize_t got = ::pread(fd, magic, sizeof magic, 0);
    // compression is sniffed from magic bytes like niffler (src/util.rs:57-70)
    if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        ::close(fd);
        gzFile gz = gzopen(path, "rb");
        if (!gz) return yh::fail(std::string("Can't open gzip file ") + path);
        gzbuffer(gz, 1 << 20);
        std::vector<char> buf;
        size_t used = 0;
        for (;;) {
            if (buf.size() - used < (1u << 20)) buf.resize(buf.size() * 2 + (4u << 20));
            const int n = gzread(gz, buf.data() + used, 1u << 20);
            if (n < 0) {
                gzclose(gz);
                return yh::fail(std::string("gzip read error in ") + path);
            }
            if (n == 0) break;
            used += (size_t)n;
        }
        gzclose(gz);
        return build(buf.data(), used, format, n_threads, out);
    }
    if ((got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h') ||
        (got >= 6 && magic[0] == 0xFD && std::memcmp(magic + 1, "7zXZ", 4) == 0)) {
        ::close(fd);
        return yh::fail(std::string(path) + ": bzip2/xz input is not supported in this build "
                                             "(no bzlib.h / lzma.h in the image); decompress first");
    }
    if (st.st_size == 0) {
        ::close(fd);
        return build("", 0, format, n_threads, out);
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return yh::fail(std::string("mmap failed for ") + path);
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    const int rc = build((const char *)m, (size_t)st.st_size, format, n_threads, out);
    munmap(m, (size_t)st.st_size);
    return rc;
}

int yacrd_csr_get(const yacrd_csr *c, yacrd_csr_view *v)
{
    if (!c || !v) return yh::fail("null argument");
    v->n_reads = c->lengths.size();
    v->n_intervals = c->offsets.empty() ? 0 : c->offsets.back();
    v->n_records = c->n_records;
    v->offsets = c->offsets.data();
    v->intervals = c->intervals.get();
    v->lengths = c->lengths.data();
    v->name_off = c->name_off.data();
    v->names = c->names.data();
    return 0;
}

int64_t yacrd_csr_find(const yacrd_csr *cc, const char *name, size_t n)
{
    if (!cc || cc->lengths.empty()) return -1;
    yacrd_csr *c = const_cast<yacrd_csr *>(cc);
    std::call_once(c->table_once, build_find_index, c);
    const uint64_t h = yh::hash_bytes(name, n);
    uint64_t s = h & c->mask;
    while (uint32_t v = c->table[s]) {
        const uint32_t id = v - 1;
        if (c->name_off[id + 1] - c->name_off[id] == n &&
            std::memcmp(c->names.data() + c->name_off[id], name, n) == 0)
            return id;
        s = (s + 1) & c->mask;
    }
    return -1;
}

void yacrd_csr_free(yacrd_csr *c) { delete c; }

} // extern "C"
