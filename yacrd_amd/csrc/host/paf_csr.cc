// paf_csr.cc — overlap ingest: PAF / M4 text -> overlap records -> the CSR the engine consumes.
//
// Replaces reference Reads2Ovl::init_paf / init_m4 (src/reads2ovl/mod.rs:83-145; column contract
// src/io.rs:23-50) and FullMemory::add_overlap_and_length (src/reads2ovl/fullmemory.rs:82-90):
//   * both reads of every record get an interval (mod.rs:108-109 / :140-141)
//   * a read's length is the FIRST length seen for its id (fullmemory.rs:82-90)
//   * records may carry extra columns (csv `flexible(true)`), empty lines are skipped
//   * a short or non-numeric record is an error (the reference bails, mod.rs:93-97)
// Reads are numbered in first-appearance order.  The input is cut into blocks of whole lines that
// a pool of parse threads takes in turn (a mapped file is sliced in place; a gzip / bzip2 / xz
// stream is decoded by a reader thread that feeds the pool).  All threads intern read ids into one
// shared table whose entries remember where in the file their id came first, so numbering and the
// first-length rule do not depend on the thread count or on timing.
// Two destinations for the parsed records:
//   * yacrd_csr_from_file / _from_memory: kept on the host, grouped into the CSR by the host;
//   * yacrd_ingest_stream: handed to a yacrd_rec_sink buffer by buffer while the parse goes on
//     (the engine's yacrd_stream moves them over PCIe from pinned memory and builds the CSR in HBM).
// Record syntax follows the csv crate the reference reads with (csv 1.3 / csv-core 0.1.11,
// Cargo.lock:241; unpinned by the reference's tests): `"`-quoted fields with `""`
// escapes, records ended by \n, \r\n or \r, integers with an optional `+` or a `0x` prefix.  One
// deviation, loud: a quoted field may not contain a line break.
#include "../../../include/yacrd_host.h"
#include "codec.h"
#include "host_common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
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
    bool streamed = false;             // records went to a sink: no offsets / intervals on the host
    std::vector<uint32_t> handle_map;  // streamed: record handle -> read id (~0u = unused handle)
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

using Rec = yacrd_ovl_rec; // a / b hold IdTable handles until the global numbering exists

// Zero-filled memory for the id table, carved out of 64 MiB anonymous regions with MADV_HUGEPAGE:
// the table is probed at random, and on 4 KiB pages every probe also misses the TLB.
inline uint64_t load64(const char *p)
{
    uint64_t w;
    std::memcpy(&w, p, 8);
    return w;
}
inline uint64_t load32(const char *p)
{
    uint32_t w;
    std::memcpy(&w, p, 4);
    return w;
}

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
// (byte offset of the record * 2 + 0/1 for the first / second id) at which its id was seen and the
// length given there: global numbering = order of those positions (first appearance in the
// file) and a read's length = the first length seen (fullmemory.rs:82-90), whatever the timing.
// What intern() returns is a HANDLE: a 32-bit number unique to the id, dense enough to index an
// array (entry blocks draw their handle ranges from one global counter; at most the last block of
// a shard is partly unused), valid before the numbering exists — records carry handles to the GPU
// while the parse is still running, and a handle -> read id map follows at the end.
struct IdTable {
    static constexpr uint32_t kIdxBits = 22, kBlock0 = 64, kBlocks = 17; // 64 * (2^17 - 1) > 2^22
    static constexpr size_t kShards = 1024; // capacity: kShards * 2^22 ids, far beyond the 2^32 - 2 cap
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
    // One cache line per slot: the id's first bytes and a copy of its first position sit next to the hash, so
    // a hit on an id of up to kInline bytes (a 36-character UUID fits) touches this line and nothing else (with
    // the name and the position behind two more pointers a lookup was three dependent cache misses: 60 % of the
    // parse).  `pos_hint` is never below the entry's true first position (both are lowered under the shard lock,
    // the entry first), so `pos >= pos_hint` is a safe reason to skip the lock.
    static constexpr uint32_t kInline = 36;
    struct alignas(64) Slot {
        uint64_t hash;
        std::atomic<uint32_t> idx1; // entry number + 1, 0 = empty; stored last (release)
        uint32_t nlen;
        std::atomic<uint64_t> pos_hint;
        uint32_t handle;            // what intern() returns for this id
        char name[kInline];
    };
    static_assert(sizeof(Slot) == 64, "one cache line");
    // An index is one word — the slots' address (64-byte aligned) | log2 of their number — so that a shard
    // publishes a bigger one with one store and a lookup finds its first slot from one load.
    struct Index {
        uint32_t mask;
        Slot *slots;
        explicit Index(uint64_t w = 0) : mask(w ? (1u << (w & 63u)) - 1u : 0u), slots(reinterpret_cast<Slot *>(w & ~(uint64_t)63)) {}
        explicit operator bool() const { return slots != nullptr; }
        uint64_t word() const { return slots ? (reinterpret_cast<uint64_t>(slots) | (uint64_t)__builtin_ctz(mask + 1u)) : 0; }
    };
    struct alignas(64) Shard {
        alignas(64) std::atomic<bool> lock{false};
        alignas(64) std::atomic<uint64_t> index{0}; // Index::word()
        Hot *hot[kBlocks] = {};
        Cold *cold[kBlocks] = {};
        uint32_t handle_base[kBlocks] = {}; // handle of the block's first entry
        uint32_t n_entries = 0;
        char *name_cur = nullptr;
        size_t name_left = 0;
    };
    static constexpr int shard_shift = 64 - 10; // shard = hash >> shard_shift
    std::unique_ptr<Shard[]> shards;
    HugePool pool; // indexes, entries and name bytes; released as a whole
    std::atomic<bool> overflow{false};
    std::atomic<uint64_t> next_handle{0};

    IdTable() { shards.reset(new Shard[kShards]); }
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
    static uint32_t handle_of(const Shard &sh, uint32_t idx)
    {
        uint32_t k, at;
        locate(idx, k, at);
        return sh.handle_base[k] + at;
    }
    Index new_index(uint32_t cap)
    {
        Index ix;
        ix.slots = (Slot *)pool.alloc((size_t)cap * sizeof(Slot)); // all-zero = empty slots
        ix.mask = cap - 1;
        return ix;
    }
    // lock held.  Puts entry idx into ix (no duplicates possible).
    static void place(const Index &ix, uint64_t hash, uint32_t nlen, uint32_t idx, const char *name, uint64_t first_pos,
                      uint32_t handle)
    {
        uint32_t s2 = (uint32_t)hash & ix.mask;
        while (ix.slots[s2].idx1.load(std::memory_order_relaxed)) s2 = (s2 + 1) & ix.mask;
        Slot &sl = ix.slots[s2];
        sl.hash = hash;
        sl.nlen = nlen;
        sl.handle = handle;
        sl.pos_hint.store(first_pos, std::memory_order_relaxed);
        std::memcpy(sl.name, name, std::min<size_t>(nlen, kInline));
        sl.idx1.store(idx + 1, std::memory_order_release);
    }
    // a slot's inline bytes against an id (n <= kInline bytes are compared here; both sides are readable for
    // 8 bytes at every offset used: the slot is 64 bytes, an id of >= 8 bytes is read inside itself)
    static bool same_inline(const char *a, const char *b, size_t n)
    {
        if (n > kInline) n = kInline;
        if (n >= 8) {
            size_t i = 0;
            for (; i + 8 < n; i += 8)
                if (load64(a + i) != load64(b + i)) return false;
            return load64(a + n - 8) == load64(b + n - 8);
        }
        return std::memcmp(a, b, n) == 0;
    }
    // returns the entry number, or ~0u when absent; `slot` = where it was found
    static uint32_t probe(const Shard &sh, const Index &ix, const char *p, size_t n, uint64_t h, Slot **slot = nullptr)
    {
        uint32_t s2 = (uint32_t)h & ix.mask;
        for (;;) {
            Slot &sl = ix.slots[s2];
            const uint32_t v = sl.idx1.load(std::memory_order_acquire);
            if (!v) return ~0u;
            if (sl.hash == h && sl.nlen == n && same_inline(sl.name, p, n) &&
                (n <= kInline || std::memcmp(hot(sh, v - 1).name, p, n) == 0)) {
                if (slot) *slot = &sl;
                return v - 1;
            }
            s2 = (s2 + 1) & ix.mask;
        }
    }
    // the cache line a lookup of hash h will look at first (for prefetching ahead of intern())
    const void *first_slot(uint64_t h) const
    {
        const Index ix(shards[(size_t)(h >> shard_shift)].index.load(std::memory_order_acquire));
        return ix ? (const void *)&ix.slots[(uint32_t)h & ix.mask] : nullptr;
    }
    // returns the id's handle
    uint32_t intern(const char *p, size_t n, uint64_t h, uint64_t length, uint64_t pos)
    {
        Shard &sh = shards[(size_t)(h >> shard_shift)];
        const Index ix(sh.index.load(std::memory_order_acquire));
        Slot *hit = nullptr;
        uint32_t idx = ix ? probe(sh, ix, p, n, h, &hit) : ~0u;
        if (idx != ~0u && pos >= hit->pos_hint.load(std::memory_order_relaxed))
            return hit->handle; // the common case: nothing shared is written, the shard's word and one slot are read

        while (sh.lock.exchange(true, std::memory_order_acquire))
            while (sh.lock.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        Index cur(sh.index.load(std::memory_order_relaxed));
        if (!cur) {
            cur = new_index(64);
            sh.index.store(cur.word(), std::memory_order_release);
        }
        Slot *cur_slot = nullptr;
        idx = probe(sh, cur, p, n, h, &cur_slot); // in the CURRENT index (somebody else may have inserted the id meanwhile)
        uint32_t k, at;
        if (idx == ~0u) {
            idx = sh.n_entries;
            if (idx >= (1u << kIdxBits)) {
                overflow.store(true, std::memory_order_relaxed);
                sh.lock.store(false, std::memory_order_release);
                return 0;
            }
            locate(idx, k, at);
            if (!sh.hot[k]) {
                const uint64_t base = next_handle.fetch_add((uint64_t)kBlock0 << k, std::memory_order_relaxed);
                if (base + ((uint64_t)kBlock0 << k) >= 0xFFFFFFF0ull) {
                    overflow.store(true, std::memory_order_relaxed);
                    sh.lock.store(false, std::memory_order_release);
                    return 0;
                }
                sh.handle_base[k] = (uint32_t)base;
                sh.cold[k] = (Cold *)pool.alloc(sizeof(Cold) * ((size_t)kBlock0 << k));
                sh.hot[k] = (Hot *)pool.alloc(sizeof(Hot) * ((size_t)kBlock0 << k));
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
            if ((uint64_t)(idx + 2) * 2 > (uint64_t)cur.mask + 1) { // keep the load below 1/2
                const Index bigger = new_index((cur.mask + 1) * 2);
                for (uint32_t k2 = 0; k2 <= idx; k2++) {
                    const Cold &c2 = cold(sh, k2);
                    const Hot &h2 = hot(sh, k2);
                    place(bigger, c2.hash, c2.nlen, k2, h2.name, h2.first_pos.load(std::memory_order_relaxed),
                          handle_of(sh, k2));
                }
                sh.index.store(bigger.word(), std::memory_order_release);
            } else {
                place(cur, h, (uint32_t)n, idx, e->name, pos, sh.handle_base[k] + at);
            }
        } else {
            locate(idx, k, at);
            Hot &e = sh.hot[k][at];
            if (pos < e.first_pos.load(std::memory_order_relaxed)) {
                sh.cold[k][at].first_len = length;
                e.first_pos.store(pos, std::memory_order_relaxed);
            }
            // (the slot of the current index follows the entry; slots of replaced indexes keep their larger
            // hints and send their readers here)
            if (pos < cur_slot->pos_hint.load(std::memory_order_relaxed)) cur_slot->pos_hint.store(pos, std::memory_order_relaxed);
        }
        const uint32_t handle = sh.handle_base[k] + at;
        sh.lock.store(false, std::memory_order_release);
        return handle;
    }
};

// ---- where a parse thread puts its records ----------------------------------------------------
struct RecOut {
    Rec *cur = nullptr, *lim = nullptr;
    std::string error;
    virtual ~RecOut() {}
    virtual bool more() = 0; // cur == lim: make room (false: `error` says why)
};

// host destination: 4 MiB segments of records that stay where they are
struct SegOut final : RecOut {
    struct Seg {
        Rec *p;
        size_t n;
    };
    static constexpr size_t kSegRecs = ((size_t)4 << 20) / sizeof(Rec); // 4 MiB: two huge pages
    std::vector<Seg> segs;
    bool more() override
    {
        close_segment();
        Rec *p = HugeAlloc<Rec>().allocate(kSegRecs);
        segs.push_back(Seg{p, 0});
        cur = p;
        lim = p + kSegRecs;
        return true;
    }
    void close_segment()
    {
        if (!segs.empty() && cur) segs.back().n = (size_t)(cur - segs.back().p);
    }
    void release()
    {
        for (Seg &s : segs) HugeAlloc<Rec>().deallocate(s.p, kSegRecs);
        segs.clear();
        cur = lim = nullptr;
    }
    ~SegOut() override { release(); }
};

// streaming destination: buffers borrowed from the sink (pinned memory on its way over PCIe)
struct SinkOut final : RecOut {
    const yacrd_rec_sink *sink = nullptr;
    Rec *base = nullptr;
    uint64_t committed = 0;
    bool more() override
    {
        if (!flush()) return false;
        uint64_t cap = 0;
        Rec *p = nullptr;
        if (sink->acquire(sink->ctx, &p, &cap) != 0 || !p || cap == 0) {
            error = "the record sink refused to hand out a buffer";
            return false;
        }
        base = cur = p;
        lim = p + cap;
        return true;
    }
    bool flush()
    {
        if (!base) return true;
        const uint64_t n = (uint64_t)(cur - base);
        Rec *b = base;
        base = cur = lim = nullptr;
        if (sink->commit(sink->ctx, b, n) != 0) {
            error = "the record sink failed to take a buffer";
            return false;
        }
        committed += n;
        return true;
    }
};

// ---- field syntax --------------------------------------------------------------------------------
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
// the csv crate's integer fields: `0x` + hex digits (from_str_radix) or FromStr (decimal, optional +)
inline bool parse_csv_u64(const char *p, const char *e, uint64_t &out)
{
    if (e - p >= 2 && p[0] == '0' && p[1] == 'x') {
        p += 2;
        if (p < e && *p == '+') p++;
        if (p == e) return false;
        uint64_t v = 0;
        for (; p < e; p++) {
            const unsigned char c = (unsigned char)*p;
            unsigned d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
            else return false;
            if (v >> 60) return false;
            v = (v << 4) | d;
        }
        out = v;
        return true;
    }
    return parse_u64(p, e, out);
}
inline bool parse_csv_u32(const char *p, const char *e, uint32_t &out)
{
    uint64_t v;
    if (!parse_csv_u64(p, e, v) || v > 0xFFFFFFFFull) return false;
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
// Rust's f64::from_str: decimal digits with optional sign, point and exponent, or inf / infinity /
// nan in any case.  strtod also takes hex floats, nan(...) and leading blanks: excluded here.
inline bool is_rust_f64(const char *p, const char *e)
{
    if (p == e) return false;
    for (const char *q = p; q < e; q++)
        if (*q == 'x' || *q == 'X' || *q == '(' || *q == ' ' || *q == '\t' || *q == 'p' || *q == 'P') return false;
    const std::string s(p, e);
    char *endp = nullptr;
    errno = 0;
    (void)std::strtod(s.c_str(), &endp);
    return endp && *endp == '\0';
}

// 64-bit hash of an id, eight bytes at a time (the byte-at-a-time FNV chain cost ~40 cycles per
// id: two per line).  Must agree with hash_id_scan below.
inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}
// whole words, then the LAST eight bytes again (overlapping the previous word when n % 8 != 0);
// short ids take two overlapping 4-byte loads or three single bytes: no byte loops, no calls
inline uint64_t hash_id(const char *p, size_t n)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ ((uint64_t)n * 0xff51afd7ed558ccdull);
    if (n >= 8) {
        size_t i = 0;
        for (; i + 8 < n; i += 8) {
            h = (h ^ load64(p + i)) * 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
        h = (h ^ load64(p + n - 8)) * 0xc4ceb9fe1a85ec53ull;
    } else if (n >= 4) {
        h = (h ^ (load32(p) | (load32(p + n - 4) << 32))) * 0xc4ceb9fe1a85ec53ull;
    } else if (n > 0) {
        h = (h ^ ((uint64_t)(unsigned char)p[0] | ((uint64_t)(unsigned char)p[n >> 1] << 8) |
                  ((uint64_t)(unsigned char)p[n - 1] << 16))) * 0xc4ceb9fe1a85ec53ull;
    }
    return mix64(h);
}

struct Fields { // one overlap record, syntax checked
    const char *ida, *idb;
    size_t na, nb;
    uint64_t la, lb;
    uint32_t sa, ea, sb, eb;
};

// ---- PAF fast path: one forward scan per line ----------------------------------------------------
// Accepts the plain form of a record (src/io.rs:23-34): 9 leading tab-separated fields, decimal
// u64 lengths and u32 positions with an optional '+', one-character strand, anything after
// ignored.  Only for lines that hold neither a '"' nor a '\r' (scan_line says so): a quote may open a
// quoted field — also in the ignored columns, where it may swallow delimiters — and a lone \r ends a
// csv record.  Whatever it does not accept gets a second look by parse_record_general (0x integers ...).
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
inline bool scan_id(const char *&p, const char *le, const char *&b, size_t &n)
{
    b = p;
    if (p < le && *p == '"') return false; // quoted field: the general parser's business
    const char *t = p;
    while (t < le && *t != '\t') t++; // ids are short: a byte loop beats a memchr call
    if (t == le) return false; // an id must be followed by more fields
    n = (size_t)(t - p);
    p = t + 1;
    return true;
}
inline bool parse_paf_fast(const char *p, const char *le, Fields &f)
{
    const char *q = p;
    uint64_t sa, ea, sb, eb;
    if (!(scan_id(q, le, f.ida, f.na) && scan_uint(q, le, ~0ull, f.la, false) &&
          scan_uint(q, le, 0xFFFFFFFFull, sa, false) && scan_uint(q, le, 0xFFFFFFFFull, ea, false)))
        return false;
    if (le - q >= 2 && (unsigned char)q[0] < 0x80 && q[0] != '\t' && q[1] == '\t') {
        q += 2; // strand: one ASCII character and a tab, the usual case
    } else { // exactly one UTF-8 scalar, then a tab
        const char *t = (const char *)std::memchr(q, '\t', (size_t)(le - q));
        if (!t || !is_one_char(q, t) || *q == '"') return false;
        q = t + 1;
    }
    if (!(scan_id(q, le, f.idb, f.nb) && scan_uint(q, le, ~0ull, f.lb, false) &&
          scan_uint(q, le, 0xFFFFFFFFull, sb, false) && scan_uint(q, le, 0xFFFFFFFFull, eb, true)))
        return false;
    f.sa = (uint32_t)sa;
    f.ea = (uint32_t)ea;
    f.sb = (uint32_t)sb;
    f.eb = (uint32_t)eb;
    return true;
}

// ---- one line: where it ends, and whether it holds a character the fast path must not see ('"', '\r').
// Sixteen bytes at a time (SSE2 is part of x86-64): one pass instead of memchr + a second look at the
// ignored columns.
inline const char *scan_line(const char *p, const char *end, bool &special)
{
    special = false;
    const __m128i nl = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r'), qt = _mm_set1_epi8('"');
    unsigned sp = 0;
    for (; p + 16 <= end; p += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
        const unsigned m_nl = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, nl));
        const unsigned m_sp = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, cr), _mm_cmpeq_epi8(v, qt)));
        if (m_nl) {
            const unsigned at = (unsigned)__builtin_ctz(m_nl);
            special = (sp | (m_sp & ((1u << at) - 1u))) != 0;
            return p + at;
        }
        sp |= m_sp;
    }
    for (; p < end; p++) {
        if (*p == '\n') break;
        sp |= (*p == '\r' || *p == '"') ? 1u : 0u;
    }
    special = sp != 0;
    return p;
}

// ---- general record parser: csv-crate field syntax, PAF and M4 --------------------------------
// Scans ONE record from p: fields are split at `delim`, the record ends at an unquoted '\r' (csv's
// default terminator takes \r, \n and \r\n alike; \n never gets here) or at le; `rec_end` is where
// it stopped.  A field that starts with '"' is quoted: it runs to the closing quote, `""` inside
// stands for one quote, delimiters and \r inside are data, and whatever follows the closing quote
// up to the delimiter is appended as is (csv-core's DFA: StartField / InField / InQuotedField /
// InDoubleEscapedQuote).  A quote inside an unquoted field is data.  The first `need` fields are
// returned: unquoted ones point into the line, quoted ones into `scratch` (reserved up front, so
// it never reallocates under them).
bool split_fields(const char *p, const char *le, char delim, int need, const char **fb, const char **fe,
                  std::string &scratch, const char *&rec_end)
{
    scratch.clear();
    scratch.reserve((size_t)(le - p) + 1);
    int nf = 0;
    const char *q = p;
    for (;;) {
        if (q < le && *q == '"') {
            const size_t s0 = scratch.size();
            q++;
            bool closed = false;
            while (q < le) {
                if (*q == '"') {
                    if (q + 1 < le && q[1] == '"') {
                        scratch.push_back('"');
                        q += 2;
                        continue;
                    }
                    q++;
                    closed = true;
                    break;
                }
                scratch.push_back(*q++);
            }
            if (!closed) return false; // the quote never closes on this line
            while (q < le && *q != delim && *q != '\r') scratch.push_back(*q++);
            if (nf < need) {
                fb[nf] = scratch.data() + s0;
                fe[nf] = scratch.data() + scratch.size();
            }
        } else {
            const char *d = q;
            while (d < le && *d != delim && *d != '\r') d++;
            if (nf < need) {
                fb[nf] = q;
                fe[nf] = d;
            }
            q = d;
        }
        nf++;
        if (q >= le || *q == '\r') break;
        q++; // the delimiter
    }
    rec_end = q;
    return nf >= need;
}
bool parse_record_general(const char *p, const char *le, int format, Fields &f, std::string &scratch,
                          const char *&rec_end)
{
    const char *fb[12], *fe[12];
    rec_end = le;
    if (format == FMT_PAF) { // src/io.rs:23-34
        if (!split_fields(p, le, '\t', 9, fb, fe, scratch, rec_end)) return false;
        f.ida = fb[0], f.na = (size_t)(fe[0] - fb[0]);
        f.idb = fb[5], f.nb = (size_t)(fe[5] - fb[5]);
        return parse_csv_u64(fb[1], fe[1], f.la) && parse_csv_u32(fb[2], fe[2], f.sa) &&
               parse_csv_u32(fb[3], fe[3], f.ea) && is_one_char(fb[4], fe[4]) &&
               parse_csv_u64(fb[6], fe[6], f.lb) && parse_csv_u32(fb[7], fe[7], f.sb) &&
               parse_csv_u32(fb[8], fe[8], f.eb);
    }
    // src/io.rs:36-50: a b err shared strand_a begin_a end_a len_a strand_b begin_b end_b len_b
    if (!split_fields(p, le, ' ', 12, fb, fe, scratch, rec_end)) return false;
    uint64_t shared;
    f.ida = fb[0], f.na = (size_t)(fe[0] - fb[0]);
    f.idb = fb[1], f.nb = (size_t)(fe[1] - fb[1]);
    return is_rust_f64(fb[2], fe[2]) && parse_csv_u64(fb[3], fe[3], shared) && is_one_char(fb[4], fe[4]) &&
           parse_csv_u32(fb[5], fe[5], f.sa) && parse_csv_u32(fb[6], fe[6], f.ea) &&
           parse_csv_u64(fb[7], fe[7], f.la) && is_one_char(fb[8], fe[8]) &&
           parse_csv_u32(fb[9], fe[9], f.sb) && parse_csv_u32(fb[10], fe[10], f.eb) &&
           parse_csv_u64(fb[11], fe[11], f.lb);
}

// ---- one block of whole lines --------------------------------------------------------------------
struct BlockResult {
    uint64_t lines = 0;      // '\n'-terminated lines seen (for error messages)
    uint64_t records = 0;
    std::string error;
    uint64_t error_line = 0; // 1-based within the block
};

void parse_block(const char *begin, const char *end, uint64_t pos0, int format, IdTable &ids,
                 RecOut &out, BlockResult &res)
{
    std::string scratch;
    const char *p = begin;
    // Records are interned in batches: the fast path only tokenises a line, hashes its two ids and asks for the
    // cache lines their lookups will read first; the lookups of a batch then run over lines that are already on
    // their way (one at a time each lookup waited for its own miss: 60 % of the parse).
    constexpr int kBatch = 16;
    struct Pending {
        Fields f;
        uint64_t ha, hb, pos;
        uint64_t line;
    } pend[kBatch];
    int n_pend = 0;
    auto emit_hashed = [&](const Fields &f, uint64_t ha, uint64_t hb, uint64_t pos) -> bool {
        if (f.la > 0xFFFFFFFFull || f.lb > 0xFFFFFFFFull) {
            res.error = "read length >= 2^32 is not supported by the engine";
            return false;
        }
        if (out.cur == out.lim && !out.more()) {
            res.error = out.error;
            return false;
        }
        Rec &r = *out.cur++;
        r.sa = f.sa;
        r.ea = f.ea;
        r.sb = f.sb;
        r.eb = f.eb;
        r.a = ids.intern(f.ida, f.na, ha, f.la, pos);
        r.b = ids.intern(f.idb, f.nb, hb, f.lb, pos + 1);
        res.records++;
        return true;
    };
    auto flush = [&]() -> bool { // the pending records, in line order
        for (int i = 0; i < n_pend; i++) {
            if (!emit_hashed(pend[i].f, pend[i].ha, pend[i].hb, pend[i].pos)) {
                res.error_line = pend[i].line;
                n_pend = 0;
                return false;
            }
        }
        n_pend = 0;
        return true;
    };
    auto defer = [&](const Fields &f, const char *rec_start) -> bool { // ids point into the block: they stay valid
        Pending &q = pend[n_pend];
        q.f = f;
        q.ha = hash_id(f.ida, f.na);
        q.hb = hash_id(f.idb, f.nb);
        q.pos = (pos0 + (uint64_t)(rec_start - begin)) * 2;
        q.line = res.lines;
        if (const void *l = ids.first_slot(q.ha)) __builtin_prefetch(l, 0, 1);
        if (const void *l = ids.first_slot(q.hb)) __builtin_prefetch(l, 0, 1);
        return ++n_pend < kBatch || flush();
    };
    auto emit = [&](const Fields &f, const char *rec_start) -> bool { // (ids that live in `scratch`: at once, in order)
        if (!flush()) return false;
        return emit_hashed(f, hash_id(f.ida, f.na), hash_id(f.idb, f.nb), (pos0 + (uint64_t)(rec_start - begin)) * 2);
    };
    const char *what = format == FMT_PAF ? "Reading of the file in paf format failed"
                                         : "Reading of the file in m4 format failed";
    while (p < end) {
        bool special;
        const char *eol = scan_line(p, end, special);
        const char *le = eol;
        if (le > p && le[-1] == '\r') { // CRLF: the line's own terminator does not count
            le--;
            if (special) special = std::memchr(p, '\r', (size_t)(le - p)) != nullptr || std::memchr(p, '"', (size_t)(le - p)) != nullptr;
        }
        res.lines++;
        if (le == p) { // csv skips empty lines
            p = eol + 1;
            continue;
        }
        Fields f;
        if (format == FMT_PAF && !special && parse_paf_fast(p, le, f)) {
            if (!defer(f, p)) {
                if (!res.error_line) res.error_line = res.lines;
                return;
            }
            p = eol + 1;
            continue;
        }
        // the general parser; a lone \r also ends a record (csv's default terminator)
        const char *s = p;
        while (s < le) {
            if (*s == '\r') { // a run of terminators: no record
                s++;
                continue;
            }
            const char *rec_end = le;
            if (!parse_record_general(s, le, format, f, scratch, rec_end)) {
                res.error = what;
                if (std::memchr(s, '"', (size_t)(le - s)))
                    res.error += " (the record holds a '\"': quoted fields follow the csv crate's rules and "
                                 "may not span lines)";
                res.error_line = res.lines;
                return;
            }
            if (!emit(f, s)) {
                if (!res.error_line) res.error_line = res.lines;
                return;
            }
            s = rec_end;
        }
        p = eol + 1;
    }
    if (!flush() && !res.error_line) res.error_line = res.lines;
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

// ---- where the text comes from -----------------------------------------------------------------
struct TextBlock {
    const char *b = nullptr, *e = nullptr;
    uint64_t pos0 = 0;  // byte offset of b in the whole (decoded) input
    size_t index = 0;   // file order
    std::unique_ptr<char[]> owner; // decoded blocks own their bytes
};
struct BlockSource {
    virtual ~BlockSource() {}
    virtual bool next(TextBlock &blk) = 0; // thread-safe; false at the end or after an error
    virtual void abort() {}                // a consumer failed: stop producing, wake everybody
    std::string error;                     // set before next() returns false for good
};

// text already in memory (a mapped file): blocks are slices, cut at line starts
struct MemSource final : BlockSource {
    const char *text;
    size_t len, block;
    std::atomic<size_t> turn{0};
    MemSource(const char *t, size_t n, size_t b) : text(t), len(n), block(std::max<size_t>(b, 1)) {}
    size_t line_start_at_or_after(size_t x) const
    {
        if (x == 0) return 0;
        if (x >= len) return len;
        const char *nl = (const char *)std::memchr(text + x - 1, '\n', len - (x - 1));
        return nl ? (size_t)(nl - text) + 1 : len;
    }
    bool next(TextBlock &blk) override
    {
        for (;;) {
            const size_t i = turn.fetch_add(1, std::memory_order_relaxed);
            if (i > len / block) return false; // (i * block cannot overflow below this bound)
            const size_t b = line_start_at_or_after(i * block), e = line_start_at_or_after((i + 1) * block);
            if (b >= len) return false;
            if (b == e) continue; // a line longer than a block swallowed this slot
            blk.b = text + b;
            blk.e = text + e;
            blk.pos0 = b;
            blk.index = i;
            blk.owner.reset();
            return true;
        }
    }
};

// a regular file: every parse thread preads the block it takes into a buffer of its own.  Mapping
// the file instead costs a minor fault per 4 KiB page, and the threads' faults serialise on the
// process's mmap lock (the parse stopped scaling at 8-16 threads); pread copies out of the page
// cache at memcpy speed with no shared lock.  Block i holds the lines that START in
// [i * block, (i + 1) * block): it reads from one byte earlier (is the first byte a line start?) to
// the end of the line that straddles its upper edge.
struct FileSource final : BlockSource {
    const int fd;
    const size_t len, block;
    std::atomic<size_t> turn{0};
    std::mutex mu;
    FileSource(int f, size_t n, size_t b) : fd(f), len(n), block(std::max<size_t>(b, 1)) {}
    bool read_at(char *dst, size_t n, size_t off)
    {
        size_t got = 0;
        while (got < n) {
            const ssize_t k = ::pread(fd, dst + got, n - got, (off_t)(off + got));
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return false;
            got += (size_t)k;
        }
        return true;
    }
    bool next(TextBlock &blk) override
    {
        for (;;) {
            const size_t i = turn.fetch_add(1, std::memory_order_relaxed);
            if (i > len / block) return false;
            const size_t lo = i * block, hi = std::min(len, lo + block);
            if (lo >= len) return false;
            const size_t from = lo ? lo - 1 : 0;
            size_t extra = 64u << 10, have = 0, cap = (hi - from) + extra;
            std::unique_ptr<char[]> buf(new (std::nothrow) char[cap]);
            bool ok = buf != nullptr;
            size_t end_rel = 0; // one past the block's last byte, relative to `from`
            while (ok) {
                const size_t want = std::min(len - from, cap);
                ok = read_at(buf.get() + have, want - have, from + have);
                if (!ok) break;
                have = want;
                if (hi >= len) {
                    end_rel = have;
                    break;
                }
                // the line that covers byte hi - 1 ends the block
                const char *nl = (const char *)std::memchr(buf.get() + (hi - 1 - from), '\n', have - (hi - 1 - from));
                if (nl) {
                    end_rel = (size_t)(nl - buf.get()) + 1;
                    break;
                }
                if (from + have >= len) {
                    end_rel = have;
                    break;
                }
                cap = (hi - from) + (extra *= 4); // a very long line: read further
                std::unique_ptr<char[]> nb(new (std::nothrow) char[cap]);
                if (!nb) {
                    ok = false;
                    break;
                }
                std::memcpy(nb.get(), buf.get(), have);
                buf.swap(nb);
            }
            if (!ok) {
                std::lock_guard<std::mutex> g(mu);
                if (error.empty()) error = "read error in the overlap file";
                return false;
            }
            size_t begin_rel = 0; // first line start at or after lo
            if (lo) {
                const char *nl = (const char *)std::memchr(buf.get(), '\n', end_rel);
                if (!nl) continue; // a line that started in an earlier block covers this one entirely
                begin_rel = (size_t)(nl - buf.get()) + 1;
            }
            if (begin_rel >= end_rel) continue;
            blk.b = buf.get() + begin_rel;
            blk.e = buf.get() + end_rel;
            blk.pos0 = from + begin_rel;
            blk.index = i;
            blk.owner = std::move(buf);
            return true;
        }
    }
};

// a compressed (or otherwise sequential) stream: one reader thread decodes blocks of whole lines
// into a bounded queue
struct DecodeSource final : BlockSource {
    yh::InStream &in;
    const size_t block, depth;
    std::mutex mu;
    std::condition_variable cv_full, cv_empty;
    std::deque<TextBlock> q;
    bool done = false, stop = false;
    std::thread reader;
    DecodeSource(yh::InStream &s, size_t block_bytes, size_t queue_depth)
        : in(s), block(block_bytes), depth(std::max<size_t>(2, queue_depth))
    {
        reader = std::thread([this] { produce(); });
    }
    ~DecodeSource() override
    {
        abort();
        if (reader.joinable()) reader.join();
    }
    void abort() override
    {
        std::lock_guard<std::mutex> g(mu);
        stop = true;
        cv_full.notify_all();
        cv_empty.notify_all();
    }
    void produce()
    {
        std::string carry; // the partial last line of the previous block
        uint64_t pos = 0;
        size_t index = 0;
        bool eof = false;
        std::string fail_msg;
        while (!eof) {
            size_t cap = block + carry.size() + 1;
            std::unique_ptr<char[]> buf(new (std::nothrow) char[cap]);
            if (!buf) {
                fail_msg = "out of memory while decoding the input";
                break;
            }
            size_t n = carry.size();
            std::memcpy(buf.get(), carry.data(), n);
            carry.clear();
            size_t cut = 0; // one past the last '\n'
            for (;;) {
                while (n < cap - 1) {
                    const long got = in.read(buf.get() + n, cap - 1 - n);
                    if (got < 0) {
                        fail_msg = yh::err_slot();
                        break;
                    }
                    if (got == 0) {
                        eof = true;
                        break;
                    }
                    n += (size_t)got;
                }
                if (!fail_msg.empty()) break;
                if (eof) {
                    cut = n;
                    break;
                }
                const char *base = buf.get();
                const void *nl = memrchr(base, '\n', n);
                if (nl) {
                    cut = (size_t)((const char *)nl - base) + 1;
                    break;
                }
                // a line longer than the block: keep reading into a bigger buffer
                const size_t bigger = cap * 2;
                std::unique_ptr<char[]> nb(new (std::nothrow) char[bigger]);
                if (!nb) {
                    fail_msg = "out of memory while decoding the input";
                    break;
                }
                std::memcpy(nb.get(), buf.get(), n);
                buf.swap(nb);
                cap = bigger;
            }
            if (!fail_msg.empty()) break;
            carry.assign(buf.get() + cut, n - cut);
            if (cut == 0) continue;
            TextBlock blk;
            blk.b = buf.get();
            blk.e = buf.get() + cut;
            blk.pos0 = pos;
            blk.index = index++;
            blk.owner = std::move(buf);
            pos += cut;
            std::unique_lock<std::mutex> g(mu);
            cv_full.wait(g, [&] { return stop || q.size() < depth; });
            if (stop) return;
            q.push_back(std::move(blk));
            cv_empty.notify_one();
        }
        std::lock_guard<std::mutex> g(mu);
        if (!fail_msg.empty()) error = fail_msg;
        done = true;
        cv_empty.notify_all();
    }
    bool next(TextBlock &blk) override
    {
        std::unique_lock<std::mutex> g(mu);
        cv_empty.wait(g, [&] { return stop || done || !q.empty(); });
        if (stop || q.empty()) return false;
        blk = std::move(q.front());
        q.pop_front();
        cv_full.notify_one();
        return true;
    }
};

// ---- parse every block of a source on NT threads -------------------------------------------------
struct Parsed {
    struct Block {
        size_t index;
        uint64_t lines, records;
        std::unique_ptr<SegOut> recs; // host destination only
    };
    std::vector<Block> blocks; // sorted by index on return
    uint64_t n_records = 0, text_bytes = 0;
};

int parse_all(BlockSource &src, int format, size_t NT, const yacrd_rec_sink *sink, IdTable &ids,
              Parsed &out)
{
    std::mutex mu;
    std::string first_error;
    size_t error_index = ~(size_t)0;
    uint64_t error_line = 0;
    std::atomic<bool> failed{false};
    auto worker = [&]() {
        SinkOut so;
        so.sink = sink;
        TextBlock blk;
        while (!failed.load(std::memory_order_relaxed) && src.next(blk)) {
            BlockResult res;
            std::unique_ptr<SegOut> seg;
            if (sink) parse_block(blk.b, blk.e, blk.pos0, format, ids, so, res);
            else {
                seg.reset(new SegOut());
                parse_block(blk.b, blk.e, blk.pos0, format, ids, *seg, res);
                seg->close_segment();
            }
            std::lock_guard<std::mutex> g(mu);
            out.text_bytes = std::max<uint64_t>(out.text_bytes, blk.pos0 + (uint64_t)(blk.e - blk.b));
            if (!res.error.empty()) {
                if (blk.index < error_index) { // report the earliest failing block
                    error_index = blk.index;
                    first_error = res.error;
                    error_line = res.error_line;
                }
                failed.store(true, std::memory_order_relaxed);
                src.abort();
            }
            out.blocks.push_back(Parsed::Block{blk.index, res.lines, res.records, std::move(seg)});
            blk.owner.reset();
        }
        if (sink && !so.flush()) {
            std::lock_guard<std::mutex> g(mu);
            if (first_error.empty()) first_error = so.error;
            failed.store(true, std::memory_order_relaxed);
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < NT; i++) th.emplace_back(worker);
        worker();
        for (auto &x : th) x.join();
    }
    std::sort(out.blocks.begin(), out.blocks.end(),
              [](const Parsed::Block &a, const Parsed::Block &b) { return a.index < b.index; });
    if (!src.error.empty() && first_error.empty()) return yh::fail(src.error);
    if (!first_error.empty()) {
        uint64_t line0 = 0; // lines of the blocks in front of the failing one (all of them were parsed
                            // unless they failed too, in which case the earliest one is reported)
        for (const auto &b : out.blocks)
            if (b.index < error_index) line0 += b.lines;
        if (error_index == ~(size_t)0) return yh::fail(first_error);
        return yh::fail(first_error + " (line " + std::to_string(line0 + error_line) + ")");
    }
    if (ids.overflow.load())
        return yh::fail("too many reads: the id table holds at most 2^32 - 2 of them (2^22 per shard of 1024)");
    for (const auto &b : out.blocks) out.n_records += b.records;
    return 0;
}

// ---- global numbering = first appearance in the file = ascending first_pos -------------------------
// Fills lengths / names of `c` and `dense` (handle -> read id, ~0u for unused handles).
int number_ids(IdTable &ids, uint64_t text_bytes, size_t NT, yacrd_csr *c, big_vector<uint32_t> &dense)
{
    const size_t S = IdTable::kShards;
    std::vector<uint64_t> shard_base(S + 1, 0);
    for (size_t i = 0; i < S; i++) shard_base[i + 1] = shard_base[i] + ids.shards[i].n_entries;
    const uint64_t R = shard_base[S];
    if (R >= 0xFFFFFFFFull) return yh::fail("too many reads: at most 2^32 - 2");
    struct Ord {
        uint64_t pos;
        uint32_t shard, idx;
    };
    // entries are bucketed by the slice of the text their first position lies in; buckets are
    // sorted in parallel
    const size_t NB = std::max<size_t>(1, std::min<size_t>(NT * 4, (size_t)(R / 4096 + 1)));
    const uint64_t span = 2 * text_bytes + 2;
    auto bucket_of = [&](uint64_t pos) { return (size_t)((unsigned __int128)pos * NB / span); };
    std::vector<uint32_t> cnt(S * NB, 0);
    parallel_for(S, NT, [&](size_t i) {
        const IdTable::Shard &sh = ids.shards[i];
        for (uint32_t k = 0; k < sh.n_entries; k++)
            cnt[i * NB + bucket_of(IdTable::hot(sh, k).first_pos.load(std::memory_order_relaxed))]++;
    });
    std::vector<uint64_t> bucket_off(NB + 1, 0), cell_off(S * NB);
    {
        uint64_t acc = 0;
        for (size_t t = 0; t < NB; t++) {
            bucket_off[t] = acc;
            for (size_t i = 0; i < S; i++) {
                cell_off[i * NB + t] = acc;
                acc += cnt[i * NB + t];
            }
        }
        bucket_off[NB] = acc;
    }
    big_vector<Ord> order(R);
    parallel_for(S, NT, [&](size_t i) {
        std::vector<uint64_t> cur(cell_off.begin() + i * NB, cell_off.begin() + (i + 1) * NB);
        const IdTable::Shard &sh = ids.shards[i];
        for (uint32_t k = 0; k < sh.n_entries; k++) {
            const uint64_t fp = IdTable::hot(sh, k).first_pos.load(std::memory_order_relaxed);
            order[cur[bucket_of(fp)]++] = Ord{fp, (uint32_t)i, k};
        }
    });
    parallel_for(NB, NT, [&](size_t t) {
        std::sort(order.begin() + bucket_off[t], order.begin() + bucket_off[t + 1],
                  [](const Ord &x, const Ord &y) { return x.pos < y.pos; });
    });
    const uint64_t n_handles = ids.next_handle.load();
    dense.assign(n_handles, ~0u);
    c->lengths.resize(R);
    c->name_off.assign(R + 1, 0);
    parallel_for(NT, NT, [&](size_t w) {
        for (uint64_t g = R * w / NT; g < R * (w + 1) / NT; g++) {
            const IdTable::Shard &sh = ids.shards[order[g].shard];
            const IdTable::Cold &e = IdTable::cold(sh, order[g].idx);
            dense[IdTable::handle_of(sh, order[g].idx)] = (uint32_t)g;
            c->lengths[g] = (uint32_t)e.first_len; // first length seen
            c->name_off[g + 1] = e.nlen;
        }
    });
    for (uint64_t g = 0; g < R; g++) c->name_off[g + 1] += c->name_off[g];
    c->names.resize(c->name_off[R]);
    parallel_for(NT, NT, [&](size_t w) {
        for (uint64_t g = R * w / NT; g < R * (w + 1) / NT; g++) {
            const IdTable::Shard &sh = ids.shards[order[g].shard];
            std::memcpy(c->names.data() + c->name_off[g], IdTable::hot(sh, order[g].idx).name,
                        IdTable::cold(sh, order[g].idx).nlen);
        }
    });
    return 0;
}

// ---- host destination: records -> CSR -----------------------------------------------------------
// Blocks are grouped into G runs of consecutive blocks (file order).  Each group counts the
// intervals it holds per read in a private array, a pass over the reads turns the counts into
// every group's first write position inside every read, and the groups fill their slices: no
// atomics, and the intervals of a read come out in line order whatever the thread count.  When G
// private arrays of R counters would be too big (> 1 GiB), fall back to shared atomic cursors
// (order inside a read then depends on thread timing; results do not: the sweep sorts).
void fill_csr(Parsed &ps, const big_vector<uint32_t> &dense, size_t NT, yacrd_csr *c)
{
    const uint64_t R = c->lengths.size();
    const size_t nb = ps.blocks.size();
    const size_t G = std::max<size_t>(1, std::min(NT, nb));
    auto for_group = [&](size_t g, auto fn) {
        for (size_t b = nb * g / G; b < nb * (g + 1) / G; b++)
            for (const SegOut::Seg &s : ps.blocks[b].recs->segs)
                for (size_t i = 0; i < s.n; i++) fn(s.p[i]);
    };
    parallel_for(nb, NT, [&](size_t b) { // handles -> read ids, in place
        for (const SegOut::Seg &s : ps.blocks[b].recs->segs)
            for (size_t i = 0; i < s.n; i++) {
                s.p[i].a = dense[s.p[i].a];
                s.p[i].b = dense[s.p[i].b];
            }
    });
    c->offsets.resize(R + 1);
    const bool private_counts = (uint64_t)G * R * sizeof(uint32_t) <= (1ull << 30);
    if (private_counts) {
        std::vector<big_vector<uint32_t>> cnt2(G);
        parallel_for(G, NT, [&](size_t g) {
            cnt2[g].assign(R, 0u);
            uint32_t *k = cnt2[g].data();
            for_group(g, [&](const Rec &r) {
                k[r.a]++;
                k[r.b]++;
            });
        });
        big_vector<uint64_t> tot(R);
        parallel_for(NT, NT, [&](size_t w) { // per read: total, and the exclusive prefix over groups
            for (uint64_t r = R * w / NT; r < R * (w + 1) / NT; r++) {
                uint64_t acc2 = 0;
                for (size_t g = 0; g < G; g++) {
                    const uint32_t n = cnt2[g][r];
                    cnt2[g][r] = (uint32_t)acc2;
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
        const uint64_t *off = c->offsets.data();
        parallel_for(G, NT, [&](size_t g) {
            uint32_t *k = cnt2[g].data(); // this group's next slot inside each read
            for_group(g, [&](const Rec &r) {
                uint64_t p = off[r.a] + k[r.a]++;
                iv[2 * p] = r.sa;
                iv[2 * p + 1] = r.ea;
                p = off[r.b] + k[r.b]++;
                iv[2 * p] = r.sb;
                iv[2 * p + 1] = r.eb;
            });
        });
    } else {
        std::vector<std::atomic<uint64_t>> cur(R + 1);
        parallel_for(NT, NT, [&](size_t w) {
            for (uint64_t r = R * w / NT; r < R * (w + 1) / NT; r++) cur[r].store(0, std::memory_order_relaxed);
        });
        parallel_for(G, NT, [&](size_t g) {
            for_group(g, [&](const Rec &r) {
                cur[r.a].fetch_add(1, std::memory_order_relaxed);
                cur[r.b].fetch_add(1, std::memory_order_relaxed);
            });
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
        parallel_for(G, NT, [&](size_t g) {
            for_group(g, [&](const Rec &r) {
                uint64_t p = cur[r.a].fetch_add(1, std::memory_order_relaxed);
                iv[2 * p] = r.sa;
                iv[2 * p + 1] = r.ea;
                p = cur[r.b].fetch_add(1, std::memory_order_relaxed);
                iv[2 * p] = r.sb;
                iv[2 * p + 1] = r.eb;
            });
        });
    }
}

size_t thread_count(int n_threads)
{
    if (n_threads <= 0) n_threads = (int)std::min(64u, usable_cpus()); // auto: every usable CPU up to 64
    return (size_t)std::max(1, n_threads);
}

// the whole ingest over one source; sink == nullptr keeps the records and builds the host CSR
int ingest(BlockSource &src, int format, int n_threads, const yacrd_rec_sink *sink, yacrd_csr **out)
{
    if (format != FMT_PAF && format != FMT_M4) return yh::fail("unknown overlap format");
    const size_t NT = thread_count(n_threads);
    Phase ph;
    IdTable ids;
    Parsed ps;
    if (parse_all(src, format, NT, sink, ids, ps)) return 1;
    ph.mark("parse");
    std::unique_ptr<yacrd_csr> c(new yacrd_csr());
    big_vector<uint32_t> dense;
    if (number_ids(ids, ps.text_bytes, NT, c.get(), dense)) return 1;
    ph.mark("number ids");
    c->n_records = ps.n_records;
    if (sink) {
        c->streamed = true;
        c->handle_map.assign(dense.begin(), dense.end());
    } else {
        fill_csr(ps, dense, NT, c.get());
        ph.mark("csr fill");
        // give the record segments back in parallel (hundreds of MB of mappings)
        parallel_for(ps.blocks.size(), NT, [&](size_t b) { ps.blocks[b].recs.reset(); });
        ph.mark("teardown");
    }
    *out = c.release();
    return 0;
}

// block size of the parse: big enough that a block's bookkeeping vanishes, small enough that the
// last blocks balance the threads and a streamed buffer goes out every few milliseconds
size_t block_bytes_for(size_t len, size_t NT)
{
    const size_t target = len / (NT * 8) + 1;
    return std::min<size_t>((size_t)8 << 20, std::max<size_t>((size_t)1 << 20, target));
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

int ingest_file(const char *path, int format, int n_threads, const yacrd_rec_sink *sink, yacrd_csr **out)
{
    if (!out || !path) return yh::fail("null argument");
    *out = nullptr;
    if (format == FMT_AUTO) format = sniff_format(path);
    if (format == FMT_AUTO)
        return yh::fail(std::string("Format detection of file ") + path + " failed");
    const size_t NT = thread_count(n_threads);
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
    const bool regular = S_ISREG(st.st_mode);
    if (!regular || yh::sniff_compression(magic, got > 0 ? (size_t)got : 0) != yh::COMP_NONE) {
        ::close(fd);
        yh::InStream in; // gzip / bzip2 / xz (or a pipe): decoded by a reader thread while the pool parses
        if (in.open(path)) return 1;
        DecodeSource src(in, (size_t)4 << 20, NT * 2 + 2);
        return ingest(src, format, (int)NT, sink, out);
    }
    if (st.st_size == 0) {
        ::close(fd);
        MemSource src("", 0, 1);
        return ingest(src, format, (int)NT, sink, out);
    }
    if (std::getenv("YACRD_INGEST_MMAP")) { // A/B: slices of a mapping instead of pread copies
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return yh::fail(std::string("mmap failed for ") + path);
        madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        int rc;
        {
            MemSource src((const char *)m, (size_t)st.st_size, block_bytes_for((size_t)st.st_size, NT));
            rc = ingest(src, format, (int)NT, sink, out);
        }
        munmap(m, (size_t)st.st_size);
        return rc;
    }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    int rc;
    {
        FileSource src(fd, (size_t)st.st_size, block_bytes_for((size_t)st.st_size, NT));
        rc = ingest(src, format, (int)NT, sink, out);
    }
    ::close(fd);
    return rc;
}

} // namespace

extern "C" {

const char *yacrd_host_last_error(void) { return yh::err_slot().c_str(); }

int yacrd_csr_from_memory(const char *text, size_t len, int format, int n_threads, yacrd_csr **out)
{
    if (!out || (!text && len)) return yh::fail("null argument");
    *out = nullptr;
    const size_t NT = thread_count(n_threads);
    MemSource src(text ? text : "", len, block_bytes_for(len, NT));
    return ingest(src, format, (int)NT, nullptr, out);
}

int yacrd_csr_from_file(const char *path, int format, int n_threads, yacrd_csr **out)
{
    return ingest_file(path, format, n_threads, nullptr, out);
}

int yacrd_ingest_stream(const char *path, int format, int n_threads, const yacrd_rec_sink *sink,
                        yacrd_csr **out)
{
    if (!sink || !sink->acquire || !sink->commit) return yh::fail("null sink");
    return ingest_file(path, format, n_threads, sink, out);
}

int yacrd_ingest_stream_memory(const char *text, size_t len, int format, int n_threads,
                               const yacrd_rec_sink *sink, yacrd_csr **out)
{
    if (!out || (!text && len)) return yh::fail("null argument");
    if (!sink || !sink->acquire || !sink->commit) return yh::fail("null sink");
    *out = nullptr;
    const size_t NT = thread_count(n_threads);
    MemSource src(text ? text : "", len, block_bytes_for(len, NT));
    return ingest(src, format, (int)NT, sink, out);
}

int yacrd_csr_handle_map(const yacrd_csr *c, const uint32_t **map, uint64_t *n_handles)
{
    if (!c || !map || !n_handles) return yh::fail("null argument");
    if (!c->streamed) return yh::fail("this CSR was not built by yacrd_ingest_stream");
    *map = c->handle_map.data();
    *n_handles = c->handle_map.size();
    return 0;
}

int yacrd_csr_get(const yacrd_csr *c, yacrd_csr_view *v)
{
    if (!c || !v) return yh::fail("null argument");
    v->n_reads = c->lengths.size();
    v->n_intervals = c->streamed ? 2 * c->n_records : (c->offsets.empty() ? 0 : c->offsets.back());
    v->n_records = c->n_records;
    v->offsets = c->streamed ? nullptr : c->offsets.data();
    v->intervals = c->streamed ? nullptr : c->intervals.get();
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
