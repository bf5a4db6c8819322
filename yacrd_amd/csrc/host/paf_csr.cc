// paf_csr.cc — overlap ingest: PAF / M4 text -> the CSR the engine consumes.
//
// Replaces reference Reads2Ovl::init_paf / init_m4 (src/reads2ovl/mod.rs:83-145; column contract
// src/io.rs:23-50) and FullMemory::add_overlap_and_length (src/reads2ovl/fullmemory.rs:82-90):
//   * both reads of every record get an interval (mod.rs:108-109 / :140-141)
//   * a read's length is the FIRST length seen for its id (fullmemory.rs:82-90)
//   * records may carry extra columns (csv `flexible(true)`), empty lines are skipped
//   * a short or non-numeric record is an error (the reference bails, mod.rs:93-97)
// Reads are numbered in first-appearance order.  Parsing is chunk-parallel: each thread interns
// ids locally, the local tables are merged in file order so numbering and the first-length rule
// do not depend on the thread count.  csv-crate quoting ("...") is not interpreted (unpinned by
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

struct Rec {
    uint32_t a, b, sa, ea, sb, eb;
};

// Interning table local to one chunk.  A slot carries the hash next to the id, so a probe
// touches one cache line of the table and (on a hash match) one of the arena.
struct Names {
    struct Slot {
        uint64_t hash;
        uint32_t id1; // id + 1, 0 = empty
        uint32_t nlen;
    };
    std::vector<char> arena;
    std::vector<uint64_t> off;  // start of each name in arena
    std::vector<uint32_t> nlen; // name length
    std::vector<uint64_t> hash;
    std::vector<uint64_t> rlen; // first length seen
    std::vector<Slot> table;
    uint64_t mask = 0;

    void init(size_t cap_pow2)
    {
        table.assign(cap_pow2, Slot{0, 0, 0});
        mask = cap_pow2 - 1;
    }
    void grow()
    {
        std::vector<Slot> nt(table.size() * 2, Slot{0, 0, 0});
        const uint64_t nm = nt.size() - 1;
        for (const Slot &sl : table) {
            if (!sl.id1) continue;
            uint64_t s = sl.hash & nm;
            while (nt[s].id1) s = (s + 1) & nm;
            nt[s] = sl;
        }
        table.swap(nt);
        mask = nm;
    }
    uint32_t intern(const char *p, size_t n, uint64_t h, uint64_t length)
    {
        uint64_t s = h & mask;
        for (;;) {
            const Slot &sl = table[s];
            if (!sl.id1) break;
            if (sl.hash == h && sl.nlen == n &&
                std::memcmp(arena.data() + off[sl.id1 - 1], p, n) == 0)
                return sl.id1 - 1;
            s = (s + 1) & mask;
        }
        const uint32_t id = (uint32_t)off.size();
        table[s] = Slot{h, id + 1, (uint32_t)n};
        off.push_back(arena.size());
        nlen.push_back((uint32_t)n);
        hash.push_back(h);
        rlen.push_back(length);
        arena.insert(arena.end(), p, p + n);
        if ((off.size() + 1) * 2 > table.size()) grow();
        return id;
    }
};

struct Chunk {
    const char *begin = nullptr, *end = nullptr;
    Names names;
    std::vector<Rec> recs;
    std::vector<uint32_t> l2g;
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
    c.names.init(1 << 12);
    c.recs.reserve((size_t)(c.end - c.begin) / 96 + 16);
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
        r.a = c.names.intern(ida, na, ha, la);
        r.b = c.names.intern(idb, nb, hb, lb);
        c.recs.push_back(r);
        p = eol + 1;
    }
}

void parse_chunk(Chunk &c, int format)
{
    const char delim = format == FMT_PAF ? '\t' : ' ';
    const int need = format == FMT_PAF ? 9 : 12;
    c.names.init(1 << 12);
    c.recs.reserve((size_t)(c.end - c.begin) / 96 + 16);
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
        r.a = c.names.intern(fb[ia], na, yh::hash_bytes(fb[ia], na), la);
        r.b = c.names.intern(fb[ib], nb, yh::hash_bytes(fb[ib], nb), lb);
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
    // auto: all cores up to 64 — beyond that the per-chunk id tables (each chunk re-interns most
    // ids) make the merge grow faster than the parse shrinks (profiles/r01_ingest*.json)
    if (n_threads <= 0) n_threads = (int)std::min(64u, std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    const size_t NT = (size_t)n_threads;
    // one chunk per thread, >= 2 MiB of text each
    const size_t T = std::max<size_t>(1, std::min<size_t>(NT, len / (2u << 20) + 1));
    Phase ph;

    std::vector<Chunk> chunks(T);
    {
        const char *p = text, *end = text + len;
        for (size_t t = 0; t < T; t++) {
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
    ph.mark("parse");

    // ---- merge the per-chunk id tables.  Global numbering = first appearance in the file, and a
    // read's length = the first length seen (fullmemory.rs:82-90): both are decided by the FIRST
    // chunk (file order) that holds the id.  The id space is sharded by hash so shards merge in
    // parallel; inside a shard, chunks are visited in file order.
    size_t S = 1;
    while (S < NT * 2 && S < 1024) S <<= 1;
    if (T == 1) S = 1;
    const int sshift = 64 - __builtin_ctzll((unsigned long long)S); // shard = hash >> sshift (S > 1)
    auto shard_of = [&](uint64_t h) { return S == 1 ? (size_t)0 : (size_t)(h >> sshift); };

    struct ChunkMerge {
        std::vector<uint32_t> by_shard;  // local ids grouped by shard, local order inside a shard
        std::vector<uint32_t> shard_off; // S + 1
        std::vector<uint64_t> owner;     // per local id: owner (chunk << 32 | local id); self if first
    };
    std::vector<ChunkMerge> cm(T);
    parallel_for(T, NT, [&](size_t t) {
        const Names &ln = chunks[t].names;
        ChunkMerge &m = cm[t];
        const size_t n = ln.off.size();
        m.shard_off.assign(S + 1, 0);
        for (size_t i = 0; i < n; i++) m.shard_off[shard_of(ln.hash[i]) + 1]++;
        for (size_t sidx = 0; sidx < S; sidx++) m.shard_off[sidx + 1] += m.shard_off[sidx];
        m.by_shard.resize(n);
        std::vector<uint32_t> cur(m.shard_off.begin(), m.shard_off.end() - 1);
        for (size_t i = 0; i < n; i++) m.by_shard[cur[shard_of(ln.hash[i])]++] = (uint32_t)i;
        m.owner.resize(n);
    });
    parallel_for(S, NT, [&](size_t sidx) {
        size_t total = 0;
        for (size_t t = 0; t < T; t++) total += cm[t].shard_off[sidx + 1] - cm[t].shard_off[sidx];
        size_t cap = 16;
        while (cap < total * 2) cap <<= 1;
        std::vector<uint64_t> tab(cap, ~0ull); // owner reference or empty
        const size_t mask = cap - 1;
        for (size_t t = 0; t < T; t++) {
            const Names &ln = chunks[t].names;
            ChunkMerge &m = cm[t];
            for (uint32_t k = m.shard_off[sidx]; k < m.shard_off[sidx + 1]; k++) {
                const uint32_t i = m.by_shard[k];
                const uint64_t h = ln.hash[i];
                size_t slot = (size_t)(h * 0x9E3779B97F4A7C15ull >> 20) & mask;
                for (;;) {
                    const uint64_t ref = tab[slot];
                    if (ref == ~0ull) {
                        tab[slot] = ((uint64_t)t << 32) | i;
                        m.owner[i] = ((uint64_t)t << 32) | i;
                        break;
                    }
                    const Names &on = chunks[ref >> 32].names;
                    const uint32_t oi = (uint32_t)ref;
                    if (on.hash[oi] == h && on.nlen[oi] == ln.nlen[i] &&
                        std::memcmp(on.arena.data() + on.off[oi], ln.arena.data() + ln.off[i],
                                    ln.nlen[i]) == 0) {
                        m.owner[i] = ref;
                        break;
                    }
                    slot = (slot + 1) & mask;
                }
            }
        }
    });
    // owners of chunk t get consecutive global ids in local (= first appearance) order
    std::vector<uint64_t> base(T + 1, 0);
    for (size_t t = 0; t < T; t++) {
        uint64_t own = 0;
        const ChunkMerge &m = cm[t];
        for (size_t i = 0; i < m.owner.size(); i++) own += m.owner[i] == (((uint64_t)t << 32) | i);
        base[t + 1] = base[t] + own;
    }
    const uint64_t R = base[T];
    if (R >= 0xFFFFFFFFull) return yh::fail("more than 2^32 - 2 reads");
    yacrd_csr *c = new yacrd_csr();
    c->lengths.resize(R);
    c->name_off.assign(R + 1, 0);
    parallel_for(T, NT, [&](size_t t) {
        Chunk &ch = chunks[t];
        const ChunkMerge &m = cm[t];
        ch.l2g.resize(m.owner.size());
        uint32_t g = (uint32_t)base[t];
        for (size_t i = 0; i < m.owner.size(); i++)
            if (m.owner[i] == (((uint64_t)t << 32) | i)) {
                ch.l2g[i] = g;
                c->lengths[g] = (uint32_t)ch.names.rlen[i]; // first length seen
                c->name_off[g + 1] = ch.names.nlen[i];
                g++;
            }
    });
    parallel_for(T, NT, [&](size_t t) {
        Chunk &ch = chunks[t];
        const ChunkMerge &m = cm[t];
        for (size_t i = 0; i < m.owner.size(); i++) {
            const uint64_t ref = m.owner[i];
            if (ref != (((uint64_t)t << 32) | i)) ch.l2g[i] = chunks[ref >> 32].l2g[(uint32_t)ref];
        }
    });
    for (uint64_t g = 0; g < R; g++) c->name_off[g + 1] += c->name_off[g];
    c->names.resize(c->name_off[R]);
    parallel_for(T, NT, [&](size_t t) {
        const Chunk &ch = chunks[t];
        const ChunkMerge &m = cm[t];
        for (size_t i = 0; i < m.owner.size(); i++)
            if (m.owner[i] == (((uint64_t)t << 32) | i))
                std::memcpy(c->names.data() + c->name_off[ch.l2g[i]],
                            ch.names.arena.data() + ch.names.off[i], ch.names.nlen[i]);
    });
    ph.mark("merge ids");

    // ---- counts -> offsets -> fill ------------------------------------------------------------
    std::vector<std::atomic<uint64_t>> cur(R + 1);
    parallel_for(NT, NT, [&](size_t w) {
        for (uint64_t r = R * w / NT; r < R * (w + 1) / NT; r++) cur[r].store(0, std::memory_order_relaxed);
    });
    parallel_for(T, NT, [&](size_t t) {
        Chunk &ch = chunks[t];
        for (Rec &r : ch.recs) {
            r.a = ch.l2g[r.a];
            r.b = ch.l2g[r.b];
            cur[r.a].fetch_add(1, std::memory_order_relaxed);
            cur[r.b].fetch_add(1, std::memory_order_relaxed);
        }
    });
    c->offsets.resize(R + 1);
    uint64_t acc = 0;
    for (uint64_t r = 0; r < R; r++) {
        c->offsets[r] = acc;
        const uint64_t n = cur[r].load(std::memory_order_relaxed);
        cur[r].store(acc, std::memory_order_relaxed);
        acc += n;
    }
    c->offsets[R] = acc;
    // new[] without () leaves the 8 B/interval buffer untouched: a vector would zero it on one
    // thread (tens of ms for 10^7 intervals) before the parallel fill overwrites every word
    c->intervals.reset(new uint32_t[2 * acc + 2]);
    uint32_t *iv = c->intervals.get();
    // With one chunk the fill is in line order; with several, the order inside a read depends on
    // thread timing (results do not: the sweep sorts).
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
    for (auto &ch : chunks) c->n_records += ch.recs.size();
    ph.mark("csr fill");
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
