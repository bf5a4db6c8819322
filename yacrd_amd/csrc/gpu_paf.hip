// gpu_paf.hip — PAF text -> read types with the PARSE on the GPU (include/yacrd_engine.h:
// yacrd_engine_ingest_paf).
//
// Reference: Reads2Ovl::init_paf (src/reads2ovl/mod.rs:83-113; one csv record = one PafRecord,
// src/io.rs:23-34: nine leading tab-separated columns, the rest ignored) feeding
// FullMemory::add_overlap_and_length (src/reads2ovl/fullmemory.rs:82-90: a read's length is the first one
// seen; reads are numbered by first appearance here, the reference's order is a hash map's).
//
// The host parser (host/paf_csr.cc) is bound by its id table: two hash lookups per line, cache misses the CPUs
// cannot hide (133 M overlaps/s on 16 CPUs).  Here the host only MOVES the text: threads pread fixed chunks of
// the file into pinned buffers, every chunk crosses PCIe at once (hipMemcpyAsync) into a mirror of the file in
// HBM; the calling thread hands every 128 MiB of landed text to the scan + parse kernels (the engine's stream waits
// for the chunks' copy events, the host for nothing).  On the device:
//   scan     newlines counted (the number of records to expect), '"' / lone CR looked for
//   parse    32 KiB of text staged in LDS per workgroup; a thread takes the lines that START in its 128 bytes: nine
//            fields checked as the host's fast path checks them, both ids hashed and looked up in an open-addressing table whose slots name the text
//            position of the id's FIRST CLAIMANT (an id is compared against the text itself: no copies), the
//            smallest position of every id kept with atomicMin; one 24-byte overlap record per line
//   number   occupied slots sorted by first position (radix_sort.h) = first-appearance numbering; the
//            length that follows the id at that position = the read's length; names gathered for the host
//   build    csr_build.h's count / scan / scatter on the records, then the engine's launch sequence
// Whatever the fast path does not take — a quote, a lone CR, a 0x integer, a malformed line, a length beyond
// u32 — makes the call return YACRD_EFALLBACK: the caller runs the host parser, which knows the csv crate's
// whole syntax and the error messages.  Plain files only (the codecs live in the host library).
#include "engine_internal.h"
#include "radix_sort.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>


using namespace yke;

namespace yk {

struct GpArgs {
    const unsigned char *text;
    u64 n;              // bytes
    u64 *claim;         // [cap]: 0 = empty, else text position of the id that claimed the slot + 1
    u64 *first_pos;     // [cap]: smallest (byte offset * 2 + side) at which the slot's id was seen
    u32 *slot_cnt;      // [cap]: intervals of the slot's id (one per record side): the CSR build's count pass, done here
    u32 mask;           // cap - 1
    OvlRec *recs;
    u64 rec_cap;
    unsigned long long *n_recs;
    u32 *status;        // bit 0: the text needs the host parser; bit 1: the id table is full
    unsigned long long *n_lines; // newline-terminated lines + an unterminated last one
    // the part of the text one launch works on (the parse runs segment by segment while later chunks still cross PCIe)
    u64 begin, end;     // scan: bytes [begin, end); parse: the tiles from begin / kGpTile on (its grid = their number)
    u64 avail;          // bytes [0, avail) of the mirror have landed (== n for the last segment)
    u32 delim;          // the field delimiter: '\t' (PAF) or ' ' (M4: src/reads2ovl/mod.rs:116-117)
    u32 partial;        // the mirror ends in front of the file's end (a range + its overhang): a record that reaches the mirror's end is not whole
    u32 skip_head;      // the mirror is a byte RANGE of the file and the byte in front of it is no newline: position 0 lies
                        // inside a line that belongs to the range before (yacrd_engines_ingest_overlaps)
};

constexpr int kGpT = 256; // threads per workgroup

// pinned host memory -> the mirror, by a kernel (see the host side: used when the mirror is a fresh allocation)
__global__ __launch_bounds__(256) void gp_blit_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n16; i += (u64)gridDim.x * 256u) dst[i] = src[i];
}
constexpr u32 kNeedHost = 1u, kTableFull = 2u;

// ---- pass 1: how many lines, and is there anything the fast path must not see ------------------------------
__global__ __launch_bounds__(kGpT) void gp_scan_kernel(GpArgs a)
{
    const u64 stride = (u64)gridDim.x * kGpT * 16u;
    u32 lines = 0, special = 0;
    for (u64 base = a.begin + ((u64)blockIdx.x * kGpT + threadIdx.x) * 16u; base < a.end; base += stride) {
        // 16 bytes per thread and step (segments begin on chunk boundaries; the mirror is padded: reads beyond n see
        // zeros, reads beyond an inner segment's end see the next chunk, which has landed)
        const uint4 v = *reinterpret_cast<const uint4 *>(a.text + base);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64 i = base + (u64)k;
            const u32 c = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            if (i < a.end) {
                lines += c == '\n' ? 1u : 0u;
                special |= c == '"' ? 1u : 0u;
                if (c == '\r') { // fine only as the first half of CRLF
                    const u32 nx = i + 1 < a.n ? a.text[i + 1] : 0u;
                    special |= nx != '\n' ? 1u : 0u;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lines += (u32)__shfl_xor((int)lines, d, 64);
    special = wave_or(special);
    if (lane_id() == 0) {
        if (lines) atomicAdd(a.n_lines, (unsigned long long)lines);
        if (special) atomicOr(a.status, kNeedHost);
    }
    if (a.end == a.n && blockIdx.x == 0 && threadIdx.x == 0 && a.n && a.text[a.n - 1] != '\n') atomicAdd(a.n_lines, 1ull);
}

// ---- pass 2: parse ---------------------------------------------------------------------------------------------
// A workgroup stages 32 KiB of text (+ 1 KiB beyond it) in LDS with coalesced 16-byte loads; a thread then takes
// the lines that START in its 128 bytes and reads their first nine fields byte by byte from LDS (byte loads from
// global memory, one cache line per lane and instruction, ran this kernel at 24 GB/s).  A record needs nothing
// behind its ninth field, so a line of any length costs its start only; fields that reach beyond the staged
// window (ids of hundreds of bytes) are read from global memory.
constexpr int kGpTile = kGpT * 128, kGpOver = 1024;

struct GpText {
    const unsigned char *lds, *glob;
    u64 t0, t1; // the staged window [t0, t1)
    __device__ __forceinline__ u32 operator[](u64 i) const { return i - t0 < t1 - t0 ? lds[i - t0] : glob[i]; }
};

__device__ __forceinline__ u64 gp_hash(const GpText &t, u64 p, u32 n)
{
    u64 h = 0xcbf29ce484222325ull ^ ((u64)n * 0x9E3779B97F4A7C15ull);
    for (u32 i = 0; i < n; i++) h = (h ^ t[p + i]) * 0x100000001b3ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    return h ^ (h >> 32);
}
// decimal u64 with an optional '+', at least one digit, then the delimiter (or, for the last field, the line's end);
// clears `ok` when the field is anything else.  `le` = the text's end: a line ends at '\n' (or "\r\n").
__device__ __forceinline__ u64 gp_uint(const GpText &t, u64 &q, u64 n, u64 limit, bool last, bool &ok, u32 dl)
{
    u64 i = q;
    u32 c = i < n ? t[i] : '\n';
    if (c == '+') {
        i++;
        c = i < n ? t[i] : '\n';
    }
    const u64 b = i;
    u64 v = 0;
    bool good = true;
    while (c - '0' <= 9u) {
        const u32 d = c - '0';
        good = good && v <= (0xFFFFFFFFFFFFFFFFull - d) / 10u;
        v = v * 10u + d;
        i++;
        c = i < n ? t[i] : '\n';
    }
    good = good && i != b && v <= limit;
    if (c == dl) {
        i++;
    } else { // the line's end: '\n', or '\r' in front of one (a lone CR never gets here: the scan refused it)
        good = good && last && (c == '\n' || c == '\r');
    }
    q = i;
    ok = ok && good;
    return v;
}
// the slot of the id at text position p (n bytes, hash h): claimed for it if it is new.  ~0u = table full.
__device__ __forceinline__ u32 gp_intern(const GpArgs &a, const GpText &t, u64 p, u32 n, u64 h)
{
    const unsigned char *g = a.text;
    u32 s = (u32)h & a.mask;
    // (a bounded walk: a table that is nearly full — more distinct ids than it was sized for, e.g. a read-to-reference
    // PAF with a new query id on every line — would otherwise cost every later lookup a walk over all of it before
    // the file is handed to the host parser anyway)
    const u32 max_probes = min(a.mask, 1024u);
    for (u32 probes = 0; probes <= max_probes; probes++, s = (s + 1u) & a.mask) {
        u64 w = __hip_atomic_load(&a.claim[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            const u64 old = atomicCAS((unsigned long long *)&a.claim[s], 0ull, (unsigned long long)(p + 1));
            if (old == 0) return s;
            w = old;
        }
        const u64 c = w - 1; // the claimant's id starts there and ends at a delimiter (or it would not have been parsed)
        if (c == p) return s;
        bool same = c + n < a.avail && g[c + n] == a.delim; // (a shorter id near the end of what has landed: not this one)
        for (u32 i = 0; same && i < n; i++) same = g[c + i] == t[p + i];
        if (same) return s;
    }
    return ~0u;
}

// M4 = false: PafRecord (src/io.rs:23-34: a la sa ea strand b lb sb eb ...); true: M4Record (src/io.rs:36-50:
// a b error shared strand_a sa ea la strand_b sb eb lb ...), space-separated.
template <bool M4>
__global__ __launch_bounds__(kGpT) void gp_parse_kernel(GpArgs a)
{
    const u32 dl = a.delim;
    __shared__ __attribute__((aligned(16))) unsigned char win[kGpTile + kGpOver];
    const u64 tile0 = a.begin + (u64)blockIdx.x * (u64)kGpTile; // (a.begin is a multiple of the tile size)
    if (tile0 >= a.n) return;
    // Bytes at and beyond `n` are not looked at: the text's end for the last segment, what has landed so far for the
    // others (their lines end long before it; one that does not is the host parser's: flagged below).
    const u64 n = a.avail;
    {
        // the window, 16 bytes per thread and step (the mirror is padded by 64 zero bytes and the last step is clipped
        // to whole 16-byte pieces inside it; an inner segment's `avail` is a chunk boundary)
        const u64 lim = a.avail == a.n ? ((a.n + 63) & ~(u64)15) : a.avail;
        const u64 want = min((u64)(kGpTile + kGpOver), lim - tile0);
        for (u64 i = (u64)threadIdx.x * 16u; i < want; i += (u64)kGpT * 16u)
            *reinterpret_cast<uint4 *>(win + i) = *reinterpret_cast<const uint4 *>(a.text + tile0 + i);
    }
    __syncthreads();
    GpText t;
    t.lds = win;
    t.glob = a.text;
    t.t0 = tile0;
    t.t1 = min(n, tile0 + (u64)(kGpTile + kGpOver));
    const u64 lo = tile0 + (u64)threadIdx.x * 128u;
    if (lo >= n) return;
    const u64 hi = min(n, lo + 128u);
    u32 status = 0;
    // lines that start in [lo, hi): position 0, or the byte after a newline
    u64 p = lo;
    if (lo != 0) {
        u64 q = lo - 1;
        while (q < hi && (q < tile0 ? (u32)a.text[q] : t[q]) != '\n') q++;
        p = q + 1; // (>= hi when no line starts here)
    } else if (a.skip_head) { // (a range that begins inside a line: that line is the previous range's)
        u64 q = 0;
        while (q < hi && t[q] != '\n') q++;
        p = q + 1;
    }
    while (p < hi) {
        // the line's first byte decides: empty lines are skipped (csv), "\r\n" alone is one as well
        const u32 c0 = t[p];
        u64 next = p; // where to look for the next line start: filled in below
        bool ok = false;
        u64 la = 0, lb = 0, sa = 0, ea = 0, sb = 0, eb = 0, ia = p, ib = p;
        u32 na = 0, nb = 0;
        if (c0 == '\n') { // an empty line: the next one starts right behind it
            p = p + 1;
            continue;
        } else if (c0 == '\r' && p + 1 < n && t[p + 1] == '\n') {
            p = p + 2;
            continue;
        } else {
            ok = true;
            u64 q = p;
            // an id: up to the delimiter (an id that runs into the line's end makes the record too short)
            auto take_id = [&](u64 &start, u32 &len) {
                start = q;
                u32 c = q < n ? t[q] : '\n';
                while (c != dl && c != '\n') {
                    q++;
                    c = q < n ? t[q] : '\n';
                }
                ok = ok && c == dl;
                len = (u32)(q - start);
                q++;
            };
            // a strand: exactly one UTF-8 scalar (serde char), judged by its lead byte like the host
            auto take_char = [&]() {
                u64 tb = q;
                u32 c = tb < n ? t[tb] : '\n';
                const u32 lead = c;
                while (c != dl && c != '\n') {
                    tb++;
                    c = tb < n ? t[tb] : '\n';
                }
                const u32 want = lead < 0x80u ? 1u : (lead >> 5) == 6u ? 2u : (lead >> 4) == 14u ? 3u : (lead >> 3) == 30u ? 4u : 0u;
                ok = ok && c == dl && want != 0u && tb - q == (u64)want;
                q = tb + 1;
            };
            // M4's error rate: the plain decimal forms only — [+-] digits [. digits] [e [+-] digits] with a digit on
            // both sides of the point — whatever else Rust's f64::from_str takes (inf, nan, ".5", "1.") is the host's
            auto take_f64 = [&]() {
                u32 c = q < n ? t[q] : '\n';
                auto step = [&]() {
                    q++;
                    c = q < n ? t[q] : '\n';
                };
                auto digits = [&]() {
                    u32 k = 0;
                    while (c - '0' <= 9u) {
                        k++;
                        step();
                    }
                    return k;
                };
                if (c == '+' || c == '-') step();
                bool good = digits() != 0u;
                if (c == '.') {
                    step();
                    good = good && digits() != 0u;
                }
                if (c == 'e' || c == 'E') {
                    step();
                    if (c == '+' || c == '-') step();
                    good = good && digits() != 0u;
                }
                ok = ok && good && c == dl;
                q++;
            };
            if constexpr (!M4) {
                take_id(ia, na);
                la = gp_uint(t, q, n, ~0ull, false, ok, dl);
                sa = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                ea = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                if (ok) take_char();
                if (ok) take_id(ib, nb);
                lb = gp_uint(t, q, n, ~0ull, false, ok, dl);
                sb = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                eb = gp_uint(t, q, n, 0xFFFFFFFFull, true, ok, dl);
            } else {
                take_id(ia, na);
                if (ok) take_id(ib, nb);
                if (ok) take_f64();
                (void)gp_uint(t, q, n, ~0ull, false, ok, dl); // _shared_min: u64
                if (ok) take_char();
                sa = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                ea = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                la = gp_uint(t, q, n, ~0ull, false, ok, dl);
                if (ok) take_char();
                sb = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                eb = gp_uint(t, q, n, 0xFFFFFFFFull, false, ok, dl);
                lb = gp_uint(t, q, n, ~0ull, true, ok, dl);
            }
            ok = ok && la <= 0xFFFFFFFFull && lb <= 0xFFFFFFFFull; // (the engine's limit; the host parser says so)
            if (!ok) status |= kNeedHost;
            if (q >= n && (n < a.n || a.partial)) status |= kNeedHost; // (a record that reaches into text still on its way, or beyond a range's overhang: megabytes long)
            next = q; // the rest of the line holds nothing for the record: its newline is looked for below
        }
        u32 s1 = 0, s2 = 0;
        if (ok) {
            s1 = gp_intern(a, t, ia, na, gp_hash(t, ia, na));
            s2 = gp_intern(a, t, ib, nb, gp_hash(t, ib, nb));
            if (s1 == ~0u || s2 == ~0u) {
                status |= kTableFull;
                ok = false;
            }
        }
        // One slot per record, handed out per wavefront: the lanes that have a record now ask together.
        const u64 m = __builtin_amdgcn_ballot_w64(ok);
        if (ok) {
            const u32 leader = (u32)__builtin_ctzll(m);
            u64 base = 0;
            if (lane_id() == leader) base = atomicAdd(a.n_recs, (unsigned long long)__builtin_popcountll(m));
            base = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(base >> 32), (int)leader) << 32) |
                   (u32)__builtin_amdgcn_readlane((int)(u32)base, (int)leader);
            const u64 at = base + (u64)__builtin_popcountll(m & ((1ull << lane_id()) - 1ull));
            const u64 pa = p * 2u, pb = p * 2u + 1u;
            if (pa < __hip_atomic_load(&a.first_pos[s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMin((unsigned long long *)&a.first_pos[s1], (unsigned long long)pa);
            if (pb < __hip_atomic_load(&a.first_pos[s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMin((unsigned long long *)&a.first_pos[s2], (unsigned long long)pb);
            if (s1 == s2) {
                atomicAdd(&a.slot_cnt[s1], 2u);
            } else {
                atomicAdd(&a.slot_cnt[s1], 1u);
                atomicAdd(&a.slot_cnt[s2], 1u);
            }
            if (at < a.rec_cap) {
                OvlRec r;
                r.a = s1, r.b = s2, r.sa = (u32)sa, r.ea = (u32)ea, r.sb = (u32)sb, r.eb = (u32)eb;
                a.recs[at] = r;
            }
        }
        // the next line start inside this thread's bytes, if any
        while (next < hi && t[next] != '\n') next++;
        p = next + 1;
        if (next >= hi) break;
    }
    if (status) atomicOr(a.status, status);
}

// ---- pass 3: the reads -------------------------------------------------------------------------------------------
// A workgroup walks its share of the table, gathers the occupied slots in LDS and asks for room in the output once
// per kCollectBuf of them: the table is 0.5-5 % full, and one returning atomic per wavefront that meets an occupied
// slot (600 k of them for 600 k reads, performed one after the other at the memory side: ~8 ns each) made this
// kernel 5.9 ms of a 110 ms call.  (The order of the output does not matter: it is sorted next.)
constexpr u32 kCollectBuf = 1024;
__global__ __launch_bounds__(256) void gp_collect_kernel(const u64 *claim, const u64 *first_pos, u32 cap, u64 *keys,
                                                         u32 *slots, u32 *n_out)
{
    __shared__ u64 s_key[kCollectBuf + 256];
    __shared__ u32 s_slot[kCollectBuf + 256];
    __shared__ u32 s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const u64 per = ((u64)cap + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per, hi = min((u64)cap, lo + per);
    auto flush = [&]() { // (every thread of the workgroup)
        __syncthreads();
        const u32 n = s_n;
        if (threadIdx.x == 0 && n) s_base = atomicAdd(n_out, n);
        __syncthreads();
        for (u32 i = threadIdx.x; i < n; i += 256u) {
            keys[s_base + i] = s_key[i];
            slots[s_base + i] = s_slot[i];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    for (u64 s0 = lo; s0 < hi; s0 += 256u) { // (uniform trip count)
        const u64 s = s0 + threadIdx.x;
        const bool used = s < hi && claim[s] != 0;
        const u64 m = __builtin_amdgcn_ballot_w64(used);
        if (m) {
            u32 base = 0;
            if (lane_id() == (u32)__builtin_ctzll(m)) base = atomicAdd(&s_n, (u32)__builtin_popcountll(m));
            base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
            if (used) {
                const u32 at = base + (u32)__builtin_popcountll(m & ((1ull << lane_id()) - 1ull));
                s_key[at] = first_pos[s];
                s_slot[at] = (u32)s;
            }
        }
        __syncthreads();
        if (s_n >= kCollectBuf) flush(); // (uniform: s_n is read behind a barrier)
    }
    flush();
}
// read g (first-appearance order) = the id first seen at keys[g]: its slot -> g, its length (PAF: the field after
// the id; M4: the 8th / 12th field of the line), the extent of its name, its number of intervals
__global__ __launch_bounds__(256) void gp_number_kernel(const unsigned char *t, const u64 *keys, const u32 *slots,
                                                        u32 n_reads, u32 *handle_map, u32 *lengths, u32 *name_len,
                                                        u64 *name_at, const u32 *slot_cnt, u32 *cnt, u32 delim)
{
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_reads) return;
    handle_map[slots[g]] = g;
    cnt[g] = slot_cnt[slots[g]];
    const u64 pos = keys[g];
    const bool m4 = delim == ' ', second = (pos & 1u) != 0;
    const int id_field = m4 ? (second ? 1 : 0) : (second ? 5 : 0), len_field = m4 ? (second ? 11 : 7) : (second ? 6 : 1);
    u64 q = pos >> 1; // the line's start
    for (int f = 0; f < id_field; f++) {
        while (t[q] != delim) q++;
        q++;
    }
    const u64 id0 = q;
    while (t[q] != delim) q++;
    name_at[g] = id0;
    name_len[g] = (u32)(q - id0);
    q++;
    for (int f = id_field + 1; f < len_field; f++) {
        while (t[q] != delim) q++;
        q++;
    }
    if (t[q] == '+') q++;
    u64 v = 0;
    while ((u32)t[q] - '0' <= 9u) v = v * 10u + ((u32)t[q++] - '0');
    lengths[g] = (u32)v; // (checked <= u32 by the parse)
}
__global__ __launch_bounds__(256) void gp_names_kernel(const unsigned char *t, const u64 *name_at, const u64 *name_off,
                                                       u32 n_reads, unsigned char *names)
{
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_reads) return;
    const u64 from = name_at[g], to = name_off[g], n = name_off[g + 1] - to;
    for (u64 i = 0; i < n; i++) names[to + i] = t[from + i];
}


// ==== several engines, one file: the reads of the engines' byte ranges merged (yacrd_engines_ingest_overlaps) ============
// Every engine has parsed a byte range of the text (parse_range) and numbered the reads IT saw.  The same read shows up in
// many ranges; the merge engine gets every range's read list — name, first position in the file, first length, number of
// intervals: a few dozen bytes per read and range, against the gigabytes of text that never leave their device — and
// interns the names once more, in a table whose slots name list ENTRIES.  Then, as for one engine: occupied slots
// sorted by first position = first-appearance numbering over the whole file; a read's length is the one seen at that
// position (src/reads2ovl/fullmemory.rs:82-90).  Every entry learns its read's number, every engine rewrites its records
// from table slots to those numbers, and the reads are dealt out to the engines as contiguous ranges of numbers.
struct GmArgs {
    const unsigned char *names; // every range's names, end to end
    const u64 *noff;            // [M + 1] entry -> its name's extent in `names`
    const u64 *fp;              // [M] first position in the FILE * 2 + side
    const u32 *len, *cnt;       // [M] the length seen there; intervals of the read inside the range
    u64 M;
    u64 *claim, *first_pos;     // [cap]: entry + 1 of the slot's first claimant; smallest first position
    u32 *slot_cnt;              // [cap]
    u32 mask;
    u32 *entry_slot;            // [M]
    u32 *status;
};
__global__ __launch_bounds__(256) void gm_shift_kernel(u64 *fp, u64 *noff, u64 n, u64 fp_add, u64 noff_add)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) fp[i] += fp_add, noff[i] += noff_add;
}
__global__ __launch_bounds__(256) void gm_intern_kernel(GmArgs a)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= a.M) return;
    const u64 p = a.noff[i];
    const u32 n = (u32)(a.noff[i + 1] - p);
    u64 h = 0xcbf29ce484222325ull ^ ((u64)n * 0x9E3779B97F4A7C15ull);
    for (u32 k = 0; k < n; k++) h = (h ^ a.names[p + k]) * 0x100000001b3ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    u32 s = (u32)h & a.mask, found = ~0u;
    for (u32 probes = 0; probes <= a.mask; probes++, s = (s + 1u) & a.mask) {
        u64 w = __hip_atomic_load(&a.claim[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            const u64 old = atomicCAS((unsigned long long *)&a.claim[s], 0ull, (unsigned long long)(i + 1));
            if (old == 0) {
                found = s;
                break;
            }
            w = old;
        }
        const u64 c = w - 1, q = a.noff[c];
        bool same = (u32)(a.noff[c + 1] - q) == n;
        for (u32 k = 0; same && k < n; k++) same = a.names[q + k] == a.names[p + k];
        if (same) {
            found = s;
            break;
        }
    }
    if (found == ~0u) { // (cannot happen: the table has twice as many slots as there are entries)
        atomicOr(a.status, kTableFull);
        return;
    }
    a.entry_slot[i] = found;
    atomicMin((unsigned long long *)&a.first_pos[found], (unsigned long long)a.fp[i]);
    atomicAdd(&a.slot_cnt[found], a.cnt[i]);
}
// read g of the whole file = the name first seen at keys[g]: its slot -> g, its intervals, where its name lies
__global__ __launch_bounds__(256) void gm_number_kernel(GmArgs a, const u32 *slots, u32 n_reads, u32 *slot_read, u32 *cnt,
                                                        u32 *name_len, u64 *name_at)
{
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_reads) return;
    const u32 s = slots[g];
    slot_read[s] = g;
    cnt[g] = a.slot_cnt[s];
    const u64 c = a.claim[s] - 1;
    name_at[g] = a.noff[c];
    name_len[g] = (u32)(a.noff[c + 1] - a.noff[c]);
}
// every entry: its read's number; the entry that saw the read first gives it its length
__global__ __launch_bounds__(256) void gm_entry_kernel(GmArgs a, const u32 *slot_read, u32 *entry_read, u32 *lengths)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= a.M) return;
    const u32 s = a.entry_slot[i], g = slot_read[s];
    entry_read[i] = g;
    if (a.fp[i] == a.first_pos[s]) lengths[g] = a.len[i];
}
// on every engine: table slot -> the range's read -> the file's read; then the records' handles
__global__ __launch_bounds__(256) void gm_slot_read_kernel(u32 *slot_map, u64 cap, const u32 *range_read_to_file)
{
    const u64 s = (u64)blockIdx.x * 256u + threadIdx.x;
    if (s >= cap) return;
    const u32 l = slot_map[s];
    if (l != 0xFFFFFFFFu) slot_map[s] = range_read_to_file[l];
}
__global__ __launch_bounds__(256) void gm_rewrite_kernel(OvlRec *recs, u64 n, const u32 *slot_map)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) {
        uint2 *ab = reinterpret_cast<uint2 *>(recs + i);
        const uint2 v = *ab;
        *ab = make_uint2(slot_map[v.x], slot_map[v.y]);
    }
}
// the reads [lo, hi) are this engine's (numbered from 0 here), every other one lives elsewhere
__global__ __launch_bounds__(256) void gm_own_kernel(u32 *own, u32 n_reads, u32 lo, u32 hi)
{
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g < n_reads) own[g] = (g >= lo && g < hi) ? g - lo : (u32)YACRD_HANDLE_ELSEWHERE;
}

} // namespace yk

namespace {

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (first position, slot) pairs by first position, on the engine's stream: radix_sort.h, as many eight-bit passes as keys
// below `key_bound` need.  The passes go back and forth between the two pairs of arrays (the inputs are scratch: clobbered);
// an odd number of them, so that the last one lands in the outputs.
int sort_by_first_position(yacrd_engine *e, u64 *keys, u64 *keys_out, u32 *slots, u32 *slots_out, u32 n, u64 key_bound, DevBuf &tmp,
                           DevBuf &part)
{
    int bits = 1;
    while (bits < 64 && (key_bound >> bits) != 0) bits++;
    int passes = (bits + 7) / 8;
    passes |= 1;
    const u32 tiles = (u32)(((u64)n + yk::kRsTile - 1) / yk::kRsTile);
    const size_t cells = (size_t)256 * tiles;
    const size_t first_at = (cells * sizeof(u32) + 255) & ~(size_t)255;
    HIP_TRY(tmp.reserve(first_at + (cells + 1) * sizeof(u64) + 64));
    u32 *hist = tmp.as<u32>();
    u64 *first = reinterpret_cast<u64 *>(tmp.as<char>() + first_at);
    u64 *ka = keys, *kb = keys_out;
    u32 *va = slots, *vb = slots_out;
    for (int p = 0; p < passes; p++) {
        const u32 shift = (u32)std::min(8 * p, 56); // (a pass beyond the key's bits sees digit 0 everywhere: a stable copy)
        const bool beyond = 8 * p >= 64;
        hipLaunchKernelGGL(yk::rs_hist_kernel, dim3(tiles), dim3(yk::kRsThreads), 0, e->stream, ka, (u64)n, beyond ? 63u : shift, tiles, hist);
        if (const int rcs = scan_u32_to_u64(e, hist, (u64)cells, first, part)) return rcs;
        hipLaunchKernelGGL(yk::rs_scatter_kernel, dim3(tiles), dim3(yk::kRsThreads), 0, e->stream, ka, va, (u64)n, beyond ? 63u : shift, tiles,
                           first, kb, vb);
        std::swap(ka, kb);
        std::swap(va, vb);
    }
    return YACRD_OK;
}
// gzip / bzip2 / xz (the magic bytes niffler looks at, src/util.rs:57-70)
bool is_compressed_magic(int fd)
{
    unsigned char mg[6] = {0};
    const ssize_t k = ::pread(fd, mg, sizeof mg, 0);
    return k >= 2 && ((mg[0] == 0x1f && mg[1] == 0x8b) || (k >= 3 && mg[0] == 'B' && mg[1] == 'Z' && mg[2] == 'h') ||
                      (k >= 6 && mg[0] == 0xFD && std::memcmp(mg + 1, "7zXZ", 4) == 0 && mg[5] == 0));
}

struct Scratch { // the call's device buffers; they stay with the engine (grow-only) and go when it is destroyed
    DevBuf text, claim, first_pos, slot_cnt, recs, ctl, keys, keys2, slots, slots2, tmp, map, name_len, name_at, name_off,
        names, cnt, part, err, gather, gmap;
    // the merge engine's (yacrd_engines_ingest_overlaps)
    DevBuf m_fp, m_len, m_cnt, m_noff, m_names, m_slot, m_read, g_claim, g_fp, g_cnt, g_keys, g_slots, g_keys2, g_slots2, g_slot_read,
        g_rcnt, g_nlen, g_nat, g_noff, g_names, g_len;
    size_t bytes()
    {
        size_t n = 0;
        for (DevBuf *b : all()) n += b->cap;
        return n;
    }
    void release()
    {
        for (DevBuf *b : all()) b->release();
    }
    ~Scratch() { release(); }

private:
    std::vector<DevBuf *> all()
    {
        return {&text, &claim, &first_pos, &slot_cnt, &recs, &ctl, &keys, &keys2, &slots, &slots2, &tmp, &map, &name_len,
                &name_at, &name_off, &names, &cnt, &part, &err, &gather, &gmap, &m_fp, &m_len, &m_cnt, &m_noff, &m_names, &m_slot,
                &m_read, &g_claim, &g_fp, &g_cnt, &g_keys, &g_slots, &g_keys2, &g_slots2, &g_slot_read, &g_rcnt, &g_nlen, &g_nat, &g_noff,
                &g_names, &g_len};
    }
};

} // namespace

extern "C" {

int yacrd_engine_trim(yacrd_engine *e)
{
    if (!e) return fail(YACRD_EINVAL, "engine is null");
    if (e->pending.active || e->host_pending) return fail(YACRD_EINVAL, "the engine has a submitted batch pending");
    DeviceGuard guard(e->device);
    if (e->paf_scratch) static_cast<Scratch *>(e->paf_scratch)->release();
    return YACRD_OK;
}

void yacrd_reads_free(yacrd_reads *r)
{
    if (!r) return;
    std::free(r->lengths);
    std::free(r->name_off);
    std::free(r->names);
    std::memset(r, 0, sizeof(*r));
}

int yacrd_engine_ingest_paf(yacrd_engine *e, const char *path, int n_threads, uint32_t coverage, double not_coverage,
                            yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    return yacrd_engine_ingest_overlaps(e, path, 1, n_threads, coverage, not_coverage, out, reads, stats);
}

} // extern "C"

namespace {
// where the text comes from: a file (pread) or memory (a compressed file the host has inflated)
struct TextSource {
    int fd = -1;
    const char *mem = nullptr;
    // `len` bytes at `off` into dst; false = read error
    bool fetch(char *dst, size_t len, u64 off) const
    {
        if (mem) {
            std::memcpy(dst, mem + off, len);
            return true;
        }
        size_t got = 0;
        while (got < len) {
            const ssize_t k = ::pread(fd, dst + got, len - got, (off_t)(off + got));
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return false;
            got += (size_t)k;
        }
        return true;
    }
};
int ingest_text(yacrd_engine *e, const TextSource &src, u64 n, bool m4, int n_threads, uint32_t coverage, double not_coverage,
                yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats);
} // namespace

extern "C" {

static int ingest_format(const char *path, int format, bool &m4)
{
    if (format == 0) { // by file name, like util::get_file_type (src/util.rs:39-55)
        if (!path) return fail(YACRD_EINVAL, "format 0 (by name) needs a file name");
        const std::string name(path);
        auto has = [&](const char *x) { return name.find(x) != std::string::npos; };
        format = (has(".m4") || has(".mhap")) ? 2 : has(".paf") ? 1 : 0;
        if (format == 0) return fail(YACRD_EINVAL, std::string("cannot tell the overlap format of ") + path);
    }
    if (format != 1 && format != 2) return fail(YACRD_EINVAL, "format: 0 = by name, 1 = PAF, 2 = M4");
    m4 = format == 2;
    return YACRD_OK;
}

int yacrd_engine_ingest_overlaps(yacrd_engine *e, const char *path, int format, int n_threads, uint32_t coverage,
                                 double not_coverage, yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    if (!e || !path || !out || !reads) return fail(YACRD_EINVAL, "null argument");
    bool m4 = false;
    if (const int rcf = ingest_format(path, format, m4)) return rcf;
    std::memset(out, 0, sizeof(*out));
    std::memset(reads, 0, sizeof(*reads));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (e->pending.active || e->host_pending) return fail(YACRD_EINVAL, "the engine has a submitted batch pending");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(YACRD_EINVAL, std::string("cannot open ") + path);
    struct FdGuard {
        int fd;
        ~FdGuard() { ::close(fd); }
    } fdg{fd};
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return fail(YACRD_EFALLBACK, "not a regular file: the host parser reads it");
    // gzip / bzip2 / xz: not text — the caller inflates the file (yacrd_text_from_file, libyacrd_host) and hands the text to
    // yacrd_engine_ingest_overlaps_mem, or takes the host parser
    if (is_compressed_magic(fd))
        return fail(YACRD_EFALLBACK, "a compressed file: inflate it (yacrd_text_from_file + yacrd_engine_ingest_overlaps_mem) or take the host parser");
    TextSource src;
    src.fd = fd;
    return ingest_text(e, src, (u64)st.st_size, m4, n_threads, coverage, not_coverage, out, reads, stats);
}

int yacrd_engine_ingest_overlaps_mem(yacrd_engine *e, const char *text, uint64_t n, int format, int n_threads, uint32_t coverage,
                                     double not_coverage, yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    if (!e || (!text && n) || !out || !reads) return fail(YACRD_EINVAL, "null argument");
    bool m4 = false;
    if (const int rcf = ingest_format(nullptr, format, m4)) return rcf;
    std::memset(out, 0, sizeof(*out));
    std::memset(reads, 0, sizeof(*reads));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (e->pending.active || e->host_pending) return fail(YACRD_EINVAL, "the engine has a submitted batch pending");
    TextSource src;
    src.mem = text ? text : "";
    return ingest_text(e, src, n, m4, n_threads, coverage, not_coverage, out, reads, stats);
}

} // extern "C"

namespace {

// (allocating and freeing ~0.8 GB of HBM per call cost 1.5 ms of a 15 ms run: the buffers stay with the engine)
Scratch *scratch_of(yacrd_engine *e)
{
    if (!e->paf_scratch) {
        e->paf_scratch = new (std::nothrow) Scratch();
        e->paf_scratch_free = [](void *p) { delete static_cast<Scratch *>(p); };
    }
    return static_cast<Scratch *>(e->paf_scratch);
}

// Bytes [begin, begin + len) of the text -> on the engine: the overlap records of the lines that START there (handles = id-table
// slots), the table, and the range's reads numbered by first appearance INSIDE the range: S.map (slot -> read), S.keys2 (first
// positions * 2 + side, relative to `begin`), e->in_len (lengths), S.cnt (intervals), S.name_off / S.names.  The whole file for
// one engine; a range per engine for several (yacrd_engines_ingest_overlaps), cut on chunk boundaries.
struct RangeOut {
    u64 n_recs = 0, cap = 0, name_bytes = 0;
    u32 R = 0;
    double t_start = 0, t_text = 0, t_parse = 0;
};
int parse_range(yacrd_engine *e, const TextSource &src, u64 file_n, u64 begin, u64 len, bool skip_head, bool m4, int n_threads, RangeOut &ro)
{
    DeviceGuard guard(e->device);
    Scratch *Sp = scratch_of(e);
    if (!Sp) return fail(YACRD_ENOMEM, "host allocation failed");
    Scratch &S = *Sp;
    // the mirror holds the range and, behind it, up to one chunk more of the file: a line that starts in the range ends there
    // (or the parse says so: kNeedHost)
    constexpr u64 kOverhang = (u64)4 << 20;
    const u64 n = std::min<u64>(file_n, begin + len + (begin + len < file_n ? kOverhang : 0)) - begin; // bytes in the mirror
    const u64 parse_end = len; // lines that start at or behind it are the next range's
    const double t_start = now_ms();
    {
        // The parse on the device wants the text, a 24-byte record per line, the id table, the CSR and the region
        // slots in HBM at once: ~2.6 x the file (measured: 37.2 GB of text -> 96 GB).  An input beyond that goes to
        // the host parser, whose streamed records need ~0.5 x: answered here, before anything is allocated.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double have = (double)free_b + (double)S.text.cap + (double)S.recs.cap + (double)e->stage.cap +
                                (double)e->in_iv.cap + (double)S.gather.cap;
            if (2.6 * (double)n + (double)((size_t)256 << 20) > have)
                return fail(YACRD_EFALLBACK, "the file is too large to be parsed in this device's free memory: the host parser streams it");
        }
    }

    // ---- what does not depend on the text's content: the mirror, the id table, the control words
    const void *mirror_before = S.text.p;
    HIP_TRY(S.text.reserve((size_t)n + 64));
    // Fresh HBM is slow to fill, whoever fills it: the text of a 37 GB file takes 0.70 s to get into a mirror that
    // is being reused and 2.2-2.4 s (hipMemcpyAsync; 5.1 s in a process's first call) or 0.85-2.4 s (copy kernels;
    // 2.1 s in a first call) into one that was allocated for this call — same box, same run, profiles/r03_e2e_large.log.
    // So the buffers stay with the engine (yacrd_engine_trim gives them back), chunks go by copy kernel into a fresh
    // mirror and by hipMemcpyAsync, 15 % faster once the memory is warm, into a reused one.
    const bool blit = S.text.p != mirror_before;
    HIP_TRY(hipMemsetAsync(S.text.as<char>() + n, 0, 64, e->stream)); // (the scan reads 16 bytes at a time)
    u64 cap = (u64)1 << 20;
    while (cap < n / 64 && cap < ((u64)1 << 31)) cap <<= 1; // ids are a small fraction of the lines; a full table = fallback
    HIP_TRY(S.claim.reserve((size_t)cap * sizeof(u64)));
    HIP_TRY(S.first_pos.reserve((size_t)cap * sizeof(u64)));
    HIP_TRY(S.slot_cnt.reserve((size_t)cap * sizeof(u32)));
    HIP_TRY(S.ctl.reserve(64));
    HIP_TRY(hipMemsetAsync(S.ctl.p, 0, 64, e->stream));
    HIP_TRY(hipMemsetAsync(S.claim.p, 0, (size_t)cap * sizeof(u64), e->stream));
    HIP_TRY(hipMemsetAsync(S.first_pos.p, 0xFF, (size_t)cap * sizeof(u64), e->stream));
    HIP_TRY(hipMemsetAsync(S.slot_cnt.p, 0, (size_t)cap * sizeof(u32), e->stream));
    // Room for the records: the number of lines is only known when the last byte has been scanned, and the parse
    // does not wait for that — so from the line density of the file's first MiB plus a quarter (never more than
    // one record per 17 bytes); a parse that outgrows it is repeated with the exact number below.
    u64 rec_cap = n / 17 + 2;
    {
        const size_t sample = (size_t)std::min<u64>(n, (u64)1 << 20);
        std::vector<char> head(sample + 1);
        if (!src.fetch(head.data(), sample, begin)) return fail(YACRD_EINVAL, "read error in the overlap file");
        u64 nl = 0;
        for (const char *q = head.data(), *end = q + sample; (q = (const char *)std::memchr(q, '\n', (size_t)(end - q))) != nullptr; q++) nl++;
        if (nl) rec_cap = std::min<u64>(rec_cap, (u64)((double)n / (double)sample * (double)nl * 1.25) + 4096);
    }
    HIP_TRY(S.recs.reserve((size_t)rec_cap * sizeof(yk::OvlRec)));
    yk::GpArgs ga{};
    ga.text = S.text.as<unsigned char>();
    ga.n = n;
    ga.n_lines = S.ctl.as<unsigned long long>();
    ga.n_recs = S.ctl.as<unsigned long long>() + 1;
    ga.status = reinterpret_cast<u32 *>(S.ctl.as<unsigned long long>() + 2);
    u32 *d_nreads = reinterpret_cast<u32 *>(S.ctl.as<unsigned long long>() + 3);
    ga.claim = S.claim.as<u64>();
    ga.first_pos = S.first_pos.as<u64>();
    ga.slot_cnt = S.slot_cnt.as<u32>();
    ga.mask = (u32)(cap - 1);
    ga.recs = S.recs.as<yk::OvlRec>();
    ga.rec_cap = rec_cap;
    ga.delim = m4 ? (u32)' ' : (u32)'\t';
    ga.skip_head = skip_head ? 1u : 0u;
    ga.partial = begin + n < file_n ? 1u : 0u;
    if ((n + yk::kGpTile - 1) / yk::kGpTile >= 0x7FFFFFFFull) return fail(YACRD_EFALLBACK, "file too large for the device parser");
    // scan + parse of the bytes [begin, end) on the engine's stream (begin on a tile boundary)
    auto launch_segment = [&](u64 begin, u64 end, u64 avail) {
        ga.begin = begin, ga.end = end, ga.avail = avail;
        const u32 scan_grid = (u32)std::min<u64>(((end - begin) / (yk::kGpT * 16u)) + 1, (u64)e->num_cu * 16);
        hipLaunchKernelGGL(yk::gp_scan_kernel, dim3(scan_grid), dim3(yk::kGpT), 0, e->stream, ga);
        const u64 tiles = (end - begin + yk::kGpTile - 1) / yk::kGpTile;
        if (tiles && m4) hipLaunchKernelGGL(yk::gp_parse_kernel<true>, dim3((u32)tiles), dim3(yk::kGpT), 0, e->stream, ga);
        else if (tiles) hipLaunchKernelGGL(yk::gp_parse_kernel<false>, dim3((u32)tiles), dim3(yk::kGpT), 0, e->stream, ga);
    };

    // ---- the text: pread chunks -> pinned buffers -> HBM, all chunks in flight at once; THIS thread hands every
    // segment of kSeg chunks to the scan + parse kernels as soon as it (and the chunk behind it: a tile's overhang,
    // the byte after a CR) has landed — the engine's stream waits for the chunks' copy events, the host for nothing
    {
        // (the pinned arena stays with the engine: pinning 100 MB costs more than moving 367 MB through it)
        // (128 MiB per scan + parse launch.  The two kernels take 0.9 ms for 367 MB, so what the overlap buys is small;
        // segments of 16 / 32 MiB got in the way of the copy threads: tools/paf_matrix.py, profiles/r03/paf_matrix.log)
        constexpr size_t kChunk = (size_t)4 << 20, kSeg = 32;
        static_assert(kChunk % yk::kGpTile == 0, "segments begin on tile boundaries");
        const size_t n_chunks = (size_t)((n + kChunk - 1) / kChunk);
        unsigned T = n_threads > 0 ? (unsigned)n_threads : 6u; // (more threads only get in each other's way: 367 MB in 10 ms with 4-8)
        T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(T, 32u), std::max<size_t>(n_chunks, 1)));
        const size_t n_buf = (size_t)2 * T;
        if (e->paf_arena_cap < n_buf * kChunk) {
            if (e->paf_arena) (void)hipHostFree(e->paf_arena);
            e->paf_arena = nullptr;
            e->paf_arena_cap = 0;
            HIP_TRY(hipHostMalloc(&e->paf_arena, n_buf * kChunk));
            e->paf_arena_cap = n_buf * kChunk;
        }
        char *arena = (char *)e->paf_arena;
        std::vector<hipStream_t> copy(T, nullptr);
        std::vector<hipEvent_t> ev(n_chunks, nullptr); // one per chunk: recorded behind its copy
        std::unique_ptr<std::atomic<int>[]> landed(new std::atomic<int>[n_chunks + 1]);
        for (size_t c = 0; c <= n_chunks; c++) landed[c].store(0);
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        for (unsigned t = 0; t < T; t++)
            if (hipStreamCreateWithFlags(&copy[t], hipStreamNonBlocking) != hipSuccess) bad = 1;
        for (size_t c = 0; c < n_chunks; c++)
            if (hipEventCreateWithFlags(&ev[c], hipEventDisableTiming) != hipSuccess) bad = 1;
        auto work = [&](unsigned t) { // thread t owns buffers 2t and 2t + 1: one fills while the other flies
            if (hipSetDevice(e->device) != hipSuccess) bad = 1;
            long prev[2] = {-1, -1}; // the chunk that last flew from each buffer
            for (int turn = 0; !bad.load(); turn ^= 1) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const size_t b = (size_t)2 * t + (size_t)turn;
                if (prev[turn] >= 0 && hipEventSynchronize(ev[(size_t)prev[turn]]) != hipSuccess) bad = 1;
                char *dst = arena + b * kChunk;
                const size_t off = c * kChunk, clen = (size_t)std::min<u64>(kChunk, n - off);
                if (!src.fetch(dst, clen, begin + (u64)off)) bad = 2;
                if (bad.load()) break;
                if (blit && (clen & 15)) std::memset(dst + clen, 0, 16 - (clen & 15)); // (the file's last piece: zeros, not leftovers, behind it)
                if (blit) { // (the arena's buffers are 4 MiB: whole 16-byte pieces; the mirror is padded by 64 bytes)
                    hipLaunchKernelGGL(yk::gp_blit_kernel, dim3(256), dim3(256), 0, copy[t], reinterpret_cast<uint4 *>(S.text.as<char>() + off),
                                       reinterpret_cast<const uint4 *>(dst), (u64)((clen + 15) / 16));
                    if (hipEventRecord(ev[c], copy[t]) != hipSuccess) bad = 1;
                } else if (hipMemcpyAsync(S.text.as<char>() + off, dst, clen, hipMemcpyHostToDevice, copy[t]) != hipSuccess ||
                           hipEventRecord(ev[c], copy[t]) != hipSuccess)
                    bad = 1;
                prev[turn] = (long)c;
                landed[c].store(1, std::memory_order_release); // (its event is recorded: the dispatcher may wait on it)
            }
            if (copy[t]) (void)hipStreamSynchronize(copy[t]);
        };
        std::vector<std::thread> th;
        if (!bad.load())
            for (unsigned t = 0; t < T; t++) th.emplace_back(work, t);
        // the dispatcher
        size_t waited = 0;
        for (size_t c0 = 0; c0 < n_chunks && !bad.load(); c0 += kSeg) {
            const size_t c1 = std::min(c0 + kSeg, n_chunks), need = std::min(c1 + 1, n_chunks);
            while (waited < need && !bad.load()) {
                if (!landed[waited].load(std::memory_order_acquire)) {
                    struct timespec ts = {0, 20000};
                    nanosleep(&ts, nullptr);
                    continue;
                }
                if (hipStreamWaitEvent(e->stream, ev[waited], 0) != hipSuccess) bad = 1;
                waited++;
            }
            if (bad.load()) break;
            if ((u64)c0 * kChunk < parse_end)
                launch_segment((u64)c0 * kChunk, std::min<u64>(parse_end, (u64)c1 * kChunk), std::min<u64>(n, (u64)need * kChunk));
        }
        for (auto &x : th) x.join();
        for (hipStream_t s2 : copy)
            if (s2) (void)hipStreamDestroy(s2);
        if (bad.load()) (void)hipStreamSynchronize(e->stream); // (kernels may still wait on events about to go)
        for (hipEvent_t x : ev)
            if (x) (void)hipEventDestroy(x);
        if (bad.load() == 2) return fail(YACRD_EINVAL, "read error in the overlap file");
        if (bad.load()) return fail(YACRD_ENODEV, "PAF text to HBM: a HIP call failed");
        (void)hipGetLastError();
    }
    const double t_text = now_ms();

    // ---- what the scan and the parse found
    unsigned long long h_ctl[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h_ctl, S.ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    if ((u32)h_ctl[2] & yk::kNeedHost)
        return fail(YACRD_EFALLBACK, "the text holds a '\"', a lone CR or a line that is not a plain PAF / M4 record (too few "
                                     "columns, a 0x integer, a length beyond u32, an error rate written as inf or .5 ...): the "
                                     "host parser decides");
    if ((u32)h_ctl[2] & yk::kTableFull) return fail(YACRD_EFALLBACK, "more read ids than the device table holds");
    const u64 n_lines = h_ctl[0];
    if (n_lines >= 0x7FFFFFFFull * 2) return fail(YACRD_EFALLBACK, "too many lines for the device parser");
    u64 n_recs = h_ctl[1];
    if (n_recs > n_lines + 2) return fail(YACRD_EINTERNAL, "device parser: more records than lines");
    if (n_recs > rec_cap) { // the estimate fell short (line lengths far from uniform): once more, with room for every record
        rec_cap = n_recs + 1;
        HIP_TRY(S.recs.reserve((size_t)rec_cap * sizeof(yk::OvlRec)));
        ga.recs = S.recs.as<yk::OvlRec>();
        ga.rec_cap = rec_cap;
        HIP_TRY(hipMemsetAsync(S.ctl.p, 0, 64, e->stream));
        HIP_TRY(hipMemsetAsync(S.claim.p, 0, (size_t)cap * sizeof(u64), e->stream));
        HIP_TRY(hipMemsetAsync(S.first_pos.p, 0xFF, (size_t)cap * sizeof(u64), e->stream));
        HIP_TRY(hipMemsetAsync(S.slot_cnt.p, 0, (size_t)cap * sizeof(u32), e->stream));
        launch_segment(0, parse_end, n);
        HIP_TRY(hipMemcpyAsync(h_ctl, S.ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        if (((u32)h_ctl[2] & (yk::kNeedHost | yk::kTableFull)) || h_ctl[1] != n_recs)
            return fail(YACRD_EINTERNAL, "device parser: the second parse disagrees with the first");
    }
    // ---- the reads: occupied slots by first position
    HIP_TRY(S.keys.reserve((size_t)cap * sizeof(u64) + 64));
    HIP_TRY(S.slots.reserve((size_t)cap * sizeof(u32) + 64));
    hipLaunchKernelGGL(yk::gp_collect_kernel, dim3((u32)std::min<u64>((cap + 255) / 256, (u64)e->num_cu * 8)), dim3(256), 0, e->stream,
                       ga.claim, ga.first_pos, (u32)cap, S.keys.as<u64>(), S.slots.as<u32>(), d_nreads);
    HIP_TRY(hipMemcpyAsync(h_ctl, S.ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const u32 R = (u32)h_ctl[3];
    const double t_parse = now_ms();
    if ((u64)R * 2 > cap && R > 1024) return fail(YACRD_EFALLBACK, "more read ids than the device table holds comfortably");

    // sort (first position, slot)
    HIP_TRY(S.keys2.reserve((size_t)R * sizeof(u64) + 64));
    HIP_TRY(S.slots2.reserve((size_t)R * sizeof(u32) + 64));
    if (R) {
        const int rcs = sort_by_first_position(e, S.keys.as<u64>(), S.keys2.as<u64>(), S.slots.as<u32>(), S.slots2.as<u32>(), R, 2 * n + 2, S.tmp, S.part);
        if (rcs) return rcs;
    }
    HIP_TRY(S.map.reserve((size_t)cap * sizeof(u32)));
    HIP_TRY(S.name_len.reserve((size_t)(R + 4) * sizeof(u32)));
    HIP_TRY(S.name_at.reserve((size_t)(R + 1) * sizeof(u64)));
    HIP_TRY(S.name_off.reserve((size_t)(R + 2) * sizeof(u64)));
    HIP_TRY(e->in_len.reserve((size_t)(R + 1) * sizeof(u32)));
    HIP_TRY(S.cnt.reserve((size_t)(R + 4) * sizeof(u32)));
    HIP_TRY(hipMemsetAsync(S.cnt.p, 0, (size_t)(R + 4) * sizeof(u32), e->stream));
    HIP_TRY(hipMemsetAsync(S.map.p, 0xFF, (size_t)cap * sizeof(u32), e->stream));
    const u32 rg = (R + 255) / 256;
    if (R)
        hipLaunchKernelGGL(yk::gp_number_kernel, dim3(rg), dim3(256), 0, e->stream, ga.text, S.keys2.as<u64>(), S.slots2.as<u32>(),
                           R, S.map.as<u32>(), e->in_len.as<u32>(), S.name_len.as<u32>(), S.name_at.as<u64>(), S.slot_cnt.as<u32>(), S.cnt.as<u32>(), ga.delim);
    {
        const int rcs = scan_u32_to_u64(e, S.name_len.as<u32>(), (u64)R, S.name_off.as<u64>(), S.part);
        if (rcs) return rcs;
    }
    u64 name_bytes = 0;
    HIP_TRY(hipMemcpyAsync(&name_bytes, S.name_off.as<u64>() + R, sizeof(u64), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(S.names.reserve((size_t)name_bytes + 64));
    if (R)
        hipLaunchKernelGGL(yk::gp_names_kernel, dim3(rg), dim3(256), 0, e->stream, ga.text, S.name_at.as<u64>(), S.name_off.as<u64>(), R,
                           S.names.as<unsigned char>());
    ro.n_recs = n_recs, ro.R = R, ro.cap = cap, ro.name_bytes = name_bytes;
    ro.t_start = t_start, ro.t_text = t_text, ro.t_parse = t_parse;
    return YACRD_OK;
}

int ingest_text(yacrd_engine *e, const TextSource &src, u64 n, bool m4, int n_threads, uint32_t coverage, double not_coverage,
                yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    DeviceGuard guard(e->device);
    RangeOut ro;
    if (const int rcp = parse_range(e, src, n, 0, n, false, m4, n_threads, ro)) return rcp;
    Scratch &S = *static_cast<Scratch *>(e->paf_scratch);
    const u32 R = ro.R;
    const u64 n_recs = ro.n_recs, cap = ro.cap, name_bytes = ro.name_bytes;
    const double t_start = ro.t_start, t_text = ro.t_text, t_parse = ro.t_parse;
    // the reads, to the host (while the CSR is built)
    reads->n_reads = R;
    reads->n_records = n_recs;
    reads->lengths = (uint32_t *)std::malloc(((size_t)R + 1) * sizeof(uint32_t));
    reads->name_off = (uint64_t *)std::malloc(((size_t)R + 1) * sizeof(uint64_t));
    reads->names = (char *)std::malloc((size_t)name_bytes + 1);
    if (!reads->lengths || !reads->name_off || !reads->names) {
        yacrd_reads_free(reads);
        return fail(YACRD_ENOMEM, "host allocation failed");
    }
    int rc = YACRD_OK;
    auto body = [&]() -> int {
        if (R) HIP_TRY(hipMemcpyAsync(reads->lengths, e->in_len.p, (size_t)R * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(reads->name_off, S.name_off.p, ((size_t)R + 1) * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
        if (name_bytes) HIP_TRY(hipMemcpyAsync(reads->names, S.names.p, (size_t)name_bytes, hipMemcpyDeviceToHost, e->stream));
        // ---- CSR on the device (csr_build.h through stream.hip's helper), then the engine
        const u64 n_iv = 2 * n_recs;
        const RecSlab slab{S.recs.as<yk::OvlRec>(), n_recs};
        const int rcb = csr_from_records(e, &slab, 1, S.map.as<u32>(), cap, R, S.cnt, S.part, S.err, nullptr, nullptr, true);
        if (rcb) return rcb;
        const double t_build = now_ms();
        int rc2 = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), R, n_iv, coverage, not_coverage);
        if (rc2) return rc2;
        const double t_run = now_ms();
        rc2 = fetch_result(e, out);
        if (stats) {
            stats->text_bytes = n;
            stats->n_records = n_recs;
            stats->n_reads = R;
            stats->text_ms = (float)(t_text - t_start);
            stats->parse_ms = (float)(t_parse - t_text);
            stats->build_ms = (float)(t_build - t_parse);
            stats->run_ms = (float)(t_run - t_build);
            stats->d2h_ms = (float)(now_ms() - t_run);
        }
        return rc2;
    };
    rc = body();
    if (rc) yacrd_reads_free(reads);
    return rc;
}

} // namespace

// ---- several engines, one file --------------------------------------------------------------------------------------
namespace {
// ---- device memory of engine `from` -> device memory of engine `to`, on `to`'s stream ---------------------------------------
// (the source is complete: its stream was waited for; the caller's current device is `to`'s).  Three routes:
//   same device                   hipMemcpyAsync, device to device
//   two devices, peer access      hipMemcpyPeerAsync: xGMI carries it
//   two devices, no peer access   staged through two pinned host buffers: device -> host on `from`'s device, host -> device
//                                 on `to`'s stream (the runtime would do the same, one pageable piece at a time)
// Which of the last two a PAIR of devices takes is found once, when a group first uses the pair (hipDeviceCanAccessPeer +
// hipDeviceEnablePeerAccess in both directions; anything but success / "already enabled" = staged).  This builder has never
// seen two GPUs in one box (missing #2), so the routes can be FORCED on one device, where the tests run:
// YACRD_TEST_FORCE_PEER_COPY=peer sends every cross-engine copy through hipMemcpyPeerAsync (source device = destination
// device is legal), =staged through the host bounce; tests/test_gpu_ingest_group.py runs the group under both.
enum PeerRoute { kRouteSame = 0, kRoutePeer = 1, kRouteStaged = 2 };
int forced_peer_route()
{
    static const int v = [] {
        const char *ev = std::getenv("YACRD_TEST_FORCE_PEER_COPY");
        if (!ev || !*ev) return -1;
        if (!std::strcmp(ev, "peer") || !std::strcmp(ev, "1")) return (int)kRoutePeer;
        if (!std::strcmp(ev, "staged") || !std::strcmp(ev, "host")) return (int)kRouteStaged;
        return -1;
    }();
    return v;
}
std::mutex g_peer_mu;
signed char g_peer_route[64][64]; // 0: not looked at yet; else PeerRoute + 1
std::atomic<unsigned long long> g_peer_copies[3]; // copies per route (yacrd_debug_peer_copy_counts: the tests look)
PeerRoute route_between(int to_dev, int from_dev)
{
    const int forced = forced_peer_route();
    if (to_dev == from_dev) return forced >= 0 ? (PeerRoute)forced : kRouteSame;
    if (forced == (int)kRouteStaged) return kRouteStaged;
    if (to_dev < 0 || from_dev < 0 || to_dev >= 64 || from_dev >= 64) return kRouteStaged;
    std::lock_guard<std::mutex> lock(g_peer_mu);
    signed char &slot = g_peer_route[to_dev][from_dev];
    if (slot) return (PeerRoute)(slot - 1);
    PeerRoute r = kRouteStaged;
    int can_a = 0, can_b = 0;
    if (hipDeviceCanAccessPeer(&can_a, to_dev, from_dev) == hipSuccess && hipDeviceCanAccessPeer(&can_b, from_dev, to_dev) == hipSuccess &&
        can_a && can_b) {
        bool ok = true;
        for (int k = 0; k < 2 && ok; k++) { // both directions: the merge pulls towards engine 0, the deal pushes away from it
            const int self = k == 0 ? to_dev : from_dev, peer = k == 0 ? from_dev : to_dev;
            DeviceGuard guard(self);
            const hipError_t er = hipDeviceEnablePeerAccess(peer, 0);
            if (er != hipSuccess && er != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError(); // ("already enabled" is sticky otherwise)
        }
        if (ok) r = kRoutePeer;
    } else {
        (void)hipGetLastError();
    }
    slot = (signed char)(r + 1);
    g_peer_route[from_dev][to_dev] = slot;
    return r;
}
// the staged route: two pinned buffers per destination engine, a piece flies host -> device while the next one is fetched
struct StageBuf {
    void *p[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    hipStream_t from_stream = nullptr;
    int from_dev = -1;
    ~StageBuf()
    {
        for (int i = 0; i < 2; i++) {
            if (p[i]) (void)hipHostFree(p[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (from_stream) {
            DeviceGuard guard(from_dev);
            (void)hipStreamDestroy(from_stream);
        }
    }
};
constexpr size_t kStagePiece = (size_t)16 << 20;
hipError_t copy_staged(yacrd_engine *to, void *dst, yacrd_engine *from, const void *src, size_t bytes)
{
    StageBuf sb; // (per call: the route is a fallback, its setup cost is not what matters on it)
    for (int i = 0; i < 2; i++) {
        hipError_t er = hipHostMalloc(&sb.p[i], std::min(bytes, kStagePiece));
        if (er == hipSuccess) er = hipEventCreateWithFlags(&sb.ev[i], hipEventDisableTiming);
        if (er != hipSuccess) return er;
    }
    sb.from_dev = from->device;
    {
        DeviceGuard guard(from->device);
        const hipError_t er = hipStreamCreateWithFlags(&sb.from_stream, hipStreamNonBlocking);
        if (er != hipSuccess) return er;
    }
    int turn = 0;
    for (size_t at = 0; at < bytes; at += kStagePiece, turn ^= 1) {
        const size_t m = std::min(kStagePiece, bytes - at);
        if (sb.busy[turn]) {
            const hipError_t er = hipEventSynchronize(sb.ev[turn]); // the buffer's last host -> device copy has left it
            if (er != hipSuccess) return er;
        }
        {
            DeviceGuard guard(from->device); // device -> host runs on the SOURCE device's stream
            hipError_t er = hipMemcpyAsync(sb.p[turn], (const char *)src + at, m, hipMemcpyDeviceToHost, sb.from_stream);
            if (er == hipSuccess) er = hipStreamSynchronize(sb.from_stream);
            if (er != hipSuccess) return er;
        }
        hipError_t er = hipMemcpyAsync((char *)dst + at, sb.p[turn], m, hipMemcpyHostToDevice, to->stream);
        if (er == hipSuccess) er = hipEventRecord(sb.ev[turn], to->stream);
        if (er != hipSuccess) return er;
        sb.busy[turn] = true;
    }
    return hipStreamSynchronize(to->stream); // (the pinned buffers go away with `sb`)
}
hipError_t copy_between(yacrd_engine *to, void *dst, yacrd_engine *from, const void *src, size_t bytes)
{
    if (!bytes) return hipSuccess;
    const PeerRoute r = route_between(to->device, from->device);
    g_peer_copies[(int)r]++;
    if (r == kRouteSame) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, to->stream);
    if (r == kRoutePeer) return hipMemcpyPeerAsync(dst, to->device, src, from->device, bytes, to->stream);
    return copy_staged(to, dst, from, src, bytes);
}

int ingest_text_group(yacrd_engine *const *E, uint32_t N, const TextSource &src, u64 n, bool m4, int n_threads, uint32_t coverage,
                      double not_coverage, yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    // ---- byte ranges: whole 4 MiB chunks (the parse's segments begin on tile boundaries), the same number for every engine
    // (YACRD_TEST_RANGE_BYTES, tests: a smaller grain — a multiple of the parse's 32 KiB tiles — so that small texts are cut too)
    static const u64 kChunk = [] {
        const char *ev = std::getenv("YACRD_TEST_RANGE_BYTES");
        const u64 v = ev ? std::strtoull(ev, nullptr, 10) : 0;
        return (v >= (u64)yk::kGpTile && v % (u64)yk::kGpTile == 0) ? v : ((u64)4 << 20);
    }();
    const u64 chunks = (n + kChunk - 1) / kChunk, per = (chunks + N - 1) / N;
    std::vector<u64> B(N + 1);
    for (uint32_t d = 0; d <= N; d++) B[d] = std::min<u64>(n, (u64)d * per * kChunk);
    for (uint32_t d = 0; d < N; d++)
        if (!scratch_of(E[d])) return fail(YACRD_ENOMEM, "host allocation failed"); // (every later scratch_of(E[d]) finds it)
    // ---- does it fit?  Answered here, before anything is allocated or moved (parse_range's own check knows only
    // its range).  Per DEVICE: what its engines' ranges need for the parse (text, records, id table, CSR and region slots of
    // as many reads: ~2.6 x the range, parse_range) plus the records every one of its engines gathers from ranges parsed on
    // OTHER devices (24 bytes per line, ~0.32 x their text at 75 bytes a line; engines of one device read each other's
    // records in place).  An input beyond that is the host parser's (its stream group needs ~0.5 x the file over all GPUs).
    {
        std::vector<int> seen;
        for (uint32_t d = 0; d < N; d++) {
            const int dev = E[d]->device;
            if (std::find(seen.begin(), seen.end(), dev) != seen.end()) continue;
            seen.push_back(dev);
            double here = 0, held = 0;
            uint32_t engines_here = 0;
            for (uint32_t k = 0; k < N; k++)
                if (E[k]->device == dev) {
                    Scratch &Sk = *scratch_of(E[k]);
                    here += (double)(B[k + 1] - B[k]);
                    held += (double)Sk.text.cap + (double)Sk.recs.cap + (double)E[k]->stage.cap + (double)E[k]->in_iv.cap + (double)Sk.gather.cap;
                    engines_here++;
                }
            const double need = 2.6 * here + 0.32 * ((double)n - here) * engines_here + (double)((size_t)256 << 20) * engines_here;
            DeviceGuard guard(dev);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b + held)
                return fail(YACRD_EFALLBACK, "the file is too large to be parsed in the GPUs' free memory (ranges + the records gathered "
                                             "from the other GPUs): the host parser streams it");
        }
    }
    uint32_t active = 0;
    while (active < N && B[active] < B[active + 1]) active++;
    if (active < 2) return ingest_text(E[0], src, n, m4, n_threads, coverage, not_coverage, out, reads, stats);
    const uint32_t A = active; // engines that parse (every engine sweeps, below)
    std::vector<char> head_skip(A, 0);
    for (uint32_t d = 1; d < A; d++) {
        char c = 0;
        if (!src.fetch(&c, 1, B[d] - 1)) return fail(YACRD_EINVAL, "read error in the overlap file");
        head_skip[d] = c != '\n';
    }
    const double t0 = now_ms();
    std::vector<RangeOut> ro(A);
    std::vector<int> codes(N, YACRD_OK);
    std::vector<std::string> errs(N);
    auto run_all = [&](uint32_t count, auto &&fn) { // fn(d) on `count` host threads, one per engine; the first failure wins
        std::vector<std::thread> th;
        for (uint32_t d = 1; d < count; d++)
            th.emplace_back([&, d] {
                codes[d] = fn(d);
                if (codes[d]) errs[d] = err_slot();
            });
        codes[0] = fn(0);
        if (codes[0]) errs[0] = err_slot();
        for (auto &t : th) t.join();
        // (a real error before "not for the device parser")
        for (uint32_t d = 0; d < count; d++)
            if (codes[d] && codes[d] != YACRD_EFALLBACK) return fail(codes[d], "engine " + std::to_string(d) + ": " + errs[d]);
        for (uint32_t d = 0; d < count; d++)
            if (codes[d]) return fail(codes[d], errs[d]);
        return (int)YACRD_OK;
    };
    // the threads that move text are shared out (engines on one device share its link as well)
    const int copy_threads = std::max(1, (n_threads > 0 ? n_threads : 8) / (int)A);
    int rc = run_all(A, [&](uint32_t d) -> int {
        const int r = parse_range(E[d], src, n, B[d], B[d + 1] - B[d], head_skip[d] != 0, m4, copy_threads, ro[d]);
        if (r) return r;
        DeviceGuard guard(E[d]->device);
        HIP_TRY(hipStreamSynchronize(E[d]->stream)); // (the names kernel: other engines read what this one made)
        return YACRD_OK;
    });
    if (rc) return rc;
    const double t_parsed = now_ms();

    // ---- the merge, on engine 0
    yacrd_engine *e0 = E[0];
    Scratch &S0 = *scratch_of(e0);
    std::vector<u64> eb(A + 1, 0), nb(A + 1, 0);
    for (uint32_t d = 0; d < A; d++) eb[d + 1] = eb[d] + ro[d].R, nb[d + 1] = nb[d] + ro[d].name_bytes;
    const u64 M = eb[A], NB = nb[A];
    if (M >= 0x7FFFFFFFull) return fail(YACRD_EFALLBACK, "too many read ids for the device parser");
    u64 capG = 1024;
    while (capG < 2 * M) capG <<= 1;
    u32 Rg = 0;
    u64 g_name_bytes = 0;
    std::vector<u32> h_rcnt;
    {
        DeviceGuard guard(e0->device);
        HIP_TRY(S0.m_fp.reserve((size_t)(M + 1) * sizeof(u64)));
        HIP_TRY(S0.m_len.reserve((size_t)(M + 1) * sizeof(u32)));
        HIP_TRY(S0.m_cnt.reserve((size_t)(M + 1) * sizeof(u32)));
        HIP_TRY(S0.m_noff.reserve((size_t)(M + 2) * sizeof(u64)));
        HIP_TRY(S0.m_names.reserve((size_t)NB + 64));
        HIP_TRY(S0.m_slot.reserve((size_t)(M + 1) * sizeof(u32)));
        HIP_TRY(S0.m_read.reserve((size_t)(M + 1) * sizeof(u32)));
        HIP_TRY(S0.g_claim.reserve((size_t)capG * sizeof(u64)));
        HIP_TRY(S0.g_fp.reserve((size_t)capG * sizeof(u64)));
        HIP_TRY(S0.g_cnt.reserve((size_t)capG * sizeof(u32)));
        HIP_TRY(S0.g_slot_read.reserve((size_t)capG * sizeof(u32)));
        HIP_TRY(S0.ctl.reserve(64));
        for (uint32_t d = 0; d < A; d++) {
            Scratch &Sd = *scratch_of(E[d]);
            const size_t R = ro[d].R;
            HIP_TRY(copy_between(e0, S0.m_fp.as<u64>() + eb[d], E[d], Sd.keys2.p, R * sizeof(u64)));
            HIP_TRY(copy_between(e0, S0.m_len.as<u32>() + eb[d], E[d], E[d]->in_len.p, R * sizeof(u32)));
            HIP_TRY(copy_between(e0, S0.m_cnt.as<u32>() + eb[d], E[d], Sd.cnt.p, R * sizeof(u32)));
            HIP_TRY(copy_between(e0, S0.m_noff.as<u64>() + eb[d], E[d], Sd.name_off.p, R * sizeof(u64)));
            HIP_TRY(copy_between(e0, S0.m_names.as<char>() + nb[d], E[d], Sd.names.p, (size_t)ro[d].name_bytes));
            if (R)
                hipLaunchKernelGGL(yk::gm_shift_kernel, dim3((u32)((R + 255) / 256)), dim3(256), 0, e0->stream, S0.m_fp.as<u64>() + eb[d],
                                   S0.m_noff.as<u64>() + eb[d], (u64)R, 2 * B[d], nb[d]);
        }
        HIP_TRY(hipMemcpyAsync(S0.m_noff.as<u64>() + M, &NB, sizeof(u64), hipMemcpyHostToDevice, e0->stream));
        HIP_TRY(hipMemsetAsync(S0.ctl.p, 0, 64, e0->stream));
        HIP_TRY(hipMemsetAsync(S0.g_claim.p, 0, (size_t)capG * sizeof(u64), e0->stream));
        HIP_TRY(hipMemsetAsync(S0.g_fp.p, 0xFF, (size_t)capG * sizeof(u64), e0->stream));
        HIP_TRY(hipMemsetAsync(S0.g_cnt.p, 0, (size_t)capG * sizeof(u32), e0->stream));
        yk::GmArgs gm{};
        gm.names = S0.m_names.as<unsigned char>(), gm.noff = S0.m_noff.as<u64>(), gm.fp = S0.m_fp.as<u64>();
        gm.len = S0.m_len.as<u32>(), gm.cnt = S0.m_cnt.as<u32>(), gm.M = M;
        gm.claim = S0.g_claim.as<u64>(), gm.first_pos = S0.g_fp.as<u64>(), gm.slot_cnt = S0.g_cnt.as<u32>(), gm.mask = (u32)(capG - 1);
        gm.entry_slot = S0.m_slot.as<u32>();
        gm.status = reinterpret_cast<u32 *>(S0.ctl.as<unsigned long long>() + 2);
        u32 *d_nreads = reinterpret_cast<u32 *>(S0.ctl.as<unsigned long long>() + 3);
        const u32 mg = (u32)((M + 255) / 256);
        if (M) hipLaunchKernelGGL(yk::gm_intern_kernel, dim3(mg), dim3(256), 0, e0->stream, gm);
        HIP_TRY(S0.g_keys.reserve((size_t)capG * sizeof(u64) + 64));
        HIP_TRY(S0.g_slots.reserve((size_t)capG * sizeof(u32) + 64));
        hipLaunchKernelGGL(yk::gp_collect_kernel, dim3((u32)std::min<u64>((capG + 255) / 256, (u64)e0->num_cu * 8)), dim3(256), 0, e0->stream,
                           gm.claim, gm.first_pos, (u32)capG, S0.g_keys.as<u64>(), S0.g_slots.as<u32>(), d_nreads);
        unsigned long long h_ctl[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(h_ctl, S0.ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, e0->stream));
        HIP_TRY(hipStreamSynchronize(e0->stream));
        HIP_TRY(hipGetLastError());
        if ((u32)h_ctl[2]) return fail(YACRD_EINTERNAL, "device parser: the merge table overflowed");
        Rg = (u32)h_ctl[3];
        HIP_TRY(S0.g_keys2.reserve((size_t)Rg * sizeof(u64) + 64));
        HIP_TRY(S0.g_slots2.reserve((size_t)Rg * sizeof(u32) + 64));
        if (Rg) {
            const int rcs = sort_by_first_position(e0, S0.g_keys.as<u64>(), S0.g_keys2.as<u64>(), S0.g_slots.as<u32>(), S0.g_slots2.as<u32>(), Rg, 2 * n + 2, S0.tmp, S0.part);
            if (rcs) return rcs;
        }
        HIP_TRY(S0.g_rcnt.reserve((size_t)(Rg + 4) * sizeof(u32)));
        HIP_TRY(S0.g_nlen.reserve((size_t)(Rg + 4) * sizeof(u32)));
        HIP_TRY(S0.g_nat.reserve((size_t)(Rg + 1) * sizeof(u64)));
        HIP_TRY(S0.g_noff.reserve((size_t)(Rg + 2) * sizeof(u64)));
        HIP_TRY(S0.g_len.reserve((size_t)(Rg + 1) * sizeof(u32)));
        const u32 rg = (Rg + 255) / 256;
        if (Rg) {
            hipLaunchKernelGGL(yk::gm_number_kernel, dim3(rg), dim3(256), 0, e0->stream, gm, S0.g_slots2.as<u32>(), Rg, S0.g_slot_read.as<u32>(),
                               S0.g_rcnt.as<u32>(), S0.g_nlen.as<u32>(), S0.g_nat.as<u64>());
            hipLaunchKernelGGL(yk::gm_entry_kernel, dim3(mg), dim3(256), 0, e0->stream, gm, S0.g_slot_read.as<u32>(), S0.m_read.as<u32>(),
                               S0.g_len.as<u32>());
        }
        if (const int rcs = scan_u32_to_u64(e0, S0.g_nlen.as<u32>(), (u64)Rg, S0.g_noff.as<u64>(), S0.part)) return rcs;
        HIP_TRY(hipMemcpyAsync(&g_name_bytes, S0.g_noff.as<u64>() + Rg, sizeof(u64), hipMemcpyDeviceToHost, e0->stream));
        HIP_TRY(hipStreamSynchronize(e0->stream));
        HIP_TRY(S0.g_names.reserve((size_t)g_name_bytes + 64));
        if (Rg)
            hipLaunchKernelGGL(yk::gp_names_kernel, dim3(rg), dim3(256), 0, e0->stream, gm.names, S0.g_nat.as<u64>(), S0.g_noff.as<u64>(), Rg,
                               S0.g_names.as<unsigned char>());
        // the reads, to the host
        reads->n_reads = Rg;
        reads->n_records = 0;
        for (uint32_t d = 0; d < A; d++) reads->n_records += ro[d].n_recs;
        reads->lengths = (uint32_t *)std::malloc(((size_t)Rg + 1) * sizeof(uint32_t));
        reads->name_off = (uint64_t *)std::malloc(((size_t)Rg + 1) * sizeof(uint64_t));
        reads->names = (char *)std::malloc((size_t)g_name_bytes + 1);
        h_rcnt.resize((size_t)Rg + 1);
        if (!reads->lengths || !reads->name_off || !reads->names) return fail(YACRD_ENOMEM, "host allocation failed");
        if (Rg) HIP_TRY(hipMemcpyAsync(reads->lengths, S0.g_len.p, (size_t)Rg * sizeof(u32), hipMemcpyDeviceToHost, e0->stream));
        HIP_TRY(hipMemcpyAsync(reads->name_off, S0.g_noff.p, ((size_t)Rg + 1) * sizeof(u64), hipMemcpyDeviceToHost, e0->stream));
        if (g_name_bytes) HIP_TRY(hipMemcpyAsync(reads->names, S0.g_names.p, (size_t)g_name_bytes, hipMemcpyDeviceToHost, e0->stream));
        if (Rg) HIP_TRY(hipMemcpyAsync(h_rcnt.data(), S0.g_rcnt.p, (size_t)Rg * sizeof(u32), hipMemcpyDeviceToHost, e0->stream));
        HIP_TRY(hipStreamSynchronize(e0->stream));
    }
    // ---- who sweeps what: contiguous ranges of read numbers with about the same number of intervals each (every engine,
    // also one that had no text to parse)
    std::vector<u32> cut(N + 1, Rg);
    {
        u64 total = 0;
        for (u32 g = 0; g < Rg; g++) total += h_rcnt[g];
        u64 run = 0;
        uint32_t o = 1;
        cut[0] = 0;
        for (u32 g = 0; g < Rg && o < N; g++) {
            while (o < N && run >= (total * o + N - 1) / N) cut[o++] = g;
            run += h_rcnt[g];
        }
        // (cut[o..N] stay Rg: engines behind the last cut own nothing)
    }
    // ---- every parsing engine: its records from table slots to the file's read numbers
    rc = run_all(A, [&](uint32_t d) -> int {
        yacrd_engine *e = E[d];
        DeviceGuard guard(e->device);
        Scratch &S = *scratch_of(e);
        const size_t R = ro[d].R;
        HIP_TRY(S.gmap.reserve((size_t)(std::max<u64>(R, Rg) + 4) * sizeof(u32)));
        HIP_TRY(copy_between(e, S.gmap.p, e0, S0.m_read.as<u32>() + eb[d], R * sizeof(u32)));
        hipLaunchKernelGGL(yk::gm_slot_read_kernel, dim3((u32)((ro[d].cap + 255) / 256)), dim3(256), 0, e->stream, S.map.as<u32>(), ro[d].cap,
                           S.gmap.as<u32>());
        if (ro[d].n_recs)
            hipLaunchKernelGGL(yk::gm_rewrite_kernel, dim3((u32)std::min<u64>((ro[d].n_recs + 255) / 256, (u64)e->num_cu * 16)), dim3(256), 0,
                               e->stream, S.recs.as<yk::OvlRec>(), ro[d].n_recs, S.map.as<u32>());
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        return YACRD_OK;
    });
    if (rc) return rc;
    const double t_merged = now_ms();

    // ---- every engine: the records of ALL ranges (its own in place, the others' copied over when they live on another
    // device), the halves that name its reads kept (csr_build.h: YACRD_HANDLE_ELSEWHERE), CSR, sweep
    std::vector<yacrd_result> parts(N);
    std::vector<double> t_built(N, 0), t_ran(N, 0);
    rc = run_all(N, [&](uint32_t o) -> int {
        yacrd_engine *e = E[o];
        DeviceGuard guard(e->device);
        Scratch &S = *scratch_of(e);
        const u32 lo = cut[o], hi = cut[o + 1], Ro = hi - lo;
        std::vector<RecSlab> slabs(A);
        u64 foreign = 0;
        // (records parsed by an engine of THIS device are read where they lie; under YACRD_TEST_FORCE_PEER_COPY every other
        // engine counts as another device's, so that the gather and its copies run on a one-GPU box)
        auto elsewhere = [&](uint32_t d) { return E[d]->device != e->device || (forced_peer_route() >= 0 && E[d] != e); };
        for (uint32_t d = 0; d < A; d++)
            if (elsewhere(d)) foreign += ro[d].n_recs;
        HIP_TRY(S.gather.reserve((size_t)foreign * sizeof(yk::OvlRec) + 64));
        u64 at = 0;
        for (uint32_t d = 0; d < A; d++) {
            Scratch &Sd = *scratch_of(E[d]);
            if (!elsewhere(d)) {
                slabs[d] = RecSlab{Sd.recs.as<yk::OvlRec>(), ro[d].n_recs};
            } else {
                yk::OvlRec *dst = S.gather.as<yk::OvlRec>() + at;
                HIP_TRY(copy_between(e, dst, E[d], Sd.recs.p, (size_t)ro[d].n_recs * sizeof(yk::OvlRec)));
                slabs[d] = RecSlab{dst, ro[d].n_recs};
                at += ro[d].n_recs;
            }
        }
        HIP_TRY(S.gmap.reserve((size_t)(Rg + 4) * sizeof(u32)));
        if (Rg) hipLaunchKernelGGL(yk::gm_own_kernel, dim3((Rg + 255) / 256), dim3(256), 0, e->stream, S.gmap.as<u32>(), Rg, lo, hi);
        HIP_TRY(e->in_len.reserve((size_t)(Ro + 1) * sizeof(u32)));
        HIP_TRY(copy_between(e, e->in_len.p, e0, S0.g_len.as<u32>() + lo, (size_t)Ro * sizeof(u32)));
        // (the merge has summed every read's intervals over the ranges: the CSR build's count pass — one more pass over
        // every range's records on every engine — is not needed)
        HIP_TRY(S.cnt.reserve((size_t)(Ro + 4) * sizeof(u32)));
        HIP_TRY(copy_between(e, S.cnt.p, e0, S0.g_rcnt.as<u32>() + lo, (size_t)Ro * sizeof(u32)));
        u64 n_iv = 0, own_iv = 0; // (the engine's CSR holds the intervals of ITS reads: the merge has their counts)
        for (u32 g = lo; g < hi; g++) own_iv += h_rcnt[g];
        int r = csr_from_records(e, slabs.data(), slabs.size(), S.gmap.as<u32>(), Rg, Ro, S.cnt, S.part, S.err, nullptr, &n_iv, true,
                                 own_iv + 1);
        if (r) return r;
        t_built[o] = now_ms();
        r = run_on_device(e, e->in_off.as<u64>(), e->in_iv.as<uint2>(), e->in_len.as<u32>(), Ro, n_iv, coverage, not_coverage);
        if (r) return r;
        t_ran[o] = now_ms();
        return fetch_result(e, &parts[o]);
    });
    auto drop = [&]() {
        for (auto &r : parts) yacrd_result_free(&r);
    };
    if (rc) {
        drop();
        return rc;
    }
    // ---- the results, end to end: the ranges are in first-appearance order already
    u64 G = 0;
    for (auto &r : parts) G += r.n_regions;
    out->bad_offsets = (uint64_t *)std::malloc(((size_t)Rg + 1) * sizeof(uint64_t));
    out->bad_regions = (uint32_t *)std::malloc((size_t)(2 * G + 2) * sizeof(uint32_t));
    out->read_type = (uint8_t *)std::malloc((size_t)Rg + 1);
    if (!out->bad_offsets || !out->bad_regions || !out->read_type) {
        drop();
        yacrd_result_free(out);
        return fail(YACRD_ENOMEM, "host allocation failed");
    }
    u64 g0 = 0;
    for (uint32_t o = 0; o < N; o++) {
        const u32 lo = cut[o], Ro = cut[o + 1] - lo;
        if (parts[o].n_reads != Ro) {
            drop();
            yacrd_result_free(out);
            return fail(YACRD_EINTERNAL, "device parser: an engine returned another number of reads than it was given");
        }
        for (u32 l = 0; l < Ro; l++) out->bad_offsets[lo + l] = g0 + parts[o].bad_offsets[l];
        if (parts[o].n_regions) std::memcpy(out->bad_regions + 2 * g0, parts[o].bad_regions, (size_t)parts[o].n_regions * 2 * sizeof(uint32_t));
        if (Ro) std::memcpy(out->read_type + lo, parts[o].read_type, Ro);
        g0 += parts[o].n_regions;
    }
    out->bad_offsets[Rg] = g0;
    out->n_reads = Rg;
    out->n_regions = g0;
    drop();
    if (stats) {
        stats->text_bytes = n;
        stats->n_records = reads->n_records;
        stats->n_reads = Rg;
        double tt = 0, tp = 0;
        for (uint32_t d = 0; d < A; d++) tt = std::max(tt, ro[d].t_text - ro[d].t_start), tp = std::max(tp, ro[d].t_parse - ro[d].t_text);
        stats->text_ms = (float)tt;
        stats->parse_ms = (float)std::max(0.0, (t_parsed - t0) - tt);
        (void)tp;
        double tb = 0, tr = 0;
        for (uint32_t o = 0; o < N; o++) tb = std::max(tb, t_built[o] - t_merged), tr = std::max(tr, t_ran[o] - t_built[o]);
        stats->build_ms = (float)((t_merged - t_parsed) + tb);
        stats->run_ms = (float)tr;
        stats->d2h_ms = (float)std::max(0.0, now_ms() - t_merged - tb - tr);
    }
    return YACRD_OK;
}
} // namespace

extern "C" {

/* tests: cross-engine copies since the library was loaded, by route — same device / hipMemcpyPeerAsync / staged through the host */
void yacrd_debug_peer_copy_counts(uint64_t out[3])
{
    for (int i = 0; i < 3; i++) out[i] = g_peer_copies[i].load();
}

static int group_args(yacrd_engine *const *engines, uint32_t n_engines, yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    if (!engines || !n_engines || !out || !reads) return fail(YACRD_EINVAL, "null argument");
    for (uint32_t d = 0; d < n_engines; d++) {
        if (!engines[d]) return fail(YACRD_EINVAL, "engine is null");
        if (engines[d]->pending.active || engines[d]->host_pending) return fail(YACRD_EINVAL, "an engine has a submitted batch pending");
        for (uint32_t k = 0; k < d; k++)
            if (engines[k] == engines[d]) return fail(YACRD_EINVAL, "the same engine twice");
    }
    std::memset(out, 0, sizeof(*out));
    std::memset(reads, 0, sizeof(*reads));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return YACRD_OK;
}

int yacrd_engines_ingest_overlaps(yacrd_engine *const *engines, uint32_t n_engines, const char *path, int format, int n_threads,
                                  uint32_t coverage, double not_coverage, yacrd_result *out, yacrd_reads *reads, yacrd_ingest_stats *stats)
{
    if (const int rca = group_args(engines, n_engines, out, reads, stats)) return rca;
    if (n_engines == 1) return yacrd_engine_ingest_overlaps(engines[0], path, format, n_threads, coverage, not_coverage, out, reads, stats);
    if (!path) return fail(YACRD_EINVAL, "null argument");
    bool m4 = false;
    if (const int rcf = ingest_format(path, format, m4)) return rcf;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(YACRD_EINVAL, std::string("cannot open ") + path);
    struct FdGuard {
        int fd;
        ~FdGuard() { ::close(fd); }
    } fdg{fd};
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return fail(YACRD_EFALLBACK, "not a regular file: the host parser reads it");
    if (is_compressed_magic(fd))
        return fail(YACRD_EFALLBACK, "a compressed file: inflate it (yacrd_text_from_file + yacrd_engines_ingest_overlaps_mem) or take the host parser");
    TextSource src;
    src.fd = fd;
    const int rc = ingest_text_group(engines, n_engines, src, (u64)st.st_size, m4, n_threads, coverage, not_coverage, out, reads, stats);
    if (rc) yacrd_reads_free(reads);
    return rc;
}

int yacrd_engines_ingest_overlaps_mem(yacrd_engine *const *engines, uint32_t n_engines, const char *text, uint64_t n, int format,
                                      int n_threads, uint32_t coverage, double not_coverage, yacrd_result *out, yacrd_reads *reads,
                                      yacrd_ingest_stats *stats)
{
    if (const int rca = group_args(engines, n_engines, out, reads, stats)) return rca;
    if (n_engines == 1)
        return yacrd_engine_ingest_overlaps_mem(engines[0], text, n, format, n_threads, coverage, not_coverage, out, reads, stats);
    if (!text && n) return fail(YACRD_EINVAL, "null argument");
    bool m4 = false;
    if (const int rcf = ingest_format(nullptr, format, m4)) return rcf;
    TextSource src;
    src.mem = text ? text : "";
    const int rc = ingest_text_group(engines, n_engines, src, n, m4, n_threads, coverage, not_coverage, out, reads, stats);
    if (rc) yacrd_reads_free(reads);
    return rc;
}

/* include/yacrd_engine_debug.h: the device parser's sort on its own (tests) */
int yacrd_debug_sort_pairs(yacrd_engine *e, uint64_t *keys, uint32_t *vals, uint64_t n, uint64_t key_bound)
{
    if (!e || (n && (!keys || !vals))) return fail(YACRD_EINVAL, "null argument");
    if (n >= 0x7FFFFFFFull) return fail(YACRD_EINVAL, "too many pairs");
    DeviceGuard guard(e->device);
    DevBuf k0, k1, v0, v1, tmp, part;
    auto body = [&]() -> int {
        HIP_TRY(k0.reserve((size_t)n * sizeof(u64) + 64));
        HIP_TRY(k1.reserve((size_t)n * sizeof(u64) + 64));
        HIP_TRY(v0.reserve((size_t)n * sizeof(u32) + 64));
        HIP_TRY(v1.reserve((size_t)n * sizeof(u32) + 64));
        if (!n) return YACRD_OK;
        HIP_TRY(hipMemcpyAsync(k0.p, keys, (size_t)n * sizeof(u64), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(v0.p, vals, (size_t)n * sizeof(u32), hipMemcpyHostToDevice, e->stream));
        if (const int rcs = sort_by_first_position(e, k0.as<u64>(), k1.as<u64>(), v0.as<u32>(), v1.as<u32>(), (u32)n, key_bound, tmp, part)) return rcs;
        HIP_TRY(hipMemcpyAsync(keys, k1.p, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(vals, v1.p, (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        return YACRD_OK;
    };
    const int rc = body();
    for (DevBuf *b : {&k0, &k1, &v0, &v1, &tmp, &part}) b->release();
    return rc;
}

} // extern "C"
