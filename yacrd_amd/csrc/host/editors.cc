// editors.cc — post-detection operations on sequence / overlap files and the .yacrd re-reader.
//
// Reference (paths relative to the reference root):
//   scrubb   src/editor/scrubbing.rs:33-236    split    src/editor/split.rs:33-226
//   filter   src/editor/filter.rs:32-228       extract  src/editor/extract.rs:32-232
//   FromReport (a .yacrd given as -i)          src/stack.rs:176-257
// The editors consume `BadPart::get_bad_part` (unknown id -> (vec![], 0), src/stack.rs:164-169) and
// the read type.  The reference recomputes type_of_read per record (e.g. scrubbing.rs:181); here
// the type comes from the engine (GPU kernel #2), an unknown read being NotBad by construction
// (0/0 = NaN > n is false, editor/mod.rs:88).  Pure byte shuffling, I/O bound: host C++.
//
// Sequence formats follow what the reference's golden files pin for noodles 0.84: FASTQ
// "@name[ description]\nseq\n+\nqual\n"; FASTA ">name[ description]\n" + sequence wrapped at 80
// columns — the published default of the third-party writer the reference calls with no options
// (noodles::fasta::Writer::new, src/editor/scrubbing.rs:84; Cargo.lock pins noodles 0.84.0 /
// noodles-fasta 0.45.0, whose writer builder defaults to 80 bases a line; the crate is not vendored in
// /root/reference and the reference's own fixtures hold no FASTA sequence beyond 22 bases).
#include "../../../include/yacrd_host.h"
#include "host_common.h"

#include "codec.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

enum FileType { FT_NONE, FT_FASTA, FT_FASTQ, FT_YACRD, FT_PAF, FT_M4, FT_YOVL };

// src/util.rs:39-55 get_file_type: substring match, in this priority
FileType file_type(const std::string &n)
{
    auto has = [&](const char *s) { return n.find(s) != std::string::npos; };
    if (has(".m4") || has(".mhap")) return FT_M4;
    if (has(".paf")) return FT_PAF;
    if (has(".yacrd")) return FT_YACRD;
    if (has(".fastq") || has(".fq")) return FT_FASTQ;
    if (has(".fasta") || has(".fa")) return FT_FASTA;
    if (has(".yovl")) return FT_YOVL;
    return FT_NONE;
}
const char *type_name(FileType t)
{
    switch (t) {
    case FT_FASTA: return "fasta";
    case FT_FASTQ: return "fastq";
    case FT_YACRD: return "yacrd";
    case FT_PAF: return "paf";
    case FT_M4: return "m4";
    default: return "yacrd overlap";
    }
}

// ---- input: plain, gzip, bzip2 or xz, sniffed from magic bytes like niffler (src/util.rs:57-70) --
struct Reader {
    yh::InStream in;
    std::vector<char> own;
    const char *buf = nullptr; // the bytes line() / peek() look at: `own` (a stream's window) or a range of a mapped file
    size_t pos = 0, end = 0;
    bool eof = false;
    bool failed = false; // read / decompression error (message in the error slot): callers check

    int open(const char *path)
    {
        if (in.open(path)) return 1;
        own.resize(1 << 20);
        buf = own.data();
        return 0;
    }
    // a range of memory as the whole input (the multi-threaded editors: one chunk of a mapped file)
    bool memory = false;
    void open_memory(const char *p, size_t n)
    {
        memory = true;
        buf = p;
        pos = 0;
        end = n;
        eof = true; // nothing behind it
    }
    yh::Compression compression() const { return in.format(); }
    bool fill()
    {
        if (eof) return false;
        const long n = in.read(own.data(), own.size());
        if (n <= 0) {
            eof = true;
            failed = n < 0; // a truncated or corrupt stream is an error, not the end of the file
            return false;
        }
        pos = 0;
        end = (size_t)n;
        return true;
    }
    // next line without its terminator; false at end of input
    bool line(std::string &out)
    {
        out.clear();
        bool any = false;
        for (;;) {
            if (pos == end && !fill()) break;
            any = true;
            const char *p = buf + pos;
            const char *nl = (const char *)std::memchr(p, '\n', end - pos);
            if (nl) {
                out.append(p, (size_t)(nl - p));
                pos = (size_t)(nl - buf) + 1;
                return true;
            }
            out.append(p, end - pos);
            pos = end;
        }
        return any && !out.empty();
    }
    int peek()
    {
        if (pos == end && !fill()) return -1;
        return (unsigned char)buf[pos];
    }
};

// ---- output: same compression as the input, level 1 (src/util.rs:72-87) ------------------------
struct Writer {
    yh::OutStream os;
    std::vector<char> buf;
    bool failed = false;
    bool to_memory = false; // everything stays in `buf` (the multi-threaded editors: one chunk's output)
    int open(const char *path, yh::Compression fmt)
    {
        if (os.open(path, fmt)) return 1;
        buf.reserve(1 << 20);
        return 0;
    }
    void open_memory(size_t expect)
    {
        to_memory = true;
        buf.reserve(expect);
    }
    void flush()
    {
        if (buf.empty() || to_memory) return;
        failed |= !os.write(buf.data(), buf.size());
        buf.clear();
    }
    void put(const char *p, size_t n)
    {
        buf.insert(buf.end(), p, p + n);
        if (!to_memory && buf.size() > (1u << 20) - 4096) flush();
    }
    void put(const std::string &s) { put(s.data(), s.size()); }
    void put(char c) { put(&c, 1); }
    int close()
    {
        flush();
        const int rc = os.close();
        return (failed || rc) ? yh::fail("Error during writing of the output file") : 0;
    }
};

// ---- BadPart lookups ---------------------------------------------------------------------------
// Name -> read index.  Open addressing over read indices (the names stay where the view has them: no copies), built
// by every usable CPU with one CAS per read: a std::unordered_map<std::string, u32> of configs[4]'s 5 M reads took
// longer to build than the GPU needs for the whole detection.  Equal names: the smallest index wins (what
// emplace() in read order did).
struct BadParts {
    const yacrd_badparts_view *v;
    std::vector<std::atomic<uint32_t>> slots;
    uint64_t mask = 0;
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    bool same(uint32_t r, const char *p, size_t n) const
    {
        const uint64_t a = v->name_off[r], b = v->name_off[r + 1];
        return b - a == n && std::memcmp(v->names + a, p, n) == 0;
    }
    explicit BadParts(const yacrd_badparts_view *view, unsigned n_threads = 0) : v(view)
    {
        uint64_t cap = 16;
        while (cap < 2 * v->n_reads + 16) cap <<= 1;
        slots = std::vector<std::atomic<uint32_t>>(cap);
        mask = cap - 1;
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_threads ? n_threads : yh::usable_cpus(), v->n_reads / 65536 + 1));
        auto work = [&](unsigned t) {
            for (uint64_t i = (cap * t) / T, e = (cap * (t + 1)) / T; i < e; i++) slots[i].store(kEmpty, std::memory_order_relaxed);
        };
        auto fill = [&](unsigned t) {
            const uint64_t r0 = v->n_reads * t / T, r1 = v->n_reads * (t + 1) / T;
            for (uint64_t r = r0; r < r1; r++) {
                const char *p = v->names + v->name_off[r];
                const size_t n = (size_t)(v->name_off[r + 1] - v->name_off[r]);
                for (uint64_t s = yh::hash_bytes(p, n) & mask;; s = (s + 1) & mask) {
                    uint32_t cur = slots[s].load(std::memory_order_acquire);
                    if (cur == kEmpty && slots[s].compare_exchange_strong(cur, (uint32_t)r, std::memory_order_acq_rel)) break;
                    // (cur now holds the slot's owner)
                    if (same(cur, p, n)) { // a duplicate name: keep the smaller index
                        while (cur > (uint32_t)r && !slots[s].compare_exchange_weak(cur, (uint32_t)r, std::memory_order_acq_rel)) {
                        }
                        break;
                    }
                }
            }
        };
        auto run = [&](auto &&fn) {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t);
            fn(0u);
            for (auto &x : th) x.join();
        };
        run(work);
        run(fill);
    }
    // get_bad_part + type: unknown id -> (empty, 0, NotBad)
    void get(const std::string &id, const uint32_t *&reg, size_t &n, uint32_t &len, int &type) const
    {
        get(id.data(), id.size(), reg, n, len, type);
    }
    void get(const char *id, size_t id_len, const uint32_t *&reg, size_t &n, uint32_t &len, int &type) const
    {
        uint32_t r = kEmpty;
        for (uint64_t s = yh::hash_bytes(id, id_len) & mask;; s = (s + 1) & mask) {
            const uint32_t cur = slots[s].load(std::memory_order_relaxed);
            if (cur == kEmpty) break;
            if (same(cur, id, id_len)) {
                r = cur;
                break;
            }
        }
        if (r == kEmpty) {
            reg = nullptr;
            n = 0;
            len = 0;
            type = 0;
            return;
        }
        reg = v->bad_regions + 2 * v->bad_offsets[r];
        n = (size_t)(v->bad_offsets[r + 1] - v->bad_offsets[r]);
        len = v->lengths[r];
        type = v->read_type[r];
    }
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\x0c' || c == '\r'; }

struct SeqRecord {
    std::string name, desc, seq, qual;
};

void write_fasta(Writer &w, const std::string &name, const std::string &desc, const char *seq,
                 size_t n)
{
    w.put('>');
    w.put(name);
    if (!desc.empty()) {
        w.put(' ');
        w.put(desc);
    }
    w.put('\n');
    for (size_t i = 0; i < n; i += 80) {
        w.put(seq + i, std::min<size_t>(80, n - i));
        w.put('\n');
    }
}

void split_definition(const std::string &line, bool any_ws, std::string &name, std::string &desc)
{ // line without the leading '@' / '>'
    size_t i = 0;
    while (i < line.size() && !(any_ws ? is_ws(line[i]) : line[i] == ' ')) i++;
    name.assign(line, 0, i);
    desc.assign(line, i < line.size() ? i + 1 : i, std::string::npos);
}

bool next_fastq(Reader &r, SeqRecord &rec, std::string &err)
{
    std::string l1, plus;
    for (;;) {
        if (!r.line(l1)) return false;
        if (!l1.empty()) break;
    }
    if (!l1.empty() && l1.back() == '\r') l1.pop_back();
    if (l1.empty() || l1[0] != '@') {
        err = "Reading of the file in fastq format failed";
        return false;
    }
    split_definition(l1.substr(1), false, rec.name, rec.desc);
    rec.seq.clear();
    rec.qual.clear();
    const bool got_seq = r.line(rec.seq);
    const bool got_plus = r.line(plus);
    const bool got_qual = r.line(rec.qual);
    if (!rec.seq.empty() && rec.seq.back() == '\r') rec.seq.pop_back();
    if (!rec.qual.empty() && rec.qual.back() == '\r') rec.qual.pop_back();
    if (!got_seq || !got_plus || plus.empty() || plus[0] != '+' || (!got_qual && !rec.seq.empty()) ||
        rec.seq.size() != rec.qual.size()) {
        err = "Reading of the file in fastq format failed";
        return false;
    }
    return true;
}

bool next_fasta(Reader &r, SeqRecord &rec, std::string &err)
{
    std::string l;
    for (;;) {
        if (!r.line(l)) return false;
        if (!l.empty()) break;
    }
    if (!l.empty() && l.back() == '\r') l.pop_back();
    if (l.empty() || l[0] != '>') {
        err = "Reading of the file in fasta format failed";
        return false;
    }
    split_definition(l.substr(1), true, rec.name, rec.desc);
    rec.seq.clear();
    for (;;) {
        const int c = r.peek();
        if (c < 0 || c == '>') break;
        if (!r.line(l)) break;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        rec.seq += l;
    }
    return true;
}

enum Op { OP_SCRUBB = 0, OP_FILTER = 1, OP_EXTRACT = 2, OP_SPLIT = 3 };
const char *op_name(int op)
{
    static const char *n[] = {"scrubbing", "filter", "extract", "split"};
    return n[op];
}

// cut positions: scrubbing.rs:195-209 (every region) / split.rs:189-198 (middle regions only)
void cut_positions(int op, const uint32_t *reg, size_t n, uint32_t len, std::vector<uint32_t> &poss,
                   size_t &first)
{
    poss.clear();
    poss.push_back(0);
    first = 0;
    if (op == OP_SCRUBB) {
        for (size_t i = 0; i < n; i++) {
            poss.push_back(reg[2 * i]);
            poss.push_back(reg[2 * i + 1]);
        }
        if (poss.back() != len) poss.push_back(len);
        if (poss.size() >= 2 && poss[0] == 0 && poss[1] == 0) first = 2;
        // chunks_exact(2): a trailing odd element is ignored
    } else {
        for (size_t i = 0; i < n; i++) {
            if (reg[2 * i] == 0 || reg[2 * i + 1] == len) continue;
            poss.push_back(reg[2 * i]);
            poss.push_back(reg[2 * i + 1]);
        }
        poss.push_back(len);
    }
}

// A record as views (into a chunk of the mapped file, or into a SeqRecord's strings).
struct RecView {
    const char *name, *desc, *seq, *qual;
    size_t name_len, desc_len, n;
    const char *raw = nullptr; // the whole record as it stands in the input, when that IS what write_fastq would write
    size_t raw_len = 0;
};

// The next FASTQ record of a memory range WITHOUT copying it, when it is in the canonical form the writer produces —
// "@name[ desc]\nseq\n+\nqual\n", no CR, a bare '+' line, a description that is not empty when a blank follows the
// name — which is what every FASTQ this tool has written, and nearly every one it reads, looks like.  false = not at
// such a record (blank lines, CRLF, "+name" lines, the file's last line without a newline, anything malformed): the
// caller takes next_fastq, which owns the reference's rules and errors.
bool fast_fastq(Reader &r, RecView &v)
{
    if (!r.memory) return false;
    const char *p = r.buf + r.pos, *end = r.buf + r.end;
    if (p >= end || *p != '@') return false;
    const char *l1 = (const char *)std::memchr(p, '\n', (size_t)(end - p));
    if (!l1 || l1 == p + 1 || l1[-1] == '\r') return false;
    const char *seq = l1 + 1;
    const char *l2 = seq < end ? (const char *)std::memchr(seq, '\n', (size_t)(end - seq)) : nullptr;
    if (!l2 || (l2 > seq && l2[-1] == '\r')) return false;
    const char *plus = l2 + 1;
    if (plus + 1 >= end || plus[0] != '+' || plus[1] != '\n') return false;
    const char *qual = plus + 2;
    const char *l4 = qual < end ? (const char *)std::memchr(qual, '\n', (size_t)(end - qual)) : nullptr;
    if (!l4 || (l4 > qual && l4[-1] == '\r') || l4 - qual != l2 - seq) return false;
    const char *sp = (const char *)std::memchr(p + 1, ' ', (size_t)(l1 - p - 1));
    if (sp && sp + 1 == l1) return false; // "name " : the writer drops the blank
    v.name = p + 1;
    v.name_len = (size_t)((sp ? sp : l1) - (p + 1));
    v.desc = sp ? sp + 1 : l1;
    v.desc_len = sp ? (size_t)(l1 - sp - 1) : 0;
    v.seq = seq;
    v.qual = qual;
    v.n = (size_t)(l2 - seq);
    v.raw = p;
    v.raw_len = (size_t)(l4 + 1 - p);
    r.pos = (size_t)(l4 + 1 - r.buf);
    return true;
}

void put_fastq(Writer &w, const char *name, size_t name_len, const char *suffix, size_t suffix_len, const char *desc, size_t desc_len,
               const char *seq, const char *qual, size_t n)
{
    w.put('@');
    w.put(name, name_len);
    w.put(suffix, suffix_len);
    if (desc_len) {
        w.put(' ');
        w.put(desc, desc_len);
    }
    w.put('\n');
    w.put(seq, n);
    w.put("\n+\n", 3);
    w.put(qual, n);
    w.put('\n');
}

int edit_sequences(int op, bool fastq, Reader &in, Writer &out, const BadParts &bp)
{
    SeqRecord rec;
    std::string err;
    std::vector<uint32_t> poss;
    char suffix[48];
    for (;;) {
        RecView v;
        if (!(fastq && fast_fastq(in, v))) {
            const bool ok = fastq ? next_fastq(in, rec, err) : next_fasta(in, rec, err);
            if (!ok) {
                if (in.failed) return 1; // message set by the decoder
                if (!err.empty()) return yh::fail(err);
                break;
            }
            v = RecView{rec.name.data(), rec.desc.data(), rec.seq.data(), rec.qual.data(), rec.name.size(), rec.desc.size(), rec.seq.size()};
        }
        // FASTQ: first whitespace token of the name (scrubbing.rs:174-179); FASTA: the name
        size_t key_len = v.name_len;
        if (fastq) {
            key_len = 0;
            while (key_len < v.name_len && !is_ws(v.name[key_len])) key_len++;
        }
        const uint32_t *reg;
        size_t n;
        uint32_t len;
        int type;
        bp.get(v.name, key_len, reg, n, len, type);
        auto copy = [&]() {
            if (v.raw) out.put(v.raw, v.raw_len);
            else if (fastq) put_fastq(out, v.name, v.name_len, "", 0, v.desc, v.desc_len, v.seq, v.qual, v.n);
            else write_fasta(out, std::string(v.name, v.name_len), std::string(v.desc, v.desc_len), v.seq, v.n);
        };
        if (op == OP_FILTER) {
            if (type == 0) copy();
            continue;
        }
        if (op == OP_EXTRACT) {
            if (type != 0) copy();
            continue;
        }
        if (type == 2) continue; // NotCovered reads are dropped
        if (op == OP_SCRUBB ? n == 0 : type == 0) {
            copy();
            continue;
        }
        size_t first;
        cut_positions(op, reg, n, len, poss, first);
        for (size_t k = first; k + 1 < poss.size(); k += 2) {
            const uint32_t p0 = poss[k], p1 = poss[k + 1];
            if (p0 > v.n || p1 > v.n) {
                std::fprintf(stderr,
                             "[ERROR] For read %.*s %s position is larger than read, it's strange check "
                             "your data. For this read, this split position and next are ignore.\n",
                             (int)v.name_len, v.name, op == OP_SCRUBB ? "scrubb" : "split");
                break;
            }
            if (p0 > p1) // the reference panics on seq[p0..p1] with p0 > p1
                return yh::fail("bad region with begin > end while cutting read " + std::string(v.name, v.name_len));
            const int sl = std::snprintf(suffix, sizeof suffix, "_%u_%u", p0, p1);
            if (fastq)
                put_fastq(out, v.name, v.name_len, suffix, (size_t)sl, v.desc, v.desc_len, v.seq + p0, v.qual + p0, p1 - p0);
            else
                write_fasta(out, std::string(v.name, v.name_len) + suffix, std::string(), v.seq + p0, p1 - p0);
        }
    }
    return 0;
}

// ---- the same over a plain (uncompressed) file with every usable CPU ------------------------------------------
// The reference's editors are one thread reading records and writing them back (scrubbing.rs:156-236 and its three
// siblings): ~3 GB/s in this restatement, which at configs[4]'s 5 M-read FASTQ is 30 times the detection's wall
// clock.  Records are independent and their order is the input's, so the file is mapped, cut into chunks at record
// boundaries, every chunk goes through the SAME record loop as above (a Reader over the chunk's bytes, a Writer into
// memory) on its own thread, and the outputs land with pwrite at offsets that follow from the sizes of the chunks
// before — byte-identical to the one-thread output.  A boundary is a line that starts a record: '>' for FASTA; for
// FASTQ an '@' line whose line after next starts with '+' and whose fourth line is a header again (a quality line
// may begin with '@', a sequence line never begins with '+').  Anything unexpected — a chunk that fails to parse, a
// boundary that cannot be found — and the whole file is done again by the one-thread loop, which owns the reference's
// error behaviour.
size_t edit_chunk()
{ // bytes per chunk (YACRD_EDIT_CHUNK overrides: the tests cut small files into many chunks)
    if (const char *e = std::getenv("YACRD_EDIT_CHUNK"))
        if (*e) return (size_t)std::max(64ll, std::atoll(e));
    return (size_t)4 << 20; // (a chunk's output is still in the cache when its thread's turn at the file comes: 16 MB measured 6.1-6.5 GB/s
                            // of FASTQ in /dev/shm, 4 MB 7.5-8.2, 1 MB 7.8-8.1, 64 MB 6.4; profiles/r06/I_edit_turns_chunks_shm.log)
}

size_t next_line(const char *base, size_t size, size_t p)
{
    if (p >= size) return size;
    const char *nl = (const char *)std::memchr(base + p, '\n', size - p);
    return nl ? (size_t)(nl - base) + 1 : size;
}
// first record start at or after `from` (npos = none found where one should be)
size_t record_start(const char *base, size_t size, size_t from, bool fastq)
{
    size_t p = from == 0 ? 0 : next_line(base, size, from - 1);
    for (int tries = 0; p < size; tries++) {
        if (!fastq) {
            if (base[p] == '>') return p;
        } else {
            if (tries > 16) return std::string::npos;
            if (base[p] == '@') {
                const size_t l2 = next_line(base, size, next_line(base, size, p));
                if (l2 < size && base[l2] == '+') {
                    const size_t l4 = next_line(base, size, next_line(base, size, l2));
                    if (l4 >= size || base[l4] == '@' || base[l4] == '\n' || base[l4] == '\r') return p;
                }
            }
        }
        p = next_line(base, size, p);
    }
    return size;
}

// 0 = done; 1 = failed with a message; -1 = not taken (the caller runs the one-thread loop)
int edit_sequences_parallel(int op, bool fastq, const char *in_path, const char *out_path, const BadParts &bp, unsigned T)
{
    const int fd = ::open(in_path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    const size_t kEditChunk = edit_chunk();
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < 2 * kEditChunk) {
        ::close(fd);
        return -1;
    }
    const size_t size = (size_t)st.st_size;
    const char *base = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) return -1;
    struct Unmap {
        const char *p;
        size_t n;
        ~Unmap() { munmap((void *)p, n); }
    } unmap{base, size};
    (void)madvise((void *)base, size, MADV_SEQUENTIAL);
    std::vector<size_t> cut{0};
    for (size_t at = kEditChunk; at < size; at += kEditChunk) {
        const size_t b = record_start(base, size, at, fastq);
        if (b == std::string::npos) return -1;
        if (b >= size) break;
        if (b > cut.back()) cut.push_back(b);
    }
    cut.push_back(size);
    const size_t n_chunks = cut.size() - 1;
    if (n_chunks < 2) return -1;
    // The chunks land at their offsets (pwrite / a shared mapping): a regular file only.  A FIFO, /dev/stdout into a pipe
    // or a process substitution takes the one-thread loop, which streams like the reference's BufWriter — asked
    // BEFORE the open, so that a pipe's reader never sees a writer come and go.
    struct stat ost;
    if (::stat(out_path, &ost) == 0 && !S_ISREG(ost.st_mode)) return -1;
    const int ofd = ::open(out_path, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (ofd < 0) return yh::fail(std::string("cannot create ") + out_path);
    if (fstat(ofd, &ost) != 0 || !S_ISREG(ost.st_mode)) {
        ::close(ofd);
        return -1;
    }
    // The output: pwrite at the chunk's offset, ONE WRITER AT A TIME in chunk order (default, round 6) — or all at once
    // (YACRD_EDIT_OUT=pwrite, rounds 4-5's default), or memcpy into a shared mapping of the file grown ahead of the
    // writers in steps and cut to size at the end (YACRD_EDIT_OUT=map).  Buffered writes into one file take the
    // inode's lock exclusively, and the writers of a shared mapping meet at the file's page tree: tools/out_probe.cc
    // (profiles/r06/I_out_probe.log, 16 GB into one file in /dev/shm) has ONE thread's pwrite at 8.6 GB/s and two /
    // four / eight threads' at 4.1 / 4.2 / 3.3 — they hand the lock round and sleep —, the mapping at 3.3-3.9 whatever
    // the threads, MADV_POPULATE_WRITE in front of the memcpy at 3.3-4.9; on the box's disk (ext4, page cache) one
    // thread 14.5 GB/s, sixteen 12.5.  (Round 4's numbers of the same shape: pwrite 3.1 GB/s on one thread that also
    // parses, 5.0 on four, 4.4 on sixteen; the mapping 4.7 on four, 1.7 on thirty-two, profiles/r04/h_*, f_*, i_*.)
    // So the threads parse side by side and there is ONE WRITER AT A TIME, chunk after chunk in order (no thread ever
    // waits inside the kernel for the lock): see the hand-over below.
    const char *oio = std::getenv("YACRD_EDIT_OUT");
    const bool out_map = oio && std::strcmp(oio, "map") == 0;
    const bool out_turns = !out_map && !(oio && std::strcmp(oio, "pwrite") == 0);
    const size_t map_len = 2 * size + ((size_t)64 << 20);
    char *obase = nullptr;
    if (out_map) {
        obase = (char *)mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
        if (obase == MAP_FAILED) {
            ::close(ofd);
            return -1;
        }
    }
    std::mutex grow_mu;
    std::atomic<size_t> file_size(0);
    auto ensure = [&](size_t need) -> bool { // the file is at least `need` bytes long
        if (need <= file_size.load(std::memory_order_acquire)) return true;
        std::lock_guard<std::mutex> g(grow_mu);
        size_t cur = file_size.load();
        if (need <= cur) return true;
        if (need > map_len) return false;
        const size_t to = std::min(map_len, std::max(need, cur + ((size_t)2 << 30)));
        if (ftruncate(ofd, (off_t)to) != 0) return false;
        file_size.store(to, std::memory_order_release);
        return true;
    };
    // Hand-over between the threads, all of it under turn_mu.  A chunk's thread PUBLISHES its output (pointer, size);
    // chunks get their offsets in order as their predecessors' sizes become known.  With turns, whoever publishes the
    // chunk the file waits for becomes THE WRITER and stays it for as long as the next chunk is ready too — the chunks
    // of threads that are already parsing their next one (two buffers per thread) — so the file never waits for a
    // sleeping thread to be woken.  Either way the run is the writes end to end + 0.2 s (YACRD_EDIT_STATS): 20.4 GB of
    // FASTQ in /dev/shm in 2.5-3.4 s of which 2.3-3.2 are pwrite, and the writes are the faster the fewer threads parse
    // beside them (2.3 s with two threads, 2.8 with four, 3.1 with six on one box: the chunk's output leaves the cache
    // before its turn comes) — profiles/r06/I_edit_4mb_*.log, J_edit_writer_*.log.
    struct Slot {
        const char *p = nullptr;
        size_t n = 0;
        long long at = -1;
        unsigned owner = 0;
        bool ready = false, done = false;
    };
    T = (unsigned)std::max<size_t>(1, std::min<size_t>(T, n_chunks));
    std::vector<Slot> slot(n_chunks);
    size_t next_write = 0;   // the first chunk without an offset (turns: not yet in the file)
    long long at_total = 0;  // its offset
    bool writer_active = false;
    std::mutex turn_mu;
    std::vector<std::condition_variable> cv(T);
    std::atomic<size_t> next(0);
    std::atomic<int> state(0); // 1 = a chunk did not parse (retry on one thread), 2 = write error
    // The chunks' bytes come through pread into a buffer the thread keeps (YACRD_EDIT_IO=mmap: straight from the mapping,
    // which is only used to find the boundaries otherwise): sixteen threads taking minor faults on one mapping get in
    // each other's way (the parser found the same in round 2), and a buffer that lives as long as its thread is not
    // mapped, zeroed and unmapped once per chunk by the allocator.
    const char *io = std::getenv("YACRD_EDIT_IO");
    const bool use_pread = !(io && std::strcmp(io, "mmap") == 0);
    const int rfd = use_pread ? ::open(in_path, O_RDONLY) : -1;
    if (use_pread && rfd < 0) {
        ::close(ofd);
        return -1;
    }
    // YACRD_EDIT_STATS=1: seconds per phase summed over the threads, on stderr (tools/edit_bench.py)
    const char *stats_env = std::getenv("YACRD_EDIT_STATS");
    const bool stats = stats_env && *stats_env == '1';
    std::atomic<long long> ns_read(0), ns_parse(0), ns_wait(0), ns_write(0);
    auto now_ns = [&] { return stats ? (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0ll; };
    auto write_at = [&](const char *p, size_t n, long long at) {
        const long long t0 = now_ns();
        if (out_map) {
            if (n && !ensure((size_t)at + n)) state.store(state.load() == 2 ? 2 : 1); // (beyond the mapping: the one-thread loop)
            else if (n) std::memcpy(obase + at, p, n);
        } else {
            for (size_t done = 0; done < n;) {
                const ssize_t k = ::pwrite(ofd, p + done, n - done, (off_t)(at + (long long)done));
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) {
                    state.store(k < 0 && (errno == ESPIPE || errno == EINVAL) ? 1 : 2); // (not seekable after all: the one-thread loop)
                    break;
                }
                done += (size_t)k;
            }
        }
        ns_write += now_ns() - t0;
    };
    auto work = [&](unsigned t) {
        constexpr size_t kNone = ~(size_t)0;
        std::vector<char> ibuf;
        const int n_bufs = out_turns ? 2 : 1;
        Writer outs[2];
        for (int k = 0; k < n_bufs; k++) outs[k].open_memory(kEditChunk + kEditChunk / 8 + 4096);
        size_t pending[2] = {kNone, kNone}; // the chunk whose output sits in outs[k] until it is in the file
        auto wait_done = [&](size_t &j) {
            if (j == kNone) return;
            const long long t0 = now_ns();
            {
                std::unique_lock<std::mutex> g(turn_mu);
                cv[t].wait(g, [&] { return slot[j].done; });
            }
            j = kNone;
            ns_wait += now_ns() - t0;
        };
        for (int k = 0;; k = (k + 1) % n_bufs) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks) break;
            wait_done(pending[k]);
            Writer &out = outs[k];
            const size_t len = cut[i + 1] - cut[i];
            const char *src = base + cut[i];
            bool skip = state.load() != 0;
            const long long t_a = now_ns();
            if (use_pread && !skip) {
                if (ibuf.size() < len) ibuf.resize(len + len / 8);
                for (size_t got = 0; got < len;) {
                    const ssize_t k = ::pread(rfd, ibuf.data() + got, len - got, (off_t)(cut[i] + got));
                    if (k < 0 && errno == EINTR) continue;
                    if (k <= 0) {
                        state.store(1);
                        skip = true;
                        break;
                    }
                    got += (size_t)k;
                }
                src = ibuf.data();
            }
            const long long t_b = now_ns();
            Reader in;
            in.open_memory(src, len);
            out.buf.clear();
            if (!skip && edit_sequences(op, fastq, in, out, bp) != 0) state.store(1);
            const long long t_c = now_ns();
            ns_read += t_b - t_a, ns_parse += t_c - t_b;
            std::unique_lock<std::mutex> g(turn_mu);
            Slot &me = slot[i];
            me.p = out.buf.data(), me.n = (skip || state.load()) ? 0 : out.buf.size(), me.owner = t, me.ready = true;
            if (!out_turns) { // every thread writes its own chunk, as soon as its offset is known
                while (next_write < n_chunks && slot[next_write].ready) {
                    Slot &s = slot[next_write++];
                    s.at = at_total, at_total += (long long)s.n;
                    if (s.owner != t) cv[s.owner].notify_one();
                }
                cv[t].wait(g, [&] { return me.at >= 0; });
                g.unlock();
                ns_wait += now_ns() - t_c;
                write_at(me.p, me.n, me.at);
                continue;
            }
            pending[k] = i;
            if (writer_active || next_write != i) continue; // (the writer comes to it, or the thread that publishes the chunk the file waits for)
            writer_active = true;
            while (next_write < n_chunks && slot[next_write].ready) {
                Slot &s = slot[next_write];
                s.at = at_total;
                g.unlock();
                write_at(s.p, s.n, s.at);
                g.lock();
                at_total += (long long)s.n, s.done = true, next_write++;
                if (s.owner != t) cv[s.owner].notify_one();
            }
            writer_active = false;
        }
        for (int k = 0; k < n_bufs; k++) wait_done(pending[k]); // (the buffers go with the thread)
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
    work(0u);
    for (auto &x : th) x.join();
    if (stats)
        std::fprintf(stderr, "yacrd edit stats: %u threads, %zu chunks: read %.2f s, parse %.2f, waiting %.2f, write %.2f (summed over threads)\n", T, n_chunks,
                     ns_read.load() * 1e-9, ns_parse.load() * 1e-9, ns_wait.load() * 1e-9, ns_write.load() * 1e-9);
    int rc = 0;
    if (out_map) {
        const long long total = at_total;
        munmap(obase, map_len);
        if (state.load() == 0 && (total < 0 || ftruncate(ofd, (off_t)total) != 0)) rc = 1;
    }
    rc |= ::close(ofd);
    if (rfd >= 0) ::close(rfd);
    if (state.load() == 1) return -1; // (the one-thread loop truncates the output and words the error)
    if (state.load() == 2 || rc != 0) return yh::fail("Error during writing of the output file");
    return 0;
}

// filter / extract on overlap files: filter.rs:140-228, extract.rs:144-232 (csv reader, not flexible)
int edit_overlaps(int op, bool paf, Reader &in, Writer &out, const BadParts &bp)
{
    const char delim = paf ? '\t' : ' ';
    const size_t ib = paf ? 5 : 1;
    std::string l, a, b;
    size_t n_fields = 0;
    while (in.line(l)) {
        if (!l.empty() && l.back() == '\r') l.pop_back();
        if (l.empty()) continue;
        size_t nf = 1, start_b = std::string::npos, end_a = l.find(delim);
        size_t p = 0;
        for (size_t i = 0; i < l.size(); i++)
            if (l[i] == delim) {
                if (nf == ib) start_b = i + 1;
                nf++;
                (void)p;
            }
        if (n_fields == 0) n_fields = nf;
        if (nf != n_fields || nf <= ib)
            return yh::fail(paf ? "Reading of the file in paf format failed"
                                : "Reading of the file in m4 format failed");
        a.assign(l, 0, end_a);
        const size_t end_b = l.find(delim, start_b);
        b.assign(l, start_b, end_b == std::string::npos ? std::string::npos : end_b - start_b);
        const uint32_t *reg;
        size_t n;
        uint32_t len;
        int ta, tb;
        bp.get(a, reg, n, len, ta);
        bp.get(b, reg, n, len, tb);
        const bool both_good = ta == 0 && tb == 0;
        if (op == OP_FILTER ? both_good : !both_good) {
            out.put(l);
            out.put('\n');
        }
    }
    if (in.failed) return 1; // message set by the decoder
    return 0;
}

} // namespace

// ---- .yacrd re-reader (FromReport, src/stack.rs:176-257) ---------------------------------------
struct yacrd_report {
    std::vector<uint64_t> name_off{0};
    std::vector<char> names;
    std::vector<uint32_t> lengths;
    std::vector<uint64_t> bad_offsets{0};
    std::vector<uint32_t> bad_regions;
};

extern "C" {

int yacrd_report_read(const char *path, yacrd_report **out)
{
    if (!path || !out) return yh::fail("null argument");
    *out = nullptr;
    Reader in;
    if (in.open(path)) return 1;
    yacrd_report *rep = new yacrd_report();
    std::unordered_map<std::string, uint32_t> seen; // HashMap::insert: the last line for an id wins
    std::string l;
    uint64_t line_no = 0;
    auto corrupt = [&](const std::string &why) {
        delete rep;
        return yh::fail("Your yacrd file " + std::string(path) + " seems corrupt at line " +
                        std::to_string(line_no) + " (1-based; " + why + ")");
    };
    struct Row {
        std::string id;
        uint32_t len;
        std::vector<uint32_t> reg;
    };
    std::vector<Row> rows;
    while (in.line(l)) {
        line_no++; // physical lines, 1-based, blank ones included
        if (!l.empty() && l.back() == '\r') l.pop_back();
        if (l.empty()) continue;
        // type \t id \t len \t regions
        size_t t1 = l.find('\t'), t2 = t1 == std::string::npos ? t1 : l.find('\t', t1 + 1),
               t3 = t2 == std::string::npos ? t2 : l.find('\t', t2 + 1);
        if (t3 == std::string::npos) return corrupt("expected 4 tab separated columns");
        Row row;
        row.id.assign(l, t1 + 1, t2 - t1 - 1);
        uint64_t len = 0;
        {
            const std::string s(l, t2 + 1, t3 - t2 - 1);
            if (s.empty()) return corrupt("length");
            for (char c : s) {
                if (c < '0' || c > '9') return corrupt("length");
                len = len * 10 + (uint64_t)(c - '0');
                if (len > 0xFFFFFFFFull) return corrupt("read length >= 2^32 is not supported");
            }
        }
        row.len = (uint32_t)len;
        const std::string body(l, t3 + 1);
        if (!body.empty()) { // parse_bad_string, stack.rs:217-241: "len,begin,end;..."
            size_t p = 0;
            while (p <= body.size()) {
                size_t e = body.find(';', p);
                if (e == std::string::npos) e = body.size();
                const std::string sub(body, p, e - p);
                size_t c1 = sub.find(','), c2 = c1 == std::string::npos ? c1 : sub.find(',', c1 + 1);
                if (c2 == std::string::npos) return corrupt("position");
                size_t c3 = sub.find(',', c2 + 1);
                auto num = [&](const std::string &s, uint32_t &v) {
                    if (s.empty()) return false;
                    uint64_t x = 0;
                    for (char c : s) {
                        if (c < '0' || c > '9') return false;
                        x = x * 10 + (uint64_t)(c - '0');
                        if (x > 0xFFFFFFFFull) return false;
                    }
                    v = (uint32_t)x;
                    return true;
                };
                uint32_t bgn, end;
                if (!num(sub.substr(c1 + 1, c2 - c1 - 1), bgn) ||
                    !num(sub.substr(c2 + 1, c3 == std::string::npos ? std::string::npos : c3 - c2 - 1), end))
                    return corrupt("position");
                row.reg.push_back(bgn);
                row.reg.push_back(end);
                p = e + 1;
                if (e == body.size()) break;
            }
        }
        auto it = seen.find(row.id);
        if (it != seen.end()) rows[it->second] = std::move(row);
        else {
            seen.emplace(row.id, (uint32_t)rows.size());
            rows.push_back(std::move(row));
        }
    }
    if (in.failed) {
        delete rep;
        return 1; // message set by the decoder
    }
    for (const Row &row : rows) {
        rep->names.insert(rep->names.end(), row.id.begin(), row.id.end());
        rep->name_off.push_back(rep->names.size());
        rep->lengths.push_back(row.len);
        rep->bad_regions.insert(rep->bad_regions.end(), row.reg.begin(), row.reg.end());
        rep->bad_offsets.push_back(rep->bad_regions.size() / 2);
    }
    *out = rep;
    return 0;
}

int yacrd_report_get(const yacrd_report *r, yacrd_badparts_view *v)
{
    if (!r || !v) return yh::fail("null argument");
    v->n_reads = r->lengths.size();
    v->name_off = r->name_off.data();
    v->names = r->names.data();
    v->lengths = r->lengths.data();
    v->bad_offsets = r->bad_offsets.data();
    v->bad_regions = r->bad_regions.data();
    v->read_type = nullptr; // classify with the engine (yacrd_engine_classify) before editing
    return 0;
}

void yacrd_report_free(yacrd_report *r) { delete r; }

int yacrd_edit_file(int op, const char *in_path, const char *out_path, const yacrd_badparts_view *bp)
{
    return yacrd_edit_file_mt(op, in_path, out_path, bp, 0);
}

int yacrd_edit_file_mt(int op, const char *in_path, const char *out_path, const yacrd_badparts_view *bp, int n_threads)
{
    if (op < 0 || op > 3 || !in_path || !out_path || !bp) return yh::fail("bad argument");
    if (bp->n_reads && (!bp->read_type || !bp->bad_offsets || !bp->lengths || !bp->name_off))
        return yh::fail("bad parts table is incomplete (read_type comes from the engine)");
    const FileType ft = file_type(in_path);
    const bool seq = ft == FT_FASTA || ft == FT_FASTQ, ovl = ft == FT_PAF || ft == FT_M4;
    if (ft == FT_NONE || ft == FT_YOVL)
        return yh::fail(std::string("Format detection of file ") + in_path + " failed");
    if (!(seq || (ovl && (op == OP_FILTER || op == OP_EXTRACT))))
        return yh::fail(std::string("Can't run ") + op_name(op) + " on " + type_name(ft) +
                        " file " + in_path);
    unsigned T = n_threads > 0 ? (unsigned)n_threads : std::min(yh::usable_cpus(), 3u); // (they parse side by side and take turns at the output, whose one writer is the bound from two threads up: edit_sequences_parallel)
    if (const char *e = std::getenv("YACRD_EDIT_THREADS"))
        if (*e) T = (unsigned)std::max(1, std::atoi(e));
    T = std::min(T, 64u);
    Reader in;
    if (in.open(in_path)) return 1;
    const BadParts table(bp, T);
    if (seq && T > 1 && in.compression() == yh::COMP_NONE) { // plain sequence files: every usable CPU
        const int rcp = edit_sequences_parallel(op, ft == FT_FASTQ, in_path, out_path, table, T);
        if (rcp >= 0) return rcp;
    }
    Writer out;
    if (out.open(out_path, in.compression())) return 1;
    const int rc = seq ? edit_sequences(op, ft == FT_FASTQ, in, out, table)
                       : edit_overlaps(op, ft == FT_PAF, in, out, table);
    if (rc) return rc;
    return out.close();
}

} // extern "C"
