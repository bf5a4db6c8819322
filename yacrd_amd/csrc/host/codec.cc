// codec.cc — gzip / bzip2 / xz streams for libyacrd_host (see codec.h).
//
// Reference behaviour being matched: niffler::get_reader sniffs the compression from the first
// bytes (src/util.rs:57-70), editors write their output in the input's format at level one
// (src/util.rs:72-87).  A stream that ends early is an error there (flate2 / bzip2 / xz2 return
// UnexpectedEof or a data error) and here.
#include "codec.h"

#include <dlfcn.h>
#include <zlib.h>

#include <cstring>
#include <mutex>

#include "host_common.h"

namespace yh {

Compression sniff_compression(const unsigned char *m, size_t n)
{
    if (n >= 2 && m[0] == 0x1f && m[1] == 0x8b) return COMP_GZIP;
    if (n >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') return COMP_BZIP2;
    if (n >= 6 && m[0] == 0xFD && std::memcmp(m + 1, "7zXZ", 4) == 0 && m[5] == 0) return COMP_XZ;
    return COMP_NONE;
}

namespace {

// ---- libbz2 / liblzma by hand: the runtime libraries are in the image, their headers are not ----
struct bz_stream {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
enum { BZ_RUN = 0, BZ_FINISH = 2, BZ_OK = 0, BZ_RUN_OK = 1, BZ_FINISH_OK = 3, BZ_STREAM_END = 4 };
struct Bz2Api {
    int (*DecompressInit)(bz_stream *, int, int) = nullptr;
    int (*Decompress)(bz_stream *) = nullptr;
    int (*DecompressEnd)(bz_stream *) = nullptr;
    int (*CompressInit)(bz_stream *, int, int, int) = nullptr;
    int (*Compress)(bz_stream *, int) = nullptr;
    int (*CompressEnd)(bz_stream *) = nullptr;
    bool ok = false;
};
const Bz2Api &bz2()
{
    static Bz2Api api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        api.DecompressInit = (decltype(api.DecompressInit))dlsym(h, "BZ2_bzDecompressInit");
        api.Decompress = (decltype(api.Decompress))dlsym(h, "BZ2_bzDecompress");
        api.DecompressEnd = (decltype(api.DecompressEnd))dlsym(h, "BZ2_bzDecompressEnd");
        api.CompressInit = (decltype(api.CompressInit))dlsym(h, "BZ2_bzCompressInit");
        api.Compress = (decltype(api.Compress))dlsym(h, "BZ2_bzCompress");
        api.CompressEnd = (decltype(api.CompressEnd))dlsym(h, "BZ2_bzCompressEnd");
        api.ok = api.DecompressInit && api.Decompress && api.DecompressEnd && api.CompressInit &&
                 api.Compress && api.CompressEnd;
    });
    return api;
}

struct lzma_stream { // liblzma 5.x ABI (stable since 5.0)
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    const void *allocator;
    void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
enum { LZMA_RUN = 0, LZMA_FINISH = 3, LZMA_OK = 0, LZMA_STREAM_END = 1, LZMA_CONCATENATED = 0x08,
       LZMA_CHECK_CRC64 = 4 };
struct LzmaApi {
    int (*stream_decoder)(lzma_stream *, uint64_t, uint32_t) = nullptr;
    int (*easy_encoder)(lzma_stream *, uint32_t, int) = nullptr;
    int (*code)(lzma_stream *, int) = nullptr;
    void (*end)(lzma_stream *) = nullptr;
    bool ok = false;
};
const LzmaApi &lzma()
{
    static LzmaApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"liblzma.so.5", "liblzma.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        api.stream_decoder = (decltype(api.stream_decoder))dlsym(h, "lzma_stream_decoder");
        api.easy_encoder = (decltype(api.easy_encoder))dlsym(h, "lzma_easy_encoder");
        api.code = (decltype(api.code))dlsym(h, "lzma_code");
        api.end = (decltype(api.end))dlsym(h, "lzma_end");
        api.ok = api.stream_decoder && api.easy_encoder && api.code && api.end;
    });
    return api;
}

constexpr size_t kIoBuf = 1u << 20;

} // namespace

// ---- input -------------------------------------------------------------------------------------
struct InStream::Impl {
    FILE *f = nullptr;
    std::vector<unsigned char> in;
    size_t in_pos = 0, in_end = 0;
    bool file_eof = false, done = false;
    bool mid_stream = false; // inside a compressed member (its end has not been seen yet)
    z_stream zs{};
    bz_stream bs{};
    lzma_stream ls{};
    bool z_init = false, b_init = false, l_init = false;

    bool refill()
    {
        if (file_eof) return false;
        in_pos = 0;
        in_end = std::fread(in.data(), 1, in.size(), f);
        if (in_end < in.size()) file_eof = true;
        return in_end > 0;
    }
};

InStream::~InStream()
{
    if (!impl_) return;
    if (impl_->z_init) inflateEnd(&impl_->zs);
    if (impl_->b_init) bz2().DecompressEnd(&impl_->bs);
    if (impl_->l_init) lzma().end(&impl_->ls);
    if (impl_->f) std::fclose(impl_->f);
    delete impl_;
}

int InStream::open(const char *path)
{
    path_ = path;
    impl_ = new Impl();
    impl_->f = std::fopen(path, "rb");
    if (!impl_->f) return fail(std::string("Can't open file ") + path + " to read");
    impl_->in.resize(kIoBuf);
    impl_->refill();
    fmt_ = sniff_compression(impl_->in.data(), impl_->in_end);
    if (fmt_ == COMP_GZIP) {
        if (inflateInit2(&impl_->zs, 15 + 16) != Z_OK) return fail("zlib initialisation failed");
        impl_->z_init = true;
    } else if (fmt_ == COMP_BZIP2) {
        if (!bz2().ok) return fail(std::string(path) + ": bzip2 input needs libbz2.so.1.0, which could not be loaded");
        if (bz2().DecompressInit(&impl_->bs, 0, 0) != BZ_OK) return fail("libbz2 initialisation failed");
        impl_->b_init = true;
    } else if (fmt_ == COMP_XZ) {
        if (!lzma().ok) return fail(std::string(path) + ": xz input needs liblzma.so.5, which could not be loaded");
        if (lzma().stream_decoder(&impl_->ls, UINT64_MAX, LZMA_CONCATENATED) != LZMA_OK)
            return fail("liblzma initialisation failed");
        impl_->l_init = true;
    }
    return 0;
}

long InStream::read(char *dst, size_t cap)
{
    Impl &s = *impl_;
    if (s.done || cap == 0) return 0;
    if (fmt_ == COMP_NONE) {
        if (s.in_pos == s.in_end && !s.refill()) {
            s.done = true;
            if (std::ferror(s.f)) {
                fail("read error in " + path_);
                return -1;
            }
            return 0;
        }
        const size_t n = std::min(cap, s.in_end - s.in_pos);
        std::memcpy(dst, s.in.data() + s.in_pos, n);
        s.in_pos += n;
        return (long)n;
    }
    const char *what = fmt_ == COMP_GZIP ? "gzip" : fmt_ == COMP_BZIP2 ? "bzip2" : "xz";
    size_t produced = 0;
    while (produced == 0) {
        if (s.in_pos == s.in_end && !s.file_eof) s.refill();
        const bool no_input = s.in_pos == s.in_end;
        if (std::ferror(s.f)) {
            fail("read error in " + path_);
            return -1;
        }
        if (fmt_ == COMP_GZIP) {
            if (no_input) { // the file ended: fine between members, an error inside one
                if (s.mid_stream) {
                    fail(std::string("unexpected end of ") + what + " stream in " + path_);
                    return -1;
                }
                s.done = true;
                return 0;
            }
            s.zs.next_in = s.in.data() + s.in_pos;
            s.zs.avail_in = (uInt)(s.in_end - s.in_pos);
            s.zs.next_out = (Bytef *)dst;
            s.zs.avail_out = (uInt)std::min<size_t>(cap, 1u << 30);
            s.mid_stream = true;
            const int rc = inflate(&s.zs, Z_NO_FLUSH);
            s.in_pos = s.in_end - s.zs.avail_in;
            produced = (size_t)((char *)s.zs.next_out - dst);
            if (rc == Z_STREAM_END) { // another member may follow (gzip files concatenate)
                s.mid_stream = false;
                inflateReset(&s.zs);
                // trailing zero padding / garbage after the last member ends the input, like gzip(1)
                if (s.in_pos < s.in_end && s.in.data()[s.in_pos] != 0x1f) {
                    s.done = true;
                    break;
                }
            } else if (rc != Z_OK && !(rc == Z_BUF_ERROR && produced == 0 && s.zs.avail_in == 0)) {
                fail(std::string("corrupt ") + what + " stream in " + path_ +
                     (s.zs.msg ? std::string(": ") + s.zs.msg : std::string()));
                return -1;
            }
        } else if (fmt_ == COMP_BZIP2) {
            if (no_input) {
                if (s.mid_stream) {
                    fail(std::string("unexpected end of ") + what + " stream in " + path_);
                    return -1;
                }
                s.done = true;
                return 0;
            }
            s.bs.next_in = (char *)s.in.data() + s.in_pos;
            s.bs.avail_in = (unsigned)(s.in_end - s.in_pos);
            s.bs.next_out = dst;
            s.bs.avail_out = (unsigned)std::min<size_t>(cap, 1u << 30);
            s.mid_stream = true;
            const int rc = bz2().Decompress(&s.bs);
            s.in_pos = s.in_end - s.bs.avail_in;
            produced = (size_t)(s.bs.next_out - dst);
            if (rc == BZ_STREAM_END) {
                s.mid_stream = false;
                bz2().DecompressEnd(&s.bs);
                std::memset(&s.bs, 0, sizeof s.bs);
                s.b_init = bz2().DecompressInit(&s.bs, 0, 0) == BZ_OK;
                if (!s.b_init) {
                    fail("libbz2 initialisation failed");
                    return -1;
                }
            } else if (rc != BZ_OK) {
                fail(std::string("corrupt ") + what + " stream in " + path_);
                return -1;
            }
        } else {
            s.ls.next_in = s.in.data() + s.in_pos;
            s.ls.avail_in = s.in_end - s.in_pos;
            s.ls.next_out = (uint8_t *)dst;
            s.ls.avail_out = cap;
            const int rc = lzma().code(&s.ls, (no_input && s.file_eof) ? LZMA_FINISH : LZMA_RUN);
            s.in_pos = s.in_end - s.ls.avail_in;
            produced = (size_t)((char *)s.ls.next_out - dst);
            if (rc == LZMA_STREAM_END) {
                s.done = true;
                break;
            }
            if (rc != LZMA_OK) { // LZMA_BUF_ERROR at LZMA_FINISH = the stream stops short
                fail(std::string(no_input ? "unexpected end of " : "corrupt ") + what + " stream in " + path_);
                return -1;
            }
        }
    }
    return (long)produced;
}

// ---- output ------------------------------------------------------------------------------------
struct OutStream::Impl {
    FILE *f = nullptr;
    Compression fmt = COMP_NONE;
    gzFile gz = nullptr;
    bz_stream bs{};
    lzma_stream ls{};
    bool b_init = false, l_init = false, failed = false;
    std::vector<unsigned char> out;

    bool drain(const unsigned char *p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }
    // run the encoder over [p, p + n) (finish = flush everything and end the stream)
    bool encode(const char *p, size_t n, bool finish)
    {
        if (fmt == COMP_BZIP2) {
            bs.next_in = const_cast<char *>(p);
            bs.avail_in = (unsigned)n;
            for (;;) {
                bs.next_out = (char *)out.data();
                bs.avail_out = (unsigned)out.size();
                const int rc = bz2().Compress(&bs, finish ? BZ_FINISH : BZ_RUN);
                if (!drain(out.data(), out.size() - bs.avail_out)) return false;
                if (finish ? rc == BZ_STREAM_END : (rc == BZ_RUN_OK && bs.avail_in == 0 && bs.avail_out != 0)) return true;
                if (rc != BZ_RUN_OK && rc != BZ_FINISH_OK) return false;
            }
        }
        ls.next_in = (const uint8_t *)p;
        ls.avail_in = n;
        for (;;) {
            ls.next_out = out.data();
            ls.avail_out = out.size();
            const int rc = lzma().code(&ls, finish ? LZMA_FINISH : LZMA_RUN);
            if (!drain(out.data(), out.size() - ls.avail_out)) return false;
            if (rc == LZMA_STREAM_END) return true;
            if (rc != LZMA_OK) return false;
            if (!finish && ls.avail_in == 0 && ls.avail_out != 0) return true;
        }
    }
};

OutStream::~OutStream()
{
    if (!impl_) return;
    if (impl_->gz) gzclose(impl_->gz);
    if (impl_->b_init) bz2().CompressEnd(&impl_->bs);
    if (impl_->l_init) lzma().end(&impl_->ls);
    if (impl_->f) std::fclose(impl_->f);
    delete impl_;
}

int OutStream::open(const char *path, Compression fmt)
{
    impl_ = new Impl();
    impl_->fmt = fmt;
    const std::string cant = std::string("Can't open file ") + path + " to write";
    if (fmt == COMP_GZIP) {
        impl_->gz = gzopen(path, "wb1");
        if (!impl_->gz) return fail(cant);
        gzbuffer(impl_->gz, 1 << 20);
        return 0;
    }
    impl_->f = std::fopen(path, "wb");
    if (!impl_->f) return fail(cant);
    if (fmt == COMP_BZIP2) {
        if (!bz2().ok) return fail("bzip2 output needs libbz2.so.1.0, which could not be loaded");
        if (bz2().CompressInit(&impl_->bs, 1, 0, 0) != BZ_OK) return fail("libbz2 initialisation failed");
        impl_->b_init = true;
        impl_->out.resize(kIoBuf);
    } else if (fmt == COMP_XZ) {
        if (!lzma().ok) return fail("xz output needs liblzma.so.5, which could not be loaded");
        if (lzma().easy_encoder(&impl_->ls, 1, LZMA_CHECK_CRC64) != LZMA_OK) return fail("liblzma initialisation failed");
        impl_->l_init = true;
        impl_->out.resize(kIoBuf);
    }
    return 0;
}

bool OutStream::write(const char *p, size_t n)
{
    Impl &s = *impl_;
    if (s.failed) return false;
    if (n == 0) return true;
    if (s.fmt == COMP_GZIP) s.failed = gzwrite(s.gz, p, (unsigned)n) != (int)n;
    else if (s.fmt == COMP_NONE) s.failed = std::fwrite(p, 1, n, s.f) != n;
    else s.failed = !s.encode(p, n, false);
    return !s.failed;
}

int OutStream::close()
{
    if (!impl_) return 0;
    Impl &s = *impl_;
    if (!s.failed && (s.fmt == COMP_BZIP2 || s.fmt == COMP_XZ)) s.failed = !s.encode(nullptr, 0, true);
    if (s.gz) {
        s.failed |= gzclose(s.gz) != Z_OK;
        s.gz = nullptr;
    }
    if (s.f) {
        s.failed |= std::fclose(s.f) != 0;
        s.f = nullptr;
    }
    return s.failed ? fail("Error during writing of the output file") : 0;
}

} // namespace yh

// ---- a whole compressed file into memory (yacrd_text_*: the text the device parser takes) ---------------------------
// The reference reads compressed overlap files through niffler like any other (src/util.rs:57-70); the device
// parser wants the TEXT in one piece.  A gzip / bzip2 / xz stream is one thread's work whatever reads it (a deflate
// stream cannot be entered in the middle): ~0.3-0.5 GB/s of text, which then is the ingest's wall clock whichever
// parser follows.  The exception is BGZF (bgzip, htslib: the usual way bioinformatics files are gzipped when they are
// meant to be read fast): members of at most 64 KiB that say their own size in an extra field, independent of each
// other — found by walking the headers, inflated on every usable CPU into their final places (every member ends with its
// uncompressed size).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <thread>

#include "../../../include/yacrd_host.h"

namespace {

struct Mapping { // a grow-only anonymous mapping: the text lands in one piece without a realloc copy
    char *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t want)
    {
        if (want <= cap) return true;
        size_t ncap = std::max<size_t>(want, cap ? cap * 2 : (size_t)64 << 20);
        ncap = (ncap + 4095) & ~(size_t)4095;
        void *q = p ? mremap(p, cap, ncap, MREMAP_MAYMOVE)
                    : mmap(nullptr, ncap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (q == MAP_FAILED) return false;
        p = (char *)q;
        cap = ncap;
        return true;
    }
    void release()
    {
        if (p) munmap(p, cap);
        p = nullptr;
        cap = 0;
    }
};

struct BgzfMember {
    uint64_t at, csize, out, isize; // the member in the file; its place and size in the text
};
// BGZF (SAM spec 4.1): gzip members with FLG.FEXTRA and a "BC" subfield holding BSIZE = member size - 1.
// false = not BGZF from its first member to its last (the caller streams the file instead).
bool bgzf_members(const unsigned char *f, uint64_t n, std::vector<BgzfMember> &out, uint64_t &total)
{
    uint64_t at = 0;
    total = 0;
    while (at < n) {
        if (n - at < 18 + 8 || f[at] != 0x1f || f[at + 1] != 0x8b || f[at + 2] != 8 || !(f[at + 3] & 4)) return false;
        const uint32_t xlen = f[at + 10] | (f[at + 11] << 8);
        if (n - at < 12 + (uint64_t)xlen) return false;
        uint64_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const unsigned char *sf = f + at + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (uint64_t)(sf[4] | (sf[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + (uint64_t)xlen + 8 || at + bsize > n) return false;
        const unsigned char *tail = f + at + bsize - 4;
        const uint64_t isize = (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) | ((uint64_t)tail[3] << 24);
        // (name / comment / header-crc fields do not occur in BGZF: the deflate data starts behind the extra field)
        if (f[at + 3] & ~4u) return false;
        out.push_back(BgzfMember{at, bsize, total, isize});
        total += isize;
        at += bsize;
    }
    return !out.empty();
}

} // namespace

extern "C" {

int yacrd_text_from_file(const char *path, int n_threads, yacrd_text *out)
{
    if (!path || !out) return yh::fail("null argument");
    std::memset(out, 0, sizeof(*out));
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return yh::fail(std::string("Can't open file ") + path + " to read");
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return 2; // not a regular file: the caller streams it
    }
    unsigned char magic[8] = {0};
    const ssize_t got = ::pread(fd, magic, sizeof magic, 0);
    const yh::Compression fmt = yh::sniff_compression(magic, got > 0 ? (size_t)got : 0);
    if (fmt == yh::COMP_NONE) {
        ::close(fd);
        return 2; // plain text: read it where it lies
    }
    out->compression = (int)fmt;
    Mapping m;
    const uint64_t csize = (uint64_t)st.st_size;
    unsigned T = n_threads > 0 ? (unsigned)n_threads : yh::usable_cpus();
    if (fmt == yh::COMP_GZIP && csize >= 28) {
        const unsigned char *f = (const unsigned char *)mmap(nullptr, csize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (f != MAP_FAILED) {
            std::vector<BgzfMember> mem;
            uint64_t total = 0;
            if (bgzf_members(f, csize, mem, total) && m.reserve((size_t)total + 64)) {
                std::atomic<size_t> next(0);
                std::atomic<int> bad(0);
                T = (unsigned)std::max<size_t>(1, std::min<size_t>(T, mem.size() / 16 + 1));
                auto work = [&]() {
                    z_stream zs{};
                    if (inflateInit2(&zs, -15) != Z_OK) {
                        bad = 1;
                        return;
                    }
                    for (;;) {
                        const size_t i0 = next.fetch_add(64);
                        if (i0 >= mem.size() || bad.load()) break;
                        for (size_t i = i0; i < std::min(i0 + 64, mem.size()); i++) {
                            const BgzfMember &b = mem[i];
                            const uint32_t xlen = f[b.at + 10] | (f[b.at + 11] << 8);
                            inflateReset(&zs);
                            zs.next_in = const_cast<Bytef *>(f + b.at + 12 + xlen);
                            zs.avail_in = (uInt)(b.csize - 12 - xlen - 8);
                            zs.next_out = (Bytef *)m.p + b.out;
                            zs.avail_out = (uInt)b.isize;
                            const int rc = inflate(&zs, Z_FINISH);
                            if (rc != Z_STREAM_END || zs.avail_out != 0 || zs.avail_in != 0) {
                                bad = 1;
                                break;
                            }
                            const unsigned char *tail = f + b.at + b.csize - 8;
                            const uint32_t want = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
                            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)m.p + b.out, (uInt)b.isize) != want) {
                                bad = 1;
                                break;
                            }
                        }
                    }
                    inflateEnd(&zs);
                };
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T; t++) th.emplace_back(work);
                work();
                for (auto &x : th) x.join();
                munmap((void *)f, csize);
                ::close(fd);
                if (bad.load()) {
                    m.release();
                    return yh::fail(std::string("corrupt gzip stream in ") + path);
                }
                out->data = m.p;
                out->n = total;
                out->cap = m.cap;
                out->members = mem.size();
                out->threads = T;
                return 0;
            }
            munmap((void *)f, csize);
            m.release();
        }
    }
    ::close(fd);
    // one stream: one thread, straight into the mapping
    yh::InStream in;
    if (in.open(path)) return 1;
    uint64_t n = 0;
    for (;;) {
        if (!m.reserve((size_t)n + ((size_t)8 << 20) + 64)) {
            m.release();
            return yh::fail("out of memory while decompressing " + std::string(path));
        }
        const long k = in.read(m.p + n, (size_t)8 << 20);
        if (k < 0) {
            m.release();
            return 1; // (message set by the decoder)
        }
        if (k == 0) break;
        n += (uint64_t)k;
    }
    out->data = m.p;
    out->n = n;
    out->cap = m.cap;
    out->members = 1;
    out->threads = 1;
    return 0;
}

void yacrd_text_free(yacrd_text *t)
{
    if (!t) return;
    if (t->data) munmap(t->data, (size_t)t->cap);
    std::memset(t, 0, sizeof(*t));
}

} // extern "C"
