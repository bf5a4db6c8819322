// codec.h — compressed input / output streams of libyacrd_host, sniffed from magic bytes like
// niffler 2.x does for the reference (src/util.rs:57-87): gzip, bzip2, xz, or none.
// zlib is linked; libbz2.so.1.0 and liblzma.so.5 are dlopen'ed on first use with hand-declared
// prototypes (the image ships the runtime libraries but not their headers).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace yh {

enum Compression { COMP_NONE = 0, COMP_GZIP = 1, COMP_BZIP2 = 2, COMP_XZ = 3 };

// magic-byte detection (gzip 1f 8b, bzip2 "BZh", xz fd "7zXZ" 00)
Compression sniff_compression(const unsigned char *magic, size_t n);

// Sequential decoder over a file.  read() returns the number of bytes produced (0 = clean end of
// input), or -1 with the thread's error slot set: a truncated or corrupt stream is an error, never
// a silent short read (the reference's niffler/flate2/bzip2/xz2 readers fail the same way).
class InStream {
public:
    InStream() = default;
    ~InStream();
    InStream(const InStream &) = delete;
    InStream &operator=(const InStream &) = delete;
    int open(const char *path); // 0 / 1 like yh::fail
    Compression format() const { return fmt_; }
    long read(char *dst, size_t cap);

private:
    struct Impl;
    Impl *impl_ = nullptr;
    Compression fmt_ = COMP_NONE;
    std::string path_;
};

// Sequential encoder; level 1 like the reference (niffler::compression::Level::One).
class OutStream {
public:
    OutStream() = default;
    ~OutStream();
    OutStream(const OutStream &) = delete;
    OutStream &operator=(const OutStream &) = delete;
    int open(const char *path, Compression fmt);
    bool write(const char *p, size_t n); // false on failure
    int close();                         // 0 / 1 like yh::fail

private:
    struct Impl;
    Impl *impl_ = nullptr;
};

} // namespace yh
