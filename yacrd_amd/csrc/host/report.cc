// report.cc — the .yacrd report, byte for byte the reference's format.
// Reference: src/editor/mod.rs:61-83 (report), :102-107 (bad_region_format), :51-58 (as_str);
// src/main.rs:80-84 writes one line per read.  The reference iterates an FxHashSet (order
// unspecified, its tests compare as a set, tests/run.rs:33-62); we write in CSR order.
#include "../../../include/yacrd_host.h"
#include "host_common.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline char *put_u64(char *p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

const char *kTypeName[3] = {"NotBad", "Chimeric", "NotCovered"};

} // namespace

extern "C" int yacrd_report_write(const char *path, const yacrd_csr_view *reads,
                                  const uint64_t *bad_offsets, const uint32_t *bad_regions,
                                  const uint8_t *read_type)
{
    if (!path || !reads || (reads->n_reads && (!bad_offsets || !read_type)))
        return yh::fail("null argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return yh::fail(std::string("Can't open file ") + path + " to write");
    std::vector<char> buf;
    buf.reserve(1 << 22);
    for (uint64_t r = 0; r < reads->n_reads; r++) {
        const uint64_t a = bad_offsets[r], b = bad_offsets[r + 1];
        const size_t nlen = (size_t)(reads->name_off[r + 1] - reads->name_off[r]);
        const size_t need = 16 + nlen + 24 + (size_t)(b - a) * 36 + 4;
        const size_t used = buf.size();
        buf.resize(used + need);
        char *p = buf.data() + used;
        if (read_type[r] > 2) {
            std::fclose(f);
            return yh::fail("invalid read type");
        }
        const char *tn = kTypeName[read_type[r]];
        const size_t tl = std::strlen(tn);
        std::memcpy(p, tn, tl);
        p += tl;
        *p++ = '\t';
        std::memcpy(p, reads->names + reads->name_off[r], nlen);
        p += nlen;
        *p++ = '\t';
        p = put_u64(p, reads->lengths[r]);
        *p++ = '\t';
        for (uint64_t k = a; k < b; k++) {
            const uint32_t beg = bad_regions[2 * k], end = bad_regions[2 * k + 1];
            if (k != a) *p++ = ';';
            p = put_u64(p, (uint32_t)(end - beg)); // u32 wrapping, like `b.1 - b.0` in release
            *p++ = ',';
            p = put_u64(p, beg);
            *p++ = ',';
            p = put_u64(p, end);
        }
        *p++ = '\n';
        buf.resize((size_t)(p - buf.data()));
        if (buf.size() > (1u << 22) - 65536) {
            if (std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) {
                std::fclose(f);
                return yh::fail("Error while writing the yacrd report");
            }
            buf.clear();
        }
    }
    if (!buf.empty() && std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) {
        std::fclose(f);
        return yh::fail("Error while writing the yacrd report");
    }
    if (std::fclose(f) != 0) return yh::fail("Error while writing the yacrd report");
    return 0;
}
