// host_common.h — shared bits of libyacrd_host (plain C++17).
#pragma once
#include <cstdint>
#include <string>

namespace yh {

std::string &err_slot();
inline int fail(const std::string &msg)
{
    err_slot() = msg;
    return 1;
}

// 64-bit string hash (FNV-1a core with a final avalanche); used for read-id interning.
inline uint64_t hash_bytes(const char *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) {
        h ^= (unsigned char)p[i];
        h *= 0x100000001b3ull;
    }
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 32;
    return h;
}

} // namespace yh
