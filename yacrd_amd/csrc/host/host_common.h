// host_common.h — shared bits of libyacrd_host (plain C++17).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

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

// CPUs this process may actually use: hardware threads, capped by the cgroup CPU quota (threads beyond
// the quota only get throttled).
inline unsigned usable_cpus()
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

} // namespace yh
