// synth.cc — deterministic synthetic overlap workloads.  Bench/test tooling on
// the host side; nothing here is part of the reference.  PRNG: splitmix64-seeded xoshiro256**.
#include "../../../include/yacrd_host.h"
#include "host_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed)
    {
        for (auto &v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); } // [0,1)
    uint64_t below(uint64_t n) { return n ? (uint64_t)(uniform() * (double)n) : 0; }   // [0,n)
    double normal()
    {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Line {
    uint32_t a, b;
    uint32_t sa, ea, sb, eb;
};

// Every read and every line has its own PRNG stream (seeded from the config's seed and the read /
// line number), so any range of lines can be generated on any thread: the output is the same for
// every thread count.  (Round 1 drew everything from ONE stream: 385 s for configs[4].)
inline Rng stream_rng(uint64_t seed, uint64_t kind, uint64_t id)
{
    return Rng(seed ^ (kind * 0xA0761D6478BD642Full) ^ (id * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
}

unsigned synth_threads()
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
    return std::min(n, 64u);
}

template <class F>
void parallel_ranges(uint64_t n, unsigned T, F fn) // fn(t, begin, end) over T contiguous ranges
{
    T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, n ? n : 1));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
    fn(0u, (uint64_t)0, n / T);
    for (auto &x : th) x.join();
}

struct Gen {
    yacrd_synth_cfg cfg;
    std::vector<uint32_t> len;      // read lengths
    std::vector<uint32_t> junction; // 0 = none (2 % of reads: chimera junction)
    std::vector<uint32_t> win0, win1; // win1 == 0: none (3 % of reads: only a 20 % window is covered)
    std::vector<double> zipf_cdf;   // skewed profile
    uint64_t n_round_robin = 0;

    explicit Gen(const yacrd_synth_cfg &c) : cfg(c)
    {
        const uint64_t R = c.n_reads;
        len.resize(R);
        junction.assign(R, 0);
        win0.assign(R, 0);
        win1.assign(R, 0);
        const uint32_t pct = (c.flags >> 16) & 0xFFu; // YACRD_SYNTH_F_CHIMERA_PCT
        const double chimera = pct ? std::min(pct, 97u) / 100.0 : 0.02;
        parallel_ranges(R, synth_threads(), [&](unsigned, uint64_t r0, uint64_t r1) {
            for (uint64_t r = r0; r < r1; r++) {
                Rng rng = stream_rng(c.seed, 1, r);
                double L;
                if (c.profile == YACRD_SYNTH_ONT) {
                    L = std::exp(std::log(6000.0) + 0.9 * rng.normal());
                    L = std::min(std::max(L, 500.0), 150000.0);
                } else if (c.profile == YACRD_SYNTH_SEQUEL) {
                    L = std::exp(std::log(9000.0) + 0.5 * rng.normal());
                    L = std::min(std::max(L, 1000.0), 60000.0);
                } else {
                    L = 200000.0 + rng.uniform() * 800000.0;
                }
                len[r] = (uint32_t)L;
                const double u = rng.uniform();
                if (u < chimera) {
                    junction[r] = (uint32_t)((0.2 + 0.6 * rng.uniform()) * L);
                    if (junction[r] == 0) junction[r] = 1;
                } else if (u < chimera + 0.03) {
                    const uint32_t w = std::max<uint32_t>(len[r] / 5, 2);
                    win0[r] = (uint32_t)rng.below(len[r] - w + 1);
                    win1[r] = win0[r] + w;
                }
            }
        });
        if (c.profile == YACRD_SYNTH_SKEWED) {
            n_round_robin = c.n_overlaps / 6 * 5;
            zipf_cdf.resize(R);
            double acc = 0;
            for (uint64_t k = 0; k < R; k++) {
                acc += std::pow((double)(k + 1), -1.1);
                zipf_cdf[k] = acc;
            }
            for (auto &v : zipf_cdf) v /= acc;
        }
    }

    void draw_interval(Rng &rng, uint32_t r, uint32_t &s, uint32_t &e) const
    {
        const uint32_t L = len[r];
        int64_t lo = 0, hi = L;
        if (win1[r]) {
            lo = win0[r];
            hi = win1[r];
        }
        const int64_t span = hi - lo;
        int64_t st, en;
        if (rng.uniform() < 0.6) { // dovetail anchored near one end
            const int64_t ell = 500 + (int64_t)rng.below((uint64_t)std::max<int64_t>(1, (int64_t)(0.8 * span) - 500));
            // N(0, sigma) around the read's end: clamped onto it below (SURVEY.md §8d as specified: half of
            // the dovetail ends then sit on exactly 0 / len), or — YACRD_SYNTH_F_JITTER — reflected into
            // the read, so that the ends are spread over a few dozen positions like the chain ends of a
            // real overlapper and no exact position holds a pile
            const unsigned sig = ((cfg.flags >> 8) & 0xFFu) * ((cfg.flags & YACRD_SYNTH_F_SIGMA_X4) ? 4u : 1u);
            int64_t jit = (int64_t)std::llround((sig ? (double)sig : 30.0) * rng.normal());
            const bool start_side = (rng.next() & 1) != 0;
            if (cfg.flags & YACRD_SYNTH_F_JITTER) jit = start_side ? std::llabs(jit) : -std::llabs(jit);
            if (start_side) {
                st = lo + jit;
                en = st + ell;
            } else {
                en = hi + jit;
                st = en - ell;
            }
        } else { // internal
            const int64_t ell = 500 + (int64_t)rng.below((uint64_t)std::max<int64_t>(1, span / 2 - 500));
            st = lo + (int64_t)rng.below((uint64_t)std::max<int64_t>(1, span - ell));
            en = st + ell;
        }
        st = std::min(std::max(st, lo), hi - 1);
        en = std::min(std::max(en, st + 1), hi);
        const double u = (cfg.flags & 1u) ? 1.0 : rng.uniform();
        if (u < 2e-3 && !win1[r]) {
            // abutting intervals without any state: snap both ends to an eighth-of-the-read grid;
            // two snapped intervals of a read often share a grid point (end of one == start of
            // the other), which is what the reference's zero-length-gap quirk needs
            const int64_t g = std::max<int64_t>(1, L / 8);
            st = (st + g / 2) / g * g;
            en = std::max(st + g, (en + g / 2) / g * g);
            st = std::min(st, (int64_t)L - 1);
            en = std::min<int64_t>(std::max(en, st + 1), L);
        }
        if (junction[r]) { // chimera: nothing crosses the junction
            const int64_t j = junction[r];
            if (st < j && en > j) {
                const int64_t cut = 10 + (int64_t)rng.below(91);
                if ((st + en) / 2 < j) en = std::max(st + 1, j - cut);
                else st = std::min(en - 1, j + cut);
            }
        }
        if (u >= 2e-3 && u < 2e-3 + 1e-5) en = st; // degenerate: start == end
        s = (uint32_t)st;
        e = (uint32_t)en;
    }

    void line(uint64_t i, Line &ln) const
    {
        const uint64_t R = cfg.n_reads;
        Rng rng = stream_rng(cfg.seed, 2, i);
        uint32_t a, b;
        if (cfg.profile == YACRD_SYNTH_SKEWED && i < n_round_robin) {
            a = (uint32_t)((2 * i) % R);
            b = (uint32_t)((2 * i + 1) % R);
        } else if (cfg.profile == YACRD_SYNTH_SKEWED) {
            const double u = rng.uniform();
            a = (uint32_t)(std::lower_bound(zipf_cdf.begin(), zipf_cdf.end(), u) - zipf_cdf.begin());
            if (a >= R) a = (uint32_t)R - 1;
            b = (uint32_t)rng.below(R - 1);
            if (b >= a) b++;
        } else {
            a = (uint32_t)rng.below(R);
            b = (uint32_t)rng.below(R - 1);
            if (b >= a) b++;
        }
        ln.a = a;
        ln.b = b;
        draw_interval(rng, a, ln.sa, ln.ea);
        draw_interval(rng, b, ln.sb, ln.eb);
    }
};

// decimal digits of v at p, returns the end
inline char *put_uint(char *p, uint64_t v)
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
inline char *put_id(char *p, char prefix, uint64_t v) // r%09u
{
    *p++ = prefix;
    for (int k = 8; k >= 0; k--) {
        p[k] = (char)('0' + v % 10);
        v /= 10;
    }
    return p + 9;
}

int check_cfg(const yacrd_synth_cfg *cfg)
{
    if (!cfg) return yh::fail("cfg is null");
    if (cfg->n_reads < 2 || cfg->n_reads >= 0xFFFFFFFFull) return yh::fail("n_reads out of range");
    if (cfg->profile > YACRD_SYNTH_SKEWED) return yh::fail("unknown profile");
    return 0;
}

} // namespace

extern "C" {

int yacrd_synth_csr(const yacrd_synth_cfg *cfg, uint64_t *offsets, uint32_t *intervals,
                    uint32_t *lengths)
{
    if (check_cfg(cfg)) return 1;
    if (!offsets || !intervals || !lengths) return yh::fail("null output");
    const uint64_t R = cfg->n_reads, N = cfg->n_overlaps;
    const Gen g(*cfg);
    // T contiguous line ranges; private per-range counts when they fit (no atomics, and the fill
    // below lands every interval where a sequential pass in line order would put it)
    unsigned T = synth_threads();
    while (T > 1 && (uint64_t)T * R * sizeof(uint32_t) > ((uint64_t)2 << 30)) T /= 2;
    T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, N ? N : 1));
    std::vector<std::vector<uint32_t>> cnt(T);
    parallel_ranges(N, T, [&](unsigned t, uint64_t i0, uint64_t i1) {
        cnt[t].assign(R, 0u);
        Line ln;
        for (uint64_t i = i0; i < i1; i++) {
            g.line(i, ln);
            cnt[t][ln.a]++;
            cnt[t][ln.b]++;
        }
    });
    uint64_t acc = 0;
    for (uint64_t r = 0; r < R; r++) { // offsets, and every range's first slot inside every read
        offsets[r] = acc;
        uint64_t in_read = 0;
        for (unsigned t = 0; t < T; t++) {
            const uint32_t n = cnt[t][r];
            cnt[t][r] = (uint32_t)in_read;
            in_read += n;
        }
        acc += in_read;
        lengths[r] = g.len[r];
    }
    offsets[R] = acc;
    parallel_ranges(N, T, [&](unsigned t, uint64_t i0, uint64_t i1) {
        Line ln;
        uint32_t *k = cnt[t].data();
        for (uint64_t i = i0; i < i1; i++) { // line order inside every read, like an ingest of the PAF
            g.line(i, ln);
            uint64_t p = offsets[ln.a] + k[ln.a]++;
            intervals[2 * p] = ln.sa;
            intervals[2 * p + 1] = ln.ea;
            p = offsets[ln.b] + k[ln.b]++;
            intervals[2 * p] = ln.sb;
            intervals[2 * p + 1] = ln.eb;
        }
    });
    return 0;
}

int yacrd_synth_paf(const yacrd_synth_cfg *cfg, const char *path)
{
    if (check_cfg(cfg)) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return yh::fail(std::string("cannot open ") + path);
    const Gen g(*cfg);
    const unsigned T = synth_threads();
    const uint64_t N = cfg->n_overlaps, kRound = (uint64_t)T * 65536; // lines formatted per round
    std::vector<std::vector<char>> out(T);
    bool ok = true;
    for (uint64_t base = 0; base < N && ok; base += kRound) {
        const uint64_t n = std::min(kRound, N - base);
        parallel_ranges(n, T, [&](unsigned t, uint64_t i0, uint64_t i1) {
            std::vector<char> &o = out[t];
            o.resize((size_t)(i1 - i0) * 128 + 128);
            char *p = o.data();
            Line ln;
            for (uint64_t i = base + i0; i < base + i1; i++) {
                g.line(i, ln);
                const uint32_t la = ln.ea - ln.sa, lb = ln.eb - ln.sb;
                p = put_id(p, 'r', ln.a);
                *p++ = '\t';
                p = put_uint(p, g.len[ln.a]);
                *p++ = '\t';
                p = put_uint(p, ln.sa);
                *p++ = '\t';
                p = put_uint(p, ln.ea);
                *p++ = '\t';
                *p++ = (i & 1) ? '-' : '+';
                *p++ = '\t';
                p = put_id(p, 'r', ln.b);
                *p++ = '\t';
                p = put_uint(p, g.len[ln.b]);
                *p++ = '\t';
                p = put_uint(p, ln.sb);
                *p++ = '\t';
                p = put_uint(p, ln.eb);
                *p++ = '\t';
                p = put_uint(p, std::min(la, lb));
                *p++ = '\t';
                p = put_uint(p, std::max(la, lb));
                std::memcpy(p, "\t255\ttp:A:S\n", 12);
                p += 12;
            }
            o.resize((size_t)(p - o.data()));
        });
        const unsigned used = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, n));
        for (unsigned t = 0; t < used && ok; t++)
            ok = out[t].empty() || std::fwrite(out[t].data(), 1, out[t].size(), f) == out[t].size();
    }
    if (std::fclose(f) != 0 || !ok) return yh::fail("write error");
    return 0;
}

int yacrd_synth_fastq(const yacrd_synth_cfg *cfg, uint64_t extra_reads, const char *path)
{
    if (check_cfg(cfg)) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return yh::fail(std::string("cannot open ") + path);
    const Gen g(*cfg); // same lengths as the overlaps
    const uint64_t total = cfg->n_reads + extra_reads;
    // which records are the extras (reads no overlap mentions), spread over the file: decided in one cheap pass, so
    // that the records themselves — every one with a generator of its own, seeded by its position — can be
    // formatted on every thread (configs[4]'s FASTQ is ~100 GB: minutes from one thread and one stream)
    std::vector<uint64_t> id(total); // read index, or extra index | 1 << 63
    {
        uint64_t next_extra = extra_reads ? total / extra_reads / 2 : total, emitted_extra = 0, r = 0;
        for (uint64_t i = 0; i < total; i++) {
            const bool extra = emitted_extra < extra_reads && (i == next_extra || r >= cfg->n_reads);
            if (extra) {
                id[i] = emitted_extra++ | (1ull << 63);
                next_extra += total / extra_reads;
            } else {
                id[i] = r++;
            }
        }
    }
    const unsigned T = synth_threads();
    const uint64_t kRound = (uint64_t)T * 256; // records formatted per round
    std::vector<std::vector<char>> out(T);
    bool ok = true;
    for (uint64_t base = 0; base < total && ok; base += kRound) {
        const uint64_t n = std::min(kRound, total - base);
        parallel_ranges(n, T, [&](unsigned t, uint64_t i0, uint64_t i1) {
            std::vector<char> &o = out[t];
            o.clear();
            char head[96];
            for (uint64_t i = base + i0; i < base + i1; i++) {
                Rng rng = stream_rng(cfg->seed ^ 0x5eedf00dull, 7, i);
                const bool extra = (id[i] >> 63) != 0;
                const uint64_t k = id[i] & ~(1ull << 63);
                const uint32_t len = extra ? 200 + (uint32_t)rng.below(3000) : g.len[k];
                const int hn = extra ? std::snprintf(head, sizeof(head), "@x%09llu no overlap len=%u\n", (unsigned long long)k, len)
                                     : std::snprintf(head, sizeof(head), "@r%09llu synthetic len=%u\n", (unsigned long long)k, len);
                const size_t at = o.size();
                o.resize(at + (size_t)hn + 2 * (size_t)len + 4);
                char *p = o.data() + at;
                std::memcpy(p, head, (size_t)hn);
                p += hn;
                for (uint32_t q = 0; q < len; q += 32) { // 2 bits per base
                    uint64_t w = rng.next();
                    const uint32_t m = std::min<uint32_t>(32, len - q);
                    for (uint32_t jj = 0; jj < m; jj++, w >>= 2) p[q + jj] = "ACGT"[w & 3];
                }
                p += len;
                std::memcpy(p, "\n+\n", 3);
                p += 3;
                std::memset(p, '?', len);
                p[len] = '\n';
            }
        });
        const unsigned used = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, n));
        for (unsigned t = 0; t < used && ok; t++)
            ok = out[t].empty() || std::fwrite(out[t].data(), 1, out[t].size(), f) == out[t].size();
    }
    if (std::fclose(f) != 0 || !ok) return yh::fail("write error");
    return 0;
}

} // extern "C"
