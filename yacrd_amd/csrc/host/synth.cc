// synth.cc — deterministic synthetic overlap workloads.  Bench/test tooling on
// the host side; nothing here is part of the reference.  PRNG: splitmix64-seeded xoshiro256**.
#include "../../../include/yacrd_host.h"
#include "host_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
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

struct Gen {
    yacrd_synth_cfg cfg;
    Rng rng;
    std::vector<uint32_t> len;      // read lengths
    std::vector<uint32_t> junction; // 0 = none (2 % of reads: chimera junction)
    std::vector<uint32_t> win0, win1; // win1 == 0: none (3 % of reads: only a 20 % window is covered)
    std::vector<uint32_t> last_end; // state for abutting injection
    std::vector<double> zipf_cdf;   // skewed profile
    uint64_t line_no = 0, n_round_robin = 0;

    explicit Gen(const yacrd_synth_cfg &c) : cfg(c), rng(c.seed)
    {
        const uint64_t R = c.n_reads;
        len.resize(R);
        junction.assign(R, 0);
        win0.assign(R, 0);
        win1.assign(R, 0);
        last_end.assign(R, 0);
        for (uint64_t r = 0; r < R; r++) {
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
            if (u < 0.02) {
                junction[r] = (uint32_t)((0.2 + 0.6 * rng.uniform()) * L);
                if (junction[r] == 0) junction[r] = 1;
            } else if (u < 0.05) {
                const uint32_t w = std::max<uint32_t>(len[r] / 5, 2);
                win0[r] = (uint32_t)rng.below(len[r] - w + 1);
                win1[r] = win0[r] + w;
            }
        }
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

    void draw_interval(uint32_t r, uint32_t &s, uint32_t &e)
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
            const int64_t jit = (int64_t)std::llround(30.0 * rng.normal());
            if (rng.next() & 1) {
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
        if (junction[r]) { // chimera: nothing crosses the junction
            const int64_t j = junction[r];
            if (st < j && en > j) {
                const int64_t cut = 10 + (int64_t)rng.below(91);
                if ((st + en) / 2 < j) en = std::max(st + 1, j - cut);
                else st = std::min(en - 1, j + cut);
            }
        }
        if (!(cfg.flags & 1u)) {
            const double u = rng.uniform();
            if (u < 1e-4) { // abutting: start where the read's previous interval ended
                const int64_t p = last_end[r];
                if (p > 0 && p < (int64_t)L) {
                    const int64_t ell = en - st;
                    st = p;
                    en = std::min<int64_t>(p + ell, L);
                }
            } else if (u < 1e-4 + 1e-5) { // degenerate: start == end
                en = st;
            }
        }
        s = (uint32_t)st;
        e = (uint32_t)en;
        last_end[r] = e;
    }

    void next(Line &ln)
    {
        const uint64_t R = cfg.n_reads;
        uint32_t a, b;
        if (cfg.profile == YACRD_SYNTH_SKEWED && line_no < n_round_robin) {
            a = (uint32_t)((2 * line_no) % R);
            b = (uint32_t)((2 * line_no + 1) % R);
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
        draw_interval(a, ln.sa, ln.ea);
        draw_interval(b, ln.sb, ln.eb);
        line_no++;
    }
};

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
    const uint64_t R = cfg->n_reads;
    Line ln;
    { // pass 1: intervals per read
        Gen g(*cfg);
        std::memset(offsets, 0, (R + 1) * sizeof(uint64_t));
        for (uint64_t i = 0; i < cfg->n_overlaps; i++) {
            g.next(ln);
            offsets[ln.a + 1]++;
            offsets[ln.b + 1]++;
        }
        for (uint64_t r = 0; r < R; r++) {
            offsets[r + 1] += offsets[r];
            lengths[r] = g.len[r];
        }
    }
    { // pass 2: same stream, fill in line order (the order an ingest of the PAF would produce)
        Gen g(*cfg);
        std::vector<uint64_t> cur(offsets, offsets + R);
        for (uint64_t i = 0; i < cfg->n_overlaps; i++) {
            g.next(ln);
            uint64_t p = cur[ln.a]++;
            intervals[2 * p] = ln.sa;
            intervals[2 * p + 1] = ln.ea;
            p = cur[ln.b]++;
            intervals[2 * p] = ln.sb;
            intervals[2 * p + 1] = ln.eb;
        }
    }
    return 0;
}

int yacrd_synth_paf(const yacrd_synth_cfg *cfg, const char *path)
{
    if (check_cfg(cfg)) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return yh::fail(std::string("cannot open ") + path);
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    Gen g(*cfg);
    Line ln;
    for (uint64_t i = 0; i < cfg->n_overlaps; i++) {
        g.next(ln);
        const uint32_t la = ln.ea - ln.sa, lb = ln.eb - ln.sb;
        std::fprintf(f, "r%09u\t%u\t%u\t%u\t%c\tr%09u\t%u\t%u\t%u\t%u\t%u\t255\ttp:A:S\n", ln.a,
                     g.len[ln.a], ln.sa, ln.ea, (i & 1) ? '-' : '+', ln.b, g.len[ln.b], ln.sb, ln.eb,
                     std::min(la, lb), std::max(la, lb));
    }
    if (std::fclose(f) != 0) return yh::fail("write error");
    return 0;
}

int yacrd_synth_fastq(const yacrd_synth_cfg *cfg, uint64_t extra_reads, const char *path)
{
    if (check_cfg(cfg)) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return yh::fail(std::string("cannot open ") + path);
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    Gen g(*cfg); // same lengths as the overlaps
    Rng rng(cfg->seed ^ 0x5eedf00dull);
    std::string seq, qual;
    const uint64_t total = cfg->n_reads + extra_reads;
    uint64_t next_extra = extra_reads ? total / extra_reads / 2 : total; // spread the extras
    uint64_t emitted_extra = 0, r = 0;
    for (uint64_t i = 0; i < total; i++) {
        const bool extra = emitted_extra < extra_reads && (i == next_extra || r >= cfg->n_reads);
        uint32_t len;
        if (extra) {
            len = 200 + (uint32_t)rng.below(3000);
            next_extra += total / extra_reads;
        } else {
            len = g.len[r];
        }
        seq.resize(len);
        for (uint32_t k = 0; k < len; k += 32) { // 2 bits per base
            uint64_t w = rng.next();
            const uint32_t m = std::min<uint32_t>(32, len - k);
            for (uint32_t j = 0; j < m; j++, w >>= 2) seq[k + j] = "ACGT"[w & 3];
        }
        qual.assign(len, '?');
        if (extra) std::fprintf(f, "@x%09llu no overlap len=%u\n", (unsigned long long)emitted_extra++, len);
        else std::fprintf(f, "@r%09llu synthetic len=%u\n", (unsigned long long)r++, len);
        std::fwrite(seq.data(), 1, len, f);
        std::fwrite("\n+\n", 1, 3, f);
        std::fwrite(qual.data(), 1, len, f);
        std::fputc('\n', f);
    }
    if (std::fclose(f) != 0) return yh::fail("write error");
    return 0;
}

} // extern "C"
