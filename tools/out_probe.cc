// tools/out_probe.cc — how fast can N threads put G bytes into ONE output file on this box, by the way the bytes get there?
//   pwrite    : pwrite of 16 MB chunks at their offsets (what the chunk-parallel editors do by default)
//   map       : ftruncate + one MAP_SHARED mapping, memcpy (a write fault per page)
//   populate  : the same, MADV_POPULATE_WRITE over the chunk before the memcpy (pages made in one call, no traps)
//   cfr       : copy_file_range from an input file of the same size (the bytes never enter user space)
//   sendfile  : sendfile at the output's own position is one stream: not tried
// g++ -O2 -pthread tools/out_probe.cc -o /tmp/out_probe && /tmp/out_probe <dir> <GB> <threads> [way ...]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
using clk = std::chrono::steady_clock;
static double sec(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }
int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string dir = argv[1];
    const size_t total = (size_t)(atof(argv[2]) * (1u << 30));
    const int nt = atoi(argv[3]);
    const size_t chunk = (size_t)16 << 20;
    const size_t n_chunks = total / chunk;
    const std::string in_path = dir + "/out_probe.in", out_path = dir + "/out_probe.out";
    std::vector<char> src(chunk);
    for (size_t i = 0; i < chunk; i++) src[i] = (char)('A' + i % 23);
    auto run = [&](auto fn) {
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&] {
                for (size_t i; (i = next.fetch_add(1)) < n_chunks;) fn(i);
            });
        for (auto &x : th) x.join();
    };
    bool have_in = false;
    for (int a = 4; a < argc || a == 4; a++) {
        const std::string way = a < argc ? argv[a] : "pwrite";
        if (way == "cfr" && !have_in) {
            const int fd = open(in_path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
            for (size_t i = 0; i < n_chunks; i++)
                if (pwrite(fd, src.data(), chunk, (off_t)(i * chunk)) != (ssize_t)chunk) return 1;
            close(fd);
            have_in = true;
        }
        const int ofd = open(out_path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
        if (ofd < 0) return 1;
        const auto t0 = clk::now();
        std::atomic<int> bad(0);
        if (way == "pwrite") {
            run([&](size_t i) {
                for (size_t done = 0; done < chunk;) {
                    const ssize_t k = pwrite(ofd, src.data() + done, chunk - done, (off_t)(i * chunk + done));
                    if (k <= 0) { bad = 1; return; }
                    done += (size_t)k;
                }
            });
        } else if (way == "map" || way == "populate") {
            if (ftruncate(ofd, (off_t)total) != 0) return 1;
            char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, ofd, 0);
            if (m == MAP_FAILED) return 1;
            const bool pop = way == "populate";
            run([&](size_t i) {
                if (pop && madvise(m + i * chunk, chunk, MADV_POPULATE_WRITE) != 0) bad = 2;
                std::memcpy(m + i * chunk, src.data(), chunk);
            });
            munmap(m, total);
        } else if (way == "cfr") {
            const int ifd = open(in_path.c_str(), O_RDONLY);
            run([&](size_t i) {
                off_t oi = (off_t)(i * chunk), oo = oi;
                for (size_t left = chunk; left;) {
                    const ssize_t k = copy_file_range(ifd, &oi, ofd, &oo, left, 0);
                    if (k <= 0) { bad = 3; return; }
                    left -= (size_t)k;
                }
            });
            close(ifd);
        }
        const double t_write = sec(t0);
        close(ofd);
        const double t_all = sec(t0);
        printf("%-9s %2d threads: %.2f s (close %.2f) = %.2f GB/s%s\n", way.c_str(), nt, t_all, t_all - t_write, (double)total / t_all / 1e9,
               bad ? "  FAILED" : "");
        fflush(stdout);
        unlink(out_path.c_str());
    }
    unlink(in_path.c_str());
    return 0;
}
