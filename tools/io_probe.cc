// tools/io_probe.cc — how fast can N threads get a big text file into user space on this box?
//   (a) mmap(MAP_PRIVATE) + touch every page (what a parser that walks the mapping pays)
//   (b) parallel pread() into an anonymous buffer, with and without MADV_HUGEPAGE
// g++ -O2 -pthread tools/io_probe.cc -o /tmp/io_probe && /tmp/io_probe <file> <threads>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
int main(int argc, char **argv)
{
    const char *path = argv[1];
    const int nt = atoi(argv[2]);
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    {
        auto t0 = clk::now();
        char *m = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        volatile unsigned long sink = 0;
        run([&](int t) {
            unsigned long s = 0;
            for (size_t i = n / nt * t; i < (t + 1 == nt ? n : n / nt * (t + 1)); i += 4096) s += (unsigned char)m[i];
            sink += s;
        });
        double t_touch = ms(t0);
        auto t1 = clk::now();
        munmap(m, n);
        printf("mmap+touch %d threads: %.1f ms (%.2f GB/s), munmap %.1f ms\n", nt, t_touch, n / t_touch / 1e6, ms(t1));
    }
    for (int huge = 0; huge < 2; huge++) {
        auto t0 = clk::now();
        char *buf = (char *)mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (huge) madvise(buf, n + (2 << 20), MADV_HUGEPAGE);
        run([&](int t) {
            size_t a = n / nt * t, b = (t + 1 == nt ? n : n / nt * (t + 1));
            while (a < b) {
                ssize_t g = pread(fd, buf + a, std::min<size_t>(b - a, 8u << 20), (off_t)a);
                if (g <= 0) break;
                a += (size_t)g;
            }
        });
        double t_read = ms(t0);
        auto t1 = clk::now();
        munmap(buf, n + (2 << 20));
        printf("pread into anon%s %d threads: %.1f ms (%.2f GB/s), munmap %.1f ms\n", huge ? "+THP" : "", nt, t_read, n / t_read / 1e6, ms(t1));
    }
    return 0;
}
