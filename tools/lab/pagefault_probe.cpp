#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <cstring>
using clk = std::chrono::steady_clock;
static double touch(char *p, size_t bytes, int T) {
    auto t0 = clk::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([=] { size_t b = bytes * t / T, e = bytes * (t + 1) / T; for (size_t i = b; i < e; i += 64) p[i] = (char)i; });
    for (auto &x : th) x.join();
    return std::chrono::duration<double, std::milli>(clk::now() - t0).count();
}
int main(int argc, char **argv) {
    int T = argc > 1 ? atoi(argv[1]) : 8;
    size_t bytes = (size_t)256 << 20;
    char *p = (char *)malloc(bytes);
    printf("malloc first touch  %d thr: %.1f ms\n", T, touch(p, bytes, T));
    printf("malloc second touch %d thr: %.1f ms\n", T, touch(p, bytes, T));
    free(p);
    char *q = (char *)aligned_alloc(2 << 20, bytes);
    int rc = madvise(q, bytes, MADV_HUGEPAGE);
    printf("madvise rc %d; hugepage first touch %d thr: %.1f ms\n", rc, T, touch(q, bytes, T));
    free(q);
    char *r = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    printf("populate then touch: %.1f ms\n", touch(r, bytes, T));
    auto t0 = clk::now();
    char *s = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    printf("mmap populate itself: %.1f ms\n", std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    (void)s;
}
