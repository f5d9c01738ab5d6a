// What does hipMalloc cost on this box, by size, and do calls from several host threads overlap?  (round 6: 39 calls of a multigrid
// setup at 32M triangles took 2.6 s, 16 calls at 4M between 0.8 and 73 ms)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    (void)hipSetDevice(0);
    void *w = nullptr;
    (void)hipMalloc(&w, 1 << 20);
    for (int round = 0; round < 2; round++) {
        for (double gb : {0.1, 0.5, 2.0, 8.0, 24.0}) {
            const size_t bytes = (size_t)(gb * 1073741824.0);
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, bytes);
            double t1 = now();
            (void)hipMemsetAsync(p, 0, 256, 0);
            (void)hipDeviceSynchronize();
            double t2 = now();
            (void)hipFree(p);
            double t3 = now();
            printf("round %d: hipMalloc %5.1f GB %8.2f ms (%s), first touch %6.2f ms, hipFree %8.2f ms\n", round, gb, 1e3 * (t1 - t0), hipGetErrorString(e), 1e3 * (t2 - t1), 1e3 * (t3 - t2));
        }
    }
    // four blocks of 8 GB: one after the other, then from four threads at once
    for (int round = 0; round < 2; round++) {
        std::vector<void *> p(4, nullptr);
        double t0 = now();
        for (int i = 0; i < 4; i++) (void)hipMalloc(&p[i], (size_t)8 << 30);
        double t1 = now();
        for (int i = 0; i < 4; i++) (void)hipFree(p[i]);
        double t2 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < 4; i++) th.emplace_back([&p, i] { (void)hipSetDevice(0); (void)hipMalloc(&p[i], (size_t)8 << 30); });
        for (auto &t : th) t.join();
        double t3 = now();
        for (int i = 0; i < 4; i++) (void)hipFree(p[i]);
        printf("round %d: 4 x 8 GB in sequence %8.2f ms (free %8.2f ms), from four threads %8.2f ms\n", round, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2));
    }
    // does a kernel launch wait for a hipMalloc in another thread?
    {
        void *big = nullptr;
        std::thread th([&big] { (void)hipSetDevice(0); (void)hipMalloc(&big, (size_t)24 << 30); });
        double t0 = now();
        int n = 0;
        while (now() - t0 < 0.05) {
            (void)hipMemsetAsync(w, 0, 256, 0);
            (void)hipStreamSynchronize(0);
            n++;
        }
        th.join();
        double t1 = now();
        printf("while another thread allocated 24 GB (%.1f ms in all): %d memset + sync round trips in the first 50 ms\n", 1e3 * (t1 - t0), n);
        (void)hipFree(big);
    }
    return 0;
}
