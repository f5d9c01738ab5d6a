// hbm_rw.hip -- streaming write / read / copy rates of the device (profiling aid, not part of the product).
// Build: hipcc -O3 --offload-arch=gfx950 hbm_rw.hip -o hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_write(double2 *dst, size_t n, double v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_double2(v, v + 1.0);
}
__global__ __launch_bounds__(256) void k_read(const double2 *src, size_t n, double *out)
{
    double a = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        a += v.x + v.y;
    }
    if (a == 1.2345e300) out[0] = a;
}
__global__ __launch_bounds__(256) void k_copy(const double2 *src, double2 *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// one workgroup writes contiguous 64.5 KB chunks (the store pattern of k_assemble)
__global__ __launch_bounds__(256) void k_write_chunks(double2 *dst, size_t nchunks, int words, double v)
{
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        double2 *d = dst + c * (size_t)words;
        for (int q = threadIdx.x; q < words; q += blockDim.x) d[q] = make_double2(v, v);
    }
}

int main(int argc, char **argv)
{
    const bool json = argc > 1 && std::string(argv[1]) == "--json"; // one line for bench.py: the best rate of each direction
    const size_t bytes = (size_t)4300 << 20, n = bytes / 16;
    double2 *a, *b; double *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn, const char *name, double gb) {
        fn(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        if (!json) printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, gb / ms);
        return gb / ms * 1e3; // GB/s
    };
    const double gb = bytes / 1e9;
    double best_w = 0.0, best_r = 0.0, best_c = 0.0;
    for (int grid : {2048, 8192, 65536}) {
        if (!json) printf("grid %d\n", grid);
        best_w = std::max(best_w, time([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n, 1.0); }, "write 16 B/lane", gb));
        best_r = std::max(best_r, time([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o); }, "read 16 B/lane", gb));
        best_c = std::max(best_c, time([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy (read+write bytes)", 2 * gb));
    }
    if (json) {
        printf("{\"write_gb_per_s\": %.1f, \"read_gb_per_s\": %.1f, \"copy_gb_per_s\": %.1f, \"bytes\": %zu, "
               "\"note\": \"best of grids 2048 / 8192 / 65536 x 256 threads, 16 B per lane, 5 launches each (tools/lab/hbm_rw.hip)\"}\n",
               best_w, best_r, best_c, bytes);
        return 0;
    }
    const int words = 4032; // 7 slots x 32 nodes x 18 double2
    time([&] { hipLaunchKernelGGL(k_write_chunks, dim3(2048), dim3(256), 0, 0, a, n / words, words, 2.0); }, "write 64.5 KB chunks/WG", gb);
    time([&] { hipLaunchKernelGGL(k_write_chunks, dim3(512), dim3(256), 0, 0, a, n / words, words, 2.0); }, "same, 512 WGs", gb);
    time([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); }, "hipMemsetAsync", gb);
    return 0;
}
