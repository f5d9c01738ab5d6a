// two_streams.cpp -- does RCCL accept the stream usage of libfemshell's CG driver?  One communicator, grouped
// send/recv (the halo exchange) on a second stream behind an event, all-reduces on the main stream, alternating,
// as cg_driver.cpp issues them.  One rank talks to itself (a test box has one GPU; RCCL refuses two ranks on one
// device), so this exercises RCCL's enqueue / stream-ordering logic, not its transports.  Test infrastructure only.
// Build: hipcc -O2 two_streams.cpp -o two_streams -ldl      Run: ./two_streams [iterations]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define NK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("RCCL error %d at line %d\n", (int)r_, __LINE__); return 3; } } while (0)

__global__ void k_fill(double *p, int n, double v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + i;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { printf("cannot open librccl.so.1: %s\n", dlerror()); return 1; }
#define SYM(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); if (!p_##name) { printf("missing %s\n", #name); return 1; }
    SYM(ncclGetUniqueId) SYM(ncclCommInitRank) SYM(ncclCommDestroy) SYM(ncclAllReduce) SYM(ncclSend) SYM(ncclRecv) SYM(ncclGroupStart) SYM(ncclGroupEnd)
    CK(hipSetDevice(0));
    ncclUniqueId id;
    NK(p_ncclGetUniqueId(&id));
    ncclComm_t comm;
    NK(p_ncclCommInitRank(&comm, 1, id, 0));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t ev_ready, ev_done;
    CK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    const int n = 1415 * 6; // one strip boundary of the 4M-triangle panel
    double *src, *dst, *red;
    CK(hipMalloc(&src, n * sizeof(double)));
    CK(hipMalloc(&dst, n * sizeof(double)));
    CK(hipMalloc(&red, 3 * sizeof(double)));
    std::vector<double> h(n), hr(3);
    for (int it = 0; it < iters; it++) {
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, sa, src, n, (double)it);  // "direction update"
        hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, sa, red, 3, 100.0 * it);
        CK(hipEventRecord(ev_ready, sa));
        CK(hipStreamWaitEvent(sb, ev_ready, 0));
        NK(p_ncclGroupStart());                                                                    // halo exchange
        NK(p_ncclSend(src, n, ncclDouble, 0, comm, sb));
        NK(p_ncclRecv(dst, n, ncclDouble, 0, comm, sb));
        NK(p_ncclGroupEnd());
        CK(hipEventRecord(ev_done, sb));
        hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, sa, red, 3, 100.0 * it);                 // "interior SpMV"
        CK(hipStreamWaitEvent(sa, ev_done, 0));
        NK(p_ncclAllReduce(red, red, 3, ncclDouble, ncclSum, comm, sa));                           // CG sums
        if (it % 50 == 49 || it + 1 == iters) {
            CK(hipMemcpyAsync(h.data(), dst, n * sizeof(double), hipMemcpyDeviceToHost, sa));
            CK(hipMemcpyAsync(hr.data(), red, 3 * sizeof(double), hipMemcpyDeviceToHost, sa));
            CK(hipStreamSynchronize(sa));
            for (int i = 0; i < n; i++)
                if (h[i] != (double)it + i) { printf("halo payload wrong at iteration %d, word %d\n", it, i); return 4; }
            for (int i = 0; i < 3; i++)
                if (hr[i] != 100.0 * it + i) { printf("all-reduce result wrong at iteration %d\n", it); return 5; }
        }
    }
    CK(hipStreamSynchronize(sa));
    CK(hipStreamSynchronize(sb));
    NK(p_ncclCommDestroy(comm));
    printf("two_streams ok: %d iterations\n", iters);
    return 0;
}
