// asm_lab.hip -- throwaway ablation harness for the assembly kernel (not part of the product).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../include -I../../fem-shell_amd/csrc asm_lab.hip ../../fem-shell_amd/csrc/plan.cpp -o asm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.hpp"
#include "plan.hpp"
using namespace femshell;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct SliceWalk {
    int per, first, last, step, s;
    __device__ __forceinline__ SliceWalk(int n_slices) {
        per = (n_slices + 7) >> 3; const int x = blockIdx.x & 7; first = x * per; last = min(first + per, n_slices);
        step = gridDim.x >> 3; s = first + (blockIdx.x >> 3);
    }
    __device__ __forceinline__ bool valid() const { return s < last; }
    __device__ __forceinline__ void next() { s += step; }
};

// flags: bit0 = skip global stores, bit1 = all lanes read record 0, bit2 = skip block compute, bit3 = skip phase A compute
template <int W, int F>
__global__ __launch_bounds__(256, W) void k_lab(DeviceMatrix m, MatConst mc, double *sink)
{
    extern __shared__ double lds_rec[];
    double *lds_stage = lds_rec + (size_t)m.max_slice_elems * kRecDoubles;
    double chk = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int s = w.s;
        const int e0 = m.slice_elem_ptr[s], ne = m.slice_elem_ptr[s + 1] - e0;
        __syncthreads();
        for (int i = threadIdx.x; i < ne; i += blockDim.x) {
            const int le = m.slice_elems[e0 + i];
            double rec[kRecDoubles];
            if (F & 8) {
#pragma unroll
                for (int q = 0; q < kRecDoubles; q++) rec[q] = 1.0 + 0.01 * q + le * 1e-9;
            } else {
                const int32_t *c = m.tri + 3 * (int64_t)le;
                double X[9];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const double *pt = m.xyz + 3 * (int64_t)c[q];
                    X[3 * q + 0] = pt[0]; X[3 * q + 1] = pt[1]; X[3 * q + 2] = pt[2];
                }
                tri3_record(X, mc, rec);
            }
            double2 *dst = reinterpret_cast<double2 *>(lds_rec + (size_t)i * kRecDoubles);
#pragma unroll
            for (int q = 0; q < kRecDoubles / 2; q++) dst[q] = make_double2(rec[2 * q], rec[2 * q + 1]);
        }
        __syncthreads();
        const int64_t base = m.slice_base[s];
        double2 *out = reinterpret_cast<double2 *>(m.vals + base * 36);
        const uint16_t *pairs = m.pairs16 + m.pair_ptr[base];
        const int i0 = m.item_ptr[s], ni = m.item_ptr[s + 1] - i0;
        for (int r0 = 0; r0 < ni; r0 += blockDim.x) {
            const int it = r0 + threadIdx.x;
            const bool live = it < ni;
            uint4 item = make_uint4(0, 0, 0, 0);
            if (live) item = m.items[i0 + it];
            const int slot_in_slice = (int)(item.x & 0xffffu), chunk = (int)((item.x >> 16) & 0xffu), nchunks = (int)(item.x >> 24);
            const int q0 = (int)(item.y & 0xffffffu), cnt = (int)(item.y >> 24);
            double acc[36];
#pragma unroll
            for (int i = 0; i < 36; i++) acc[i] = 0.0;
            for (int q = 0; q < cnt; q++) {
                const uint32_t pr = pairs[q0 + q];
                const double *rec = lds_rec + ((F & 2) ? 0 : (size_t)(pr >> 4) * kRecDoubles);
                if (F & 4) {
#pragma unroll
                    for (int i = 0; i < 26; i++) acc[i] += rec[i];
                } else {
                    tri3_block_add_rec(rec, (int)((pr >> 2) & 3u), (int)(pr & 3u), mc, acc);
                }
            }
            if (live && chunk > 0) {
                double2 *st = reinterpret_cast<double2 *>(lds_stage + (size_t)item.z * 36);
#pragma unroll
                for (int i = 0; i < 18; i++) st[i] = make_double2(acc[2 * i], acc[2 * i + 1]);
            }
            __syncthreads();
            if (live && chunk == 0) {
                for (int c = 1; c < nchunks; c++) {
                    const double2 *st = reinterpret_cast<const double2 *>(lds_stage + (size_t)(item.z + c - 1) * 36);
#pragma unroll
                    for (int i = 0; i < 18; i++) { const double2 v = st[i]; acc[2 * i] += v.x; acc[2 * i + 1] += v.y; }
                }
                const int k = slot_in_slice >> 5, n = slot_in_slice & 31;
                if (F & 1) {
#pragma unroll
                    for (int i = 0; i < 36; i++) chk += acc[i];
                } else {
#pragma unroll
                    for (int jp = 0; jp < 3; jp++)
#pragma unroll
                        for (int i = 0; i < 6; i++)
                            out[(k * 3 + jp) * kSliceRows + n * 6 + i] = make_double2(acc[6 * i + 2 * jp], acc[6 * i + 2 * jp + 1]);
                }
            }
            __syncthreads();
        }
    }
    if ((F & 1) && chk == 1.2345e300) sink[0] = chk;
}

template <int W, int F> float run(const DeviceMatrix &m, const MatConst &mc, double *sink, int grid, int reps)
{
    const size_t lds = ((size_t)m.max_slice_elems * kRecDoubles + (size_t)m.max_stage_rows * 36) * sizeof(double);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_lab<W, F>), dim3(grid), dim3(256), lds, 0, m, mc, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_lab<W, F>), dim3(grid), dim3(256), lds, 0, m, mc, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <class T> T *up(const std::vector<T> &v) { T *d = nullptr; CK(hipMalloc(&d, std::max<size_t>(1, v.size()) * sizeof(T))); if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 1414;
    const int grid = argc > 2 ? atoi(argv[2]) : 2048;
    const int nn = (nx + 1) * (nx + 1);
    std::vector<double> xyz(3 * (size_t)nn);
    for (int j = 0; j <= nx; j++) for (int i = 0; i <= nx; i++) { size_t a = (size_t)j * (nx + 1) + i; xyz[3*a] = 10.0 * i / nx; xyz[3*a+1] = 10.0 * j / nx; xyz[3*a+2] = 0.0; }
    std::vector<int32_t> tri; tri.reserve(6 * (size_t)nx * nx);
    for (int j = 0; j < nx; j++) for (int i = 0; i < nx; i++) { int n = i + j * (nx + 1), up_ = nx + 1; tri.insert(tri.end(), {n, n + 1, n + up_, n + 1, n + up_ + 1, n + up_}); }
    Plan p; std::string err;
    if (!build_plan(nn, xyz.data(), (int)tri.size() / 3, tri.data(), 0, nullptr, 0, 1, &p, &err)) { printf("plan: %s\n", err.c_str()); return 1; }
    DeviceMatrix m;
    m.n_own = p.n_own; m.n_pad = p.n_pad; m.n_ghost = p.n_ghost; m.n_slices = p.n_slices; m.n_ltri = p.n_ltri(); m.n_lquad = 0;
    m.xyz = up(p.xyz_local); m.tri = up(p.tri_local); m.slice_width = up(p.slice_width); m.slice_base = up(p.slice_base);
    m.cols = up(p.cols); m.pair_ptr = up(p.pair_ptr); m.pairs16 = up(p.pairs16); m.slice_elem_ptr = up(p.slice_elem_ptr);
    m.slice_elems = up(p.slice_elems); m.max_slice_elems = p.max_slice_elems; m.item_ptr = up(p.item_ptr);
    m.items = reinterpret_cast<const uint4 *>(up(p.items)); m.max_stage_rows = p.max_stage_rows;
    std::vector<uint8_t> dm(p.n_local_nodes(), 0); m.dmask = up(dm);
    double *vals; CK(hipMalloc(&vals, (size_t)p.total_slots() * 36 * 8)); m.vals = vals;
    double *sink; CK(hipMalloc(&sink, 8));
    MatConst mc; const double nu = 0.3, E = 1e7, t = 0.5;
    mc.cm = E / (1 - nu * nu); mc.cp = E * t * t * t / (12 * (1 - nu * nu)); mc.nu = nu; mc.g = (1 - nu) / 2; mc.t = t; mc.flags = 3; mc.pad = 0;
    printf("nx=%d slices=%d max_elems=%d max_stage=%d items=%zu pairs=%zu grid=%d\n", nx, p.n_slices, p.max_slice_elems, p.max_stage_rows, p.items.size(), p.pairs.size(), grid);
    const int R = 5;
    printf("W1 full            : %.3f ms\n", run<1, 0>(m, mc, sink, grid, R));
    printf("W2 full            : %.3f ms\n", run<2, 0>(m, mc, sink, grid, R));
    printf("W2 no stores       : %.3f ms\n", run<2, 1>(m, mc, sink, grid, R));
    printf("W2 rec0 broadcast  : %.3f ms\n", run<2, 2>(m, mc, sink, grid, R));
    printf("W2 no block compute: %.3f ms\n", run<2, 4>(m, mc, sink, grid, R));
    printf("W2 no phaseA math  : %.3f ms\n", run<2, 8>(m, mc, sink, grid, R));
    printf("W2 nostore+rec0    : %.3f ms\n", run<2, 3>(m, mc, sink, grid, R));
    printf("W2 nostore+nocomp  : %.3f ms\n", run<2, 5>(m, mc, sink, grid, R));
    printf("W4 no block compute: %.3f ms\n", run<4, 4>(m, mc, sink, grid, R));
    return 0;
}
