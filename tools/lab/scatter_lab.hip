// scatter_lab.hip -- the mapping SURVEY section 7 / the north star prescribe for the assembly, measured: element-parallel
// computation of the 18 x 18 element stiffness and a SCATTER of its nine 6 x 6 blocks into the global block matrix with FP64
// atomic adds (what libMesh's add_matrix does on the CPU), against the row-owner gather the library ships (k_assemble_pipe:
// no atomics, every block of K written once).  Two variants:
//   wave : one element per wavefront -- lane 0 builds the element record, lanes 0..8 a block each, all 64 lanes issue the
//          324 atomic adds, 36 consecutive doubles per block
//   lane : one element per lane -- 64 elements per wavefront, every lane builds its record and adds its nine blocks
// The element arithmetic is the library's (csrc/shell_element.hpp), the matrix a full-storage block CSR with ascending
// columns built on the host; FP64 only; no Dirichlet handling (neither variant would pay for it).  Prints ms per assembly,
// elements/s, and the Frobenius norm and trace of K for a check against the library's K (tools/lab/scatter_check.py).
//   usage: scatter_lab NX [repetitions]      build: tools/lab/build_scatter_lab.sh
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "shell_element.hpp"

using namespace femshell;

#define CHECK(x)                                                                                           \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) {                                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                                 \
            exit(1);                                                                                       \
        }                                                                                                  \
    } while (0)

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); } // global_atomic_add_f64

// one element per wavefront
__global__ __launch_bounds__(256) void k_scatter_wave(const double *__restrict__ xyz, const int32_t *__restrict__ tri,
                                                      const int32_t *__restrict__ dest, double *vals, int32_t n_elem, MatConst mc)
{
    __shared__ double rec_s[4][kRecDoubles];
    __shared__ double blk_s[4][9 * 36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + wave;
    if (e >= n_elem) return; // (whole waves leave together)
    if (lane == 0) {
        double X[9];
        for (int i = 0; i < 3; i++)
            for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3ll * tri[3ll * e + i] + d];
        double rec[kRecDoubles];
        tri3_record(X, mc, rec);
        for (int q = 0; q < kRecDoubles; q++) rec_s[wave][q] = rec[q];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane < 9) {
        double acc[36];
        for (int q = 0; q < 36; q++) acc[q] = 0.0;
        tri3_block_add_rec<RecFull>(rec_s[wave], lane / 3, lane % 3, mc, acc);
        for (int q = 0; q < 36; q++) blk_s[wave][lane * 36 + q] = acc[q];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // 324 atomic adds on 64 lanes: entry q of block b goes to vals[dest[e][b] * 36 + q]
    for (int idx = lane; idx < 324; idx += 64) {
        const int b = idx / 36, q = idx % 36;
        atomic_add_f64(vals + (int64_t)dest[9ll * e + b] * 36 + q, blk_s[wave][idx]);
    }
}

// one element per lane
__global__ __launch_bounds__(64) void k_scatter_lane(const double *__restrict__ xyz, const int32_t *__restrict__ tri,
                                                     const int32_t *__restrict__ dest, double *vals, int32_t n_elem, MatConst mc)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n_elem) return;
    double X[9];
    for (int i = 0; i < 3; i++)
        for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3ll * tri[3ll * e + i] + d];
    double rec[kRecDoubles];
    tri3_record(X, mc, rec);
    for (int b = 0; b < 9; b++) {
        double acc[36];
        for (int q = 0; q < 36; q++) acc[q] = 0.0;
        tri3_block_add_rec<RecFull>(rec, b / 3, b % 3, mc, acc);
        double *dst = vals + (int64_t)dest[9ll * e + b] * 36;
        for (int q = 0; q < 36; q++) atomic_add_f64(dst + q, acc[q]);
    }
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 1414, reps = argc > 2 ? atoi(argv[2]) : 10;
    const int n1 = nx + 1;
    const int32_t n_nodes = n1 * n1, n_elem = 2 * nx * nx;
    std::vector<double> xyz((size_t)n_nodes * 3);
    std::vector<int32_t> tri((size_t)n_elem * 3);
    for (int j = 0; j < n1; j++)
        for (int i = 0; i < n1; i++) {
            double *x = &xyz[3 * ((size_t)j * n1 + i)];
            x[0] = 10.0 * i / nx, x[1] = 10.0 * j / nx, x[2] = 0.0;
        }
    for (int j = 0, e = 0; j < nx; j++)
        for (int i = 0; i < nx; i++) { // the two triangles of a cell (the diagonal from lower left to upper right)
            const int a = j * n1 + i, b = a + 1, c = a + n1, d = c + 1;
            const int t0[3] = {a, b, d}, t1[3] = {a, d, c};
            for (int k = 0; k < 3; k++) tri[3 * (size_t)e + k] = t0[k];
            e++;
            for (int k = 0; k < 3; k++) tri[3 * (size_t)e + k] = t1[k];
            e++;
        }
    // full-storage block CSR, ascending columns
    std::vector<std::vector<int32_t>> cols((size_t)n_nodes);
    for (int32_t e = 0; e < n_elem; e++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) cols[(size_t)tri[3 * (size_t)e + a]].push_back(tri[3 * (size_t)e + b]);
    std::vector<int64_t> ptr((size_t)n_nodes + 1, 0);
    for (int32_t n = 0; n < n_nodes; n++) {
        auto &c = cols[(size_t)n];
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        ptr[(size_t)n + 1] = ptr[(size_t)n] + (int64_t)c.size();
    }
    const int64_t nnzb = ptr[(size_t)n_nodes];
    std::vector<int32_t> dest((size_t)n_elem * 9);
    for (int32_t e = 0; e < n_elem; e++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                const int32_t r = tri[3 * (size_t)e + a], cnode = tri[3 * (size_t)e + b];
                const auto &c = cols[(size_t)r];
                dest[9 * (size_t)e + 3 * a + b] = (int32_t)(ptr[(size_t)r] + (std::lower_bound(c.begin(), c.end(), cnode) - c.begin()));
            }
    MatConst mc{};
    const double nu = 0.3, E = 1e7, t = 0.5;
    mc.cm = E / (1.0 - nu * nu);
    mc.cp = E * t * t * t / (12.0 * (1.0 - nu * nu));
    mc.nu = nu;
    mc.g = (1.0 - nu) / 2.0;
    mc.t = t;
    mc.flags = kRefY21 | kRefDrillMax; // the reference's as-coded behaviour: the library's default
    double *d_xyz, *d_vals;
    int32_t *d_tri, *d_dest;
    CHECK(hipMalloc(&d_xyz, xyz.size() * 8));
    CHECK(hipMalloc(&d_tri, tri.size() * 4));
    CHECK(hipMalloc(&d_dest, dest.size() * 4));
    CHECK(hipMalloc(&d_vals, (size_t)nnzb * 36 * 8));
    CHECK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tri, tri.data(), tri.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dest, dest.data(), dest.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("panel %d x %d: %d triangles, %d nodes, %lld blocks of K (full storage, %.2f GB)\n", nx, nx, n_elem, n_nodes, (long long)nnzb,
           (double)nnzb * 288e-9);
    for (int variant = 0; variant < 2; variant++) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 2; r++) {
            CHECK(hipMemsetAsync(d_vals, 0, (size_t)nnzb * 36 * 8, 0)); // (the scatter needs K zeroed first: not timed)
            CHECK(hipEventRecord(e0, 0));
            if (variant == 0) hipLaunchKernelGGL(k_scatter_wave, dim3((n_elem + 3) / 4), dim3(256), 0, 0, d_xyz, d_tri, d_dest, d_vals, n_elem, mc);
            else hipLaunchKernelGGL(k_scatter_lane, dim3((n_elem + 63) / 64), dim3(64), 0, 0, d_xyz, d_tri, d_dest, d_vals, n_elem, mc);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) {
                best = std::min(best, ms);
                sum += ms;
            }
        }
        std::vector<double> h((size_t)nnzb * 36);
        CHECK(hipMemcpy(h.data(), d_vals, h.size() * 8, hipMemcpyDeviceToHost));
        long double fro = 0.0L, trace = 0.0L;
        for (double v : h) fro += (long double)v * v;
        for (int32_t n = 0; n < n_nodes; n++) {
            const auto &c = cols[(size_t)n];
            const int64_t d = ptr[(size_t)n] + (std::lower_bound(c.begin(), c.end(), n) - c.begin());
            for (int i = 0; i < 6; i++) trace += h[(size_t)d * 36 + 7 * i];
        }
        printf("%-28s %8.3f ms mean, %8.3f ms best = %6.3f G elements/s; 324 atomic adds per element = %.1f G atomics/s; "
               "||K||_F = %.15e  trace = %.15e\n",
               variant == 0 ? "one element per wavefront:" : "one element per lane:", sum / reps, best, n_elem / (sum / reps) * 1e-6,
               324.0 * n_elem / (sum / reps) * 1e-6, (double)sqrtl(fro), (double)trace);
    }
    return 0;
}
