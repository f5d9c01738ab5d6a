// device_common.hpp -- device helpers shared by the kernel translation units (kernels.hip, amg_kernels.hip):
// the XCD-aware slice walk, workgroup sums, the packed block-Jacobi inverse.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "plan.hpp"

namespace femshell {

// Workgroup b belongs to XCD group x = b%8 and walks the slices x*per + j, j = b/8, b/8 + G/8, ...
// of that group's contiguous eighth of the rows (per = ceil(S/8), G = gridDim.x).
struct SliceWalk {
    int per, first, last, step, s;
    __device__ __forceinline__ SliceWalk(int n_slices)
    {
        per = (n_slices + 7) >> 3;
        const int x = blockIdx.x & 7;
        first = x * per;
        last = min(first + per, n_slices);
        step = gridDim.x >> 3;
        s = first + (blockIdx.x >> 3);
    }
    __device__ __forceinline__ bool valid() const { return s < last; }
    __device__ __forceinline__ void next() { s += step; }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sum over the workgroup, valid in thread 0; sh must hold blockDim.x/64 doubles
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; i++) t += sh[i];
    __syncthreads();
    return t;
}

// minv layout: per slice 21 words (upper triangle of the symmetric 6x6 inverse, row-major) x 32 nodes, nodes fastest
constexpr int kMinvWords = 21;
__host__ __device__ __forceinline__ int minv_word(int i, int j) { return (i * (11 - i)) / 2 + j; } // i <= j

// z_row = sum_j Minv[row][j] * r[node*6+j].  The six Minv entries of the row are fetched before the
// residual of the slice is exchanged through LDS, so their latency overlaps the barrier.
struct MinvRow {
    double a[6];
};
__device__ __forceinline__ MinvRow load_minv(const DeviceMatrix &m, int sl, int t)
{
    const int n = t / 6, i = t % 6;
    const double *mi = m.minv + (int64_t)sl * kMinvWords * kSliceNodes + n;
    MinvRow r;
#pragma unroll
    for (int j = 0; j < 6; j++) r.a[j] = mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
    return r;
}
// the smoothers of the multigrid cycle: from the single-precision copy when the level has one
__device__ __forceinline__ MinvRow load_minv_smoother(const DeviceMatrix &m, int sl, int t)
{
    if (m.minv32 == nullptr) return load_minv(m, sl, t);
    const int n = t / 6, i = t % 6;
    const float *mi = m.minv32 + (int64_t)sl * kMinvWords * kSliceNodes + n;
    MinvRow r;
#pragma unroll
    for (int j = 0; j < 6; j++) r.a[j] = (double)mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
    return r;
}
__device__ __forceinline__ double apply_minv(const MinvRow &mr, int t, const double *rs)
{
    const int nb = (t / 6) * 6;
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) z += mr.a[j] * rs[nb + j];
    return z;
}


} // namespace femshell
