// device_common.hpp -- device helpers shared by the kernel translation units (kernels.hip, amg_kernels.hip):
// the XCD-aware slice walk, workgroup sums, the packed block-Jacobi inverse.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "plan.hpp"

namespace femshell {

// first failure wins: the status word (HBM, or host memory mapped into the device: context.hpp status_word) gets `value` if it
// still holds zero -- a system-scope compare-and-swap, executed on failures only
__device__ __forceinline__ void report_status(int32_t *status, int32_t value)
{
    int32_t expected = 0;
    (void)__hip_atomic_compare_exchange_strong(status, &expected, value, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


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

// Symmetric storage, second phase of a product for scalar row (node n, component j) of slice sl: acc + the transposed products
// of the row's in-list, in the plan's fixed order (the loop of k_sym_gather / k_cg_update<true> / k_cheb_step<true>: the slot
// indices of the first four entries together, then their products together -- one entry at a time is two dependent memory
// round trips per entry; the order of the additions is the same).  kT32: the products were stored as floats (DeviceMatrix::vec32)
template <bool kT32> __device__ __forceinline__ double gather_transposed(const DeviceMatrix &m, int sl, int n, int j, double acc)
{
    const float *tf = reinterpret_cast<const float *>(m.tbuf);
    const int Wi = m.in_width[sl];
    const int64_t ib = m.in_base[sl];
    int32_t slot4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) slot4[k] = (k < Wi) ? m.gat_slots[ib + (int64_t)k * kSliceNodes + n] : -1;
    double t4[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        t4[k] = slot4[k] < 0 ? 0.0 : (kT32 ? (double)tf[(int64_t)slot4[k] * 6 + j] : m.tbuf[(int64_t)slot4[k] * 6 + j]);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (slot4[k] >= 0) acc += t4[k];
    for (int k = 4; k < Wi; k++) {
        const int32_t slot = m.gat_slots[ib + (int64_t)k * kSliceNodes + n];
        if (slot >= 0) acc += kT32 ? (double)tf[(int64_t)slot * 6 + j] : m.tbuf[(int64_t)slot * 6 + j];
    }
    return acc;
}

// ---- one lane per NODE (round 5) ---------------------------------------------------------------------------------------
// The vector kernels that apply the 6x6 block-Jacobi inverse used one lane per scalar row and exchanged the residual of a node
// through LDS; their 4-byte loads of the inverse (21 words, nodes fastest) scatter over six words per wave instruction and two
// barriers per slice keep few loads in flight: 3.8-4.4 TB/s where the plain vector passes reach 6 (round 5, PMC).  With a lane
// per node -- the mapping of k_spmv_sym -- a half-wave reads one word of the inverse for its 32 nodes as one 128/256-byte
// segment, the six entries of a vector as three 16-byte words, and the apply is lane-local: no LDS, no barrier.
__device__ __forceinline__ void load_node6(const double *x, int64_t node, bool as_float, double out[6])
{
    if (as_float) {
        const float2 *xf = reinterpret_cast<const float2 *>(x) + 3 * node;
        const float2 a0 = xf[0], a1 = xf[1], a2 = xf[2];
        out[0] = a0.x; out[1] = a0.y; out[2] = a1.x; out[3] = a1.y; out[4] = a2.x; out[5] = a2.y;
    } else {
        const double2 *xd = reinterpret_cast<const double2 *>(x) + 3 * node;
        const double2 a0 = xd[0], a1 = xd[1], a2 = xd[2];
        out[0] = a0.x; out[1] = a0.y; out[2] = a1.x; out[3] = a1.y; out[4] = a2.x; out[5] = a2.y;
    }
}
__device__ __forceinline__ void store_node6(double *x, int64_t node, bool as_float, const double v[6])
{
    if (as_float) {
        float2 *xf = reinterpret_cast<float2 *>(x) + 3 * node;
        xf[0] = make_float2((float)v[0], (float)v[1]);
        xf[1] = make_float2((float)v[2], (float)v[3]);
        xf[2] = make_float2((float)v[4], (float)v[5]);
    } else {
        double2 *xd = reinterpret_cast<double2 *>(x) + 3 * node;
        xd[0] = make_double2(v[0], v[1]);
        xd[1] = make_double2(v[2], v[3]);
        xd[2] = make_double2(v[4], v[5]);
    }
}
// the 21 words of node n's inverse diagonal block (smoother: from the single-precision copy when the level has one)
__device__ __forceinline__ void node_minv(const DeviceMatrix &m, int sl, int n, bool smoother, double mv[kMinvWords])
{
    if (smoother && m.minv32 != nullptr) {
        const float *mi = m.minv32 + (int64_t)sl * kMinvWords * kSliceNodes + n;
#pragma unroll
        for (int w = 0; w < kMinvWords; w++) mv[w] = (double)mi[w * kSliceNodes];
    } else {
        const double *mi = m.minv + (int64_t)sl * kMinvWords * kSliceNodes + n;
#pragma unroll
        for (int w = 0; w < kMinvWords; w++) mv[w] = mi[w * kSliceNodes];
    }
}
// z = D^-1 r for one node; per row the order of apply_minv (j ascending from zero)
__device__ __forceinline__ void node_minv_apply(const double mv[kMinvWords], const double r[6], double z[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) s += mv[minv_word(i < j ? i : j, i < j ? j : i)] * r[j];
        z[i] = s;
    }
}
// symmetric storage, second phase of a product for the six rows of node n of slice sl: acc += the transposed products of the
// node's in-list in the plan's fixed order (per row the additions of gather_transposed); two entries' loads in flight together
template <bool kT32> __device__ __forceinline__ void node_gather(const DeviceMatrix &m, int sl, int n, double acc[6])
{
    const int Wi = m.in_width[sl];
    const int64_t ib = m.in_base[sl];
    for (int k0 = 0; k0 < Wi; k0 += 2) {
        const int32_t s0 = m.gat_slots[ib + (int64_t)k0 * kSliceNodes + n];
        const int32_t s1 = k0 + 1 < Wi ? m.gat_slots[ib + (int64_t)(k0 + 1) * kSliceNodes + n] : -1;
        double t0[6], t1[6];
        if (s0 >= 0) load_node6(m.tbuf, s0, kT32, t0);
        if (s1 >= 0) load_node6(m.tbuf, s1, kT32, t1);
        if (s0 >= 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) acc[j] += t0[j];
        }
        if (s1 >= 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) acc[j] += t1[j];
        }
    }
}
// (MEASURED, round 5, and dropped: the vectors of k_cheb_step_node moved by the whole wave -- fully coalesced 16-byte accesses of the
//  3 KB two adjacent slices occupy, turned into the node-per-lane view through LDS -- instead of three 16-byte words per lane at a
//  stride of 48 bytes: 0.6393 -> 0.6350 s and 0.6319 -> 0.6308 s per solve of the 4M-triangle panel, alternating on one box.  The
//  partial-line stores the counters show -- 312 MB written where 240 MB are due -- are not what bounds these kernels.)
// the slices a 64-lane workgroup of a node kernel walks: pairs of slices, one per half-wave (the walk of k_spmv_sym)
__host__ __device__ __forceinline__ int node_pairs(int n_slices) { return (n_slices + 1) >> 1; }

} // namespace femshell
