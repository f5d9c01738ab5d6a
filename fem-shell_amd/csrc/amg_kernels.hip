// amg_kernels.hip -- gfx950 vector kernels of the multigrid preconditioner (amg.hpp) and of the flexible PCG
// around it.  All of them are HBM streaming kernels with one lane per scalar row of a 32-node slice, like the
// CG kernels of kernels.hip; the level operators, restrictions and prolongations are multiplied by k_spmv.
#include "amg_kernels.hpp"

#include "device_common.hpp"

namespace femshell {

// ---- Chebyshev smoother ------------------------------------------------------------------------------------

// (kD32: the direction is kept in single precision -- the input of a smoothing product with DeviceMatrix::vec32 == 2)
template <bool kD32>
__global__ __launch_bounds__(192) void k_cheb_start(DeviceMatrix m, const double *__restrict__ rin, double *__restrict__ d,
                                                    double *x, double inv_theta, int accumulate, const CgScalars *gate)
{
    __shared__ double rs[kSliceRows];
    if (gate != nullptr && gate->done != 0) return;
    const int t = threadIdx.x;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv_smoother(m, sl, t);
        const double rv = rin[row];
        const double xv = accumulate ? x[row] : 0.0;
        __syncthreads();
        rs[t] = rv;
        __syncthreads();
        const double dv = inv_theta * apply_minv(mr, t, rs);
        if (kD32) reinterpret_cast<float *>(d)[row] = (float)dv;
        else d[row] = dv;
        x[row] = xv + dv;
    }
}

bool node_kernels();
int node_grid(const DeviceMatrix &m);
template <bool kD32>
__global__ __launch_bounds__(64) void k_cheb_start_node(DeviceMatrix m, const double *__restrict__ rin, double *__restrict__ d, double *x, double inv_theta,
                                  int accumulate, const CgScalars *gate);
template <bool kGather, int kVec>
__global__ __launch_bounds__(64) void k_cheb_step_node(DeviceMatrix m, const double *rin, const double *__restrict__ q, double *rout, double *__restrict__ d,
                                 double *__restrict__ x, double a, double c, const CgScalars *gate);

void launch_cheb_start(const DeviceMatrix &m, const double *rin, double *d, double *x, double inv_theta, bool accumulate,
                       const CgScalars *gate, hipStream_t st, int vec32)
{
    if (node_kernels()) {
        const dim3 g(node_grid(m)), b(64);
        if (vec32 == 2) hipLaunchKernelGGL(k_cheb_start_node<true>, g, b, 0, st, m, rin, d, x, inv_theta, accumulate ? 1 : 0, gate);
        else hipLaunchKernelGGL(k_cheb_start_node<false>, g, b, 0, st, m, rin, d, x, inv_theta, accumulate ? 1 : 0, gate);
        return;
    }
    if (vec32 == 2) hipLaunchKernelGGL(k_cheb_start<true>, dim3(slice_grid(m)), dim3(192), 0, st, m, rin, d, x, inv_theta, accumulate ? 1 : 0, gate);
    else hipLaunchKernelGGL(k_cheb_start<false>, dim3(slice_grid(m)), dim3(192), 0, st, m, rin, d, x, inv_theta, accumulate ? 1 : 0, gate);
}

// (kGather: symmetric storage, q holds the direct part of A d only -- launch_spmv_direct -- and the row adds the
// transposed products of its in-list here, as k_cg_update<true> does for the CG iteration)
// (kVec: what the smoothing product left in single precision, DeviceMatrix::vec32 -- 1: q and the transposed products are
// floats in their buffers, 2: the direction d is kept as floats too; residual and iterate stay FP64)
template <bool kGather, int kVec>
__global__ __launch_bounds__(192) void k_cheb_step(DeviceMatrix m, const double *rin, const double *__restrict__ q,
                                                   double *rout, double *__restrict__ d, double *__restrict__ x, double a,
                                                   double c, const CgScalars *gate)
{
    __shared__ double rs[kSliceRows];
    if (gate != nullptr && gate->done != 0) return;
    const int t = threadIdx.x;
    const float *qf = reinterpret_cast<const float *>(q), *tf = reinterpret_cast<const float *>(m.tbuf);
    float *df = reinterpret_cast<float *>(d);
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv_smoother(m, sl, t);
        double qv = kVec >= 1 ? (double)qf[row] : q[row];
        if (kGather) {
            const int Wi = m.in_width[sl], n = t / 6, j = t % 6;
            const int64_t ib = m.in_base[sl];
            // the slot indices of the first entries together, then their products together (as in k_cg_update): one entry
            // at a time is two dependent memory round trips per entry; the order of the additions is the same
            int32_t slot4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) slot4[k] = (k < Wi) ? m.gat_slots[ib + (int64_t)k * kSliceNodes + n] : -1;
            double t4[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                t4[k] = slot4[k] < 0 ? 0.0 : (kVec >= 1 ? (double)tf[(int64_t)slot4[k] * 6 + j] : m.tbuf[(int64_t)slot4[k] * 6 + j]);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (slot4[k] >= 0) qv += t4[k];
            for (int k = 4; k < Wi; k++) {
                const int32_t slot = m.gat_slots[ib + (int64_t)k * kSliceNodes + n];
                if (slot >= 0) qv += kVec >= 1 ? (double)tf[(int64_t)slot * 6 + j] : m.tbuf[(int64_t)slot * 6 + j];
            }
        }
        const double rn = rin[row] - qv;
        const double dv = kVec == 2 ? (double)df[row] : d[row], xv = x[row];
        rout[row] = rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double dn = a * dv + c * apply_minv(mr, t, rs);
        if (kVec == 2) df[row] = (float)dn;
        else d[row] = dn;
        x[row] = xv + dn;
    }
}

void launch_cheb_step(const DeviceMatrix &m, const double *rin, const double *q, double *rout, double *d, double *x,
                      double a, double c, const CgScalars *gate, hipStream_t st, bool gather, int vec32)
{
    if (node_kernels()) {
        const dim3 g(node_grid(m)), b(64);
        if (gather && vec32 == 2) hipLaunchKernelGGL((k_cheb_step_node<true, 2>), g, b, 0, st, m, rin, q, rout, d, x, a, c, gate);
        else if (gather && vec32 == 1) hipLaunchKernelGGL((k_cheb_step_node<true, 1>), g, b, 0, st, m, rin, q, rout, d, x, a, c, gate);
        else if (gather) hipLaunchKernelGGL((k_cheb_step_node<true, 0>), g, b, 0, st, m, rin, q, rout, d, x, a, c, gate);
        else hipLaunchKernelGGL((k_cheb_step_node<false, 0>), g, b, 0, st, m, rin, q, rout, d, x, a, c, gate);
        return;
    }
    const dim3 grid(slice_grid(m)), block(192);
    if (gather && vec32 == 2) hipLaunchKernelGGL((k_cheb_step<true, 2>), grid, block, 0, st, m, rin, q, rout, d, x, a, c, gate);
    else if (gather && vec32 == 1) hipLaunchKernelGGL((k_cheb_step<true, 1>), grid, block, 0, st, m, rin, q, rout, d, x, a, c, gate);
    else if (gather) hipLaunchKernelGGL((k_cheb_step<true, 0>), grid, block, 0, st, m, rin, q, rout, d, x, a, c, gate);
    else hipLaunchKernelGGL((k_cheb_step<false, 0>), grid, block, 0, st, m, rin, q, rout, d, x, a, c, gate);
}

// ---- one lane per node (device_common.hpp): the smoother's vector kernels without LDS and barriers -------------------------
// FEMSHELL_NODE_KERNELS=0: the one-lane-per-scalar-row kernels of rounds 1-4 (A/B runs)
bool node_kernels()
{
    static const bool on = [] {
        const char *e = getenv("FEMSHELL_NODE_KERNELS");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
// workgroups of a node kernel: 64 lanes = two slices; never more than the per-slice kernels of the same matrix launch
int node_grid(const DeviceMatrix &m)
{
    const int g = 8 * ((node_pairs(m.n_slices) + 7) / 8), cap = slice_grid(m);
    return g < cap ? g : cap;
}

template <bool kD32>
__global__ __launch_bounds__(64) void k_cheb_start_node(DeviceMatrix m, const double *__restrict__ rin, double *__restrict__ d, double *x,
                                                        double inv_theta, int accumulate, const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double mv[kMinvWords], rv[6], xv[6], z[6];
        node_minv(m, sl, n, true, mv);
        load_node6(rin, node, false, rv);
        if (accumulate) load_node6(x, node, false, xv);
        node_minv_apply(mv, rv, z);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double dv = inv_theta * z[j];
            z[j] = dv;
            xv[j] = accumulate ? xv[j] + dv : dv;
        }
        store_node6(d, node, kD32, z);
        store_node6(x, node, false, xv);
    }
}

template <bool kGather, int kVec>
__global__ __launch_bounds__(64) void k_cheb_step_node(DeviceMatrix m, const double *rin, const double *__restrict__ q, double *rout,
                                                       double *__restrict__ d, double *__restrict__ x, double a, double c,
                                                       const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double mv[kMinvWords], qv[6], rv[6], dv[6], xv[6], z[6];
        node_minv(m, sl, n, true, mv);
        load_node6(q, node, kVec >= 1, qv);
        load_node6(rin, node, false, rv);
        load_node6(d, node, kVec == 2, dv);
        load_node6(x, node, false, xv);
        if (kGather) node_gather<(kVec >= 1)>(m, sl, n, qv);
#pragma unroll
        for (int j = 0; j < 6; j++) rv[j] = rv[j] - qv[j];
        store_node6(rout, node, false, rv);
        node_minv_apply(mv, rv, z);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double dn = a * dv[j] + c * z[j];
            dv[j] = dn;
            xv[j] = xv[j] + dn;
        }
        store_node6(d, node, kVec == 2, dv);
        store_node6(x, node, false, xv);
    }
}

// ---- fused passes of the cycle (round 5) ---------------------------------------------------------------------
// The first step of a Chebyshev smoothing -- d = inv_theta D^-1 r, x (+)= d -- is a block-diagonal operation on the residual it
// starts from: it belongs in the epilogue of the kernel that PRODUCES that residual, not in a pass of its own that reads the
// residual back (k_cheb_start: 0.41 GB and 106 us per call on level 0 of the 4M-triangle panel).  Same arithmetic, same bits.

// second phase of a symmetric-storage product (k_sym_gather) + the start of the post-smoothing:
//   out = base_vec + sign (y + transposed products);  d = inv_theta D^-1 out;  x += d
// (kQ32: y and the transposed products are floats; kD32: d is stored as floats -- DeviceMatrix::vec32 of the smoother's matrix m)
template <bool kQ32, bool kD32>
__global__ __launch_bounds__(192) void k_sym_gather_start(DeviceMatrix m, const double *y, double *out, const double *base_vec, double sign,
                                                          double *__restrict__ d, double *__restrict__ x, double inv_theta, const CgScalars *s)
{
    __shared__ double rs[kSliceRows];
    if (s != nullptr && s->done != 0) return;
    const int t = threadIdx.x, n = t / 6, j = t % 6;
    const float *yf = reinterpret_cast<const float *>(y);
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv_smoother(m, sl, t);
        const double bv = base_vec[row], xv = x[row];
        const double acc = gather_transposed<kQ32>(m, sl, n, j, kQ32 ? (double)yf[row] : y[row]);
        const double rn = bv + sign * acc;
        out[row] = rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double dv = inv_theta * apply_minv(mr, t, rs);
        if (kD32) reinterpret_cast<float *>(d)[row] = (float)dv;
        else d[row] = dv;
        x[row] = xv + dv;
    }
}

template <bool kQ32, bool kD32>
__global__ __launch_bounds__(64) void k_sym_gather_start_node(DeviceMatrix m, const double *y, double *out, const double *base_vec, double sign,
                                                              double *__restrict__ d, double *__restrict__ x, double inv_theta,
                                                              const CgScalars *s)
{
    if (s != nullptr && s->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double mv[kMinvWords], acc[6], bv[6], xv[6], z[6];
        node_minv(m, sl, n, true, mv);
        load_node6(y, node, kQ32, acc);
        load_node6(base_vec, node, false, bv);
        load_node6(x, node, false, xv);
        node_gather<kQ32>(m, sl, n, acc);
#pragma unroll
        for (int j = 0; j < 6; j++) bv[j] = bv[j] + sign * acc[j];
        store_node6(out, node, false, bv);
        node_minv_apply(mv, bv, z);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double dv = inv_theta * z[j];
            z[j] = dv;
            xv[j] = xv[j] + dv;
        }
        store_node6(d, node, kD32, z);
        store_node6(x, node, false, xv);
    }
}

void launch_sym_gather_start(const DeviceMatrix &m, const double *y, double *out, const double *base_vec, double sign, double *d, double *x,
                             double inv_theta, bool q32, bool d32, const CgScalars *s, hipStream_t st)
{
    if (node_kernels()) {
        const dim3 g(node_grid(m)), b(64);
        if (q32 && d32) hipLaunchKernelGGL((k_sym_gather_start_node<true, true>), g, b, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
        else if (q32) hipLaunchKernelGGL((k_sym_gather_start_node<true, false>), g, b, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
        else if (d32) hipLaunchKernelGGL((k_sym_gather_start_node<false, true>), g, b, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
        else hipLaunchKernelGGL((k_sym_gather_start_node<false, false>), g, b, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
        return;
    }
    const dim3 grid(slice_grid(m)), block(192);
    if (q32 && d32) hipLaunchKernelGGL((k_sym_gather_start<true, true>), grid, block, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
    else if (q32) hipLaunchKernelGGL((k_sym_gather_start<true, false>), grid, block, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
    else if (d32) hipLaunchKernelGGL((k_sym_gather_start<false, true>), grid, block, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
    else hipLaunchKernelGGL((k_sym_gather_start<false, false>), grid, block, 0, st, m, y, out, base_vec, sign, d, x, inv_theta, s);
}

// the update of the flexible PCG (k_pcg_update: x += alpha p, r -= alpha q, partial sums of r.r) + what stands on both sides
// of it: in front, the second phase of q = K p on symmetric storage (kGather: q arrives as the direct part, the row adds the
// transposed products of its in-list as k_sym_gather does, and stores the whole q for the z.q of k_pcg_dots); behind, the start
// of the cycle's pre-smoothing on the new residual: d = inv_theta D^-1 r, z = d (m: level 0 as the smoother sees it)
template <bool kGather, bool kD32>
__global__ __launch_bounds__(192) void k_pcg_update_start(DeviceMatrix m, CgVectors v, double *__restrict__ d, double *__restrict__ z,
                                                          double inv_theta)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    if (v.s->done != 0) return;
    const int t = threadIdx.x, n = t / 6, j = t % 6;
    const double alpha = v.s->alpha;
    double d1 = 0.0, d2 = 0.0; // r.r and x.x (the refinement pass's stopping rule: CG_PHASE_FLEX_CONV)
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv_smoother(m, sl, t);
        const double pv = v.p[row], xv = v.x[row], rv = v.r[row];
        double qv = v.q[row];
        if (kGather) {
            qv = gather_transposed<false>(m, sl, n, j, qv);
            v.q[row] = qv;
        }
        const double xn = xv + alpha * pv;
        v.x[row] = xn;
        d2 += xn * xn;
        const double rn = rv - alpha * qv;
        v.r[row] = rn;
        d1 += rn * rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double dv = inv_theta * apply_minv(mr, t, rs);
        if (kD32) reinterpret_cast<float *>(d)[row] = (float)dv;
        else d[row] = dv;
        z[row] = dv;
    }
    const double t1 = block_sum(d1, sh);
    __syncthreads();
    const double t2 = block_sum(d2, sh);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t1;
        v.partials[gridDim.x + blockIdx.x] = t2;
    }
}

// (launched with the grid of the per-slice kernels, slice_grid(m): the scalar step reduces that many partial sums)
template <bool kGather, bool kD32>
__global__ __launch_bounds__(64) void k_pcg_update_start_node(DeviceMatrix m, CgVectors v, double *__restrict__ d, double *__restrict__ z,
                                                              double inv_theta)
{
    if (v.s->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    const double alpha = v.s->alpha;
    double d1 = 0.0, d2 = 0.0;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double mv[kMinvWords], pv[6], xv[6], rv[6], qv[6], zz[6];
        node_minv(m, sl, n, true, mv);
        load_node6(v.p, node, false, pv);
        load_node6(v.x, node, false, xv);
        load_node6(v.r, node, false, rv);
        load_node6(v.q, node, false, qv);
        if (kGather) {
            node_gather<false>(m, sl, n, qv);
            store_node6(v.q, node, false, qv);
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            xv[j] = xv[j] + alpha * pv[j];
            d2 += xv[j] * xv[j];
            const double rn = rv[j] - alpha * qv[j];
            rv[j] = rn;
            d1 += rn * rn;
        }
        store_node6(v.x, node, false, xv);
        store_node6(v.r, node, false, rv);
        node_minv_apply(mv, rv, zz);
#pragma unroll
        for (int j = 0; j < 6; j++) zz[j] = inv_theta * zz[j];
        store_node6(d, node, kD32, zz);
        store_node6(z, node, false, zz);
    }
    const double t1 = wave_sum(d1), t2 = wave_sum(d2);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t1;
        v.partials[gridDim.x + blockIdx.x] = t2;
    }
}

void launch_pcg_update_start(const DeviceMatrix &m, const CgVectors &v, double *d, double *z, double inv_theta, bool gather, bool d32,
                             hipStream_t st)
{
    if (node_kernels()) {
        const dim3 g(slice_grid(m)), b(64);
        if (gather && d32) hipLaunchKernelGGL((k_pcg_update_start_node<true, true>), g, b, 0, st, m, v, d, z, inv_theta);
        else if (gather) hipLaunchKernelGGL((k_pcg_update_start_node<true, false>), g, b, 0, st, m, v, d, z, inv_theta);
        else if (d32) hipLaunchKernelGGL((k_pcg_update_start_node<false, true>), g, b, 0, st, m, v, d, z, inv_theta);
        else hipLaunchKernelGGL((k_pcg_update_start_node<false, false>), g, b, 0, st, m, v, d, z, inv_theta);
        return;
    }
    const dim3 grid(slice_grid(m)), block(192);
    if (gather && d32) hipLaunchKernelGGL((k_pcg_update_start<true, true>), grid, block, 0, st, m, v, d, z, inv_theta);
    else if (gather) hipLaunchKernelGGL((k_pcg_update_start<true, false>), grid, block, 0, st, m, v, d, z, inv_theta);
    else if (d32) hipLaunchKernelGGL((k_pcg_update_start<false, true>), grid, block, 0, st, m, v, d, z, inv_theta);
    else hipLaunchKernelGGL((k_pcg_update_start<false, false>), grid, block, 0, st, m, v, d, z, inv_theta);
}

// ---- power iteration ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(192) void k_minv_apply_norm(DeviceMatrix m, const double *__restrict__ q, double *__restrict__ z,
                                                         double *__restrict__ partials)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    const int t = threadIdx.x;
    double acc = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv(m, sl, t);
        const double qv = q[row];
        __syncthreads();
        rs[t] = qv;
        __syncthreads();
        const double zv = apply_minv(mr, t, rs);
        z[row] = zv;
        acc += zv * zv;
    }
    const double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

void launch_minv_apply_norm(const DeviceMatrix &m, const double *q, double *z, double *partials, hipStream_t st)
{
    hipLaunchKernelGGL(k_minv_apply_norm, dim3(slice_grid(m)), dim3(192), 0, st, m, q, z, partials);
}

__global__ __launch_bounds__(256) void k_fill_hash(double *x, int64_t n_real, int64_t n_total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0); // [0,1)
        x[i] = i < n_real ? 2.0 * u - 1.0 : 0.0;
    }
}

void launch_fill_hash(double *x, int64_t n_real, int64_t n_total, hipStream_t st)
{
    const int64_t blocks = (n_total + 255) / 256;
    hipLaunchKernelGGL(k_fill_hash, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, n_real, n_total);
}

// ---- coarsest level ----------------------------------------------------------------------------------------

// one wave per row of the dense inverse (rows are read with consecutive lanes on consecutive words)
__global__ __launch_bounds__(256) void k_dense_gemv(const double *__restrict__ A, const double *__restrict__ b,
                                                    double *__restrict__ y, int n, int n_pad6, const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad6) return;
    double acc = 0.0;
    if (row < n) {
        const double *a = A + (int64_t)row * n;
        for (int j = lane; j < n; j += 64) acc += a[j] * b[j];
        acc = wave_sum(acc);
    }
    if (lane == 0) y[row] = acc;
}

void launch_dense_gemv(const double *Ainv, const double *b, double *y, int32_t n, int32_t n_pad6, const CgScalars *gate,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_dense_gemv, dim3((n_pad6 + 3) / 4), dim3(256), 0, st, Ainv, b, y, n, n_pad6, gate);
}

// ---- flexible PCG ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(192) void k_pcg_init(DeviceMatrix m, CgVectors v)
{
    __shared__ double sh[3];
    const int t = threadIdx.x;
    double d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int64_t row = (int64_t)w.s * kSliceRows + t;
        const double bv = v.b[row];
        v.x[row] = 0.0;
        v.r[row] = bv;
        d1 += bv * bv;
    }
    const double t1 = block_sum(d1, sh);
    if (threadIdx.x == 0) v.partials[blockIdx.x] = t1;
}

void launch_pcg_init(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_pcg_init, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

__global__ __launch_bounds__(192) void k_pcg_update(DeviceMatrix m, CgVectors v)
{
    __shared__ double sh[3];
    if (v.s->done != 0) return;
    const int t = threadIdx.x;
    const double alpha = v.s->alpha;
    double d1 = 0.0, d2 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int64_t row = (int64_t)w.s * kSliceRows + t;
        const double pv = v.p[row], qv = v.q[row], xv = v.x[row], rv = v.r[row];
        const double xn = xv + alpha * pv;
        v.x[row] = xn;
        d2 += xn * xn;
        const double rn = rv - alpha * qv;
        v.r[row] = rn;
        d1 += rn * rn;
    }
    const double t1 = block_sum(d1, sh);
    __syncthreads();
    const double t2 = block_sum(d2, sh);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t1;
        v.partials[gridDim.x + blockIdx.x] = t2;
    }
}

void launch_pcg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_pcg_update, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

__global__ __launch_bounds__(192) void k_pcg_norm(DeviceMatrix m, CgVectors v)
{
    __shared__ double sh[3];
    const int t = threadIdx.x;
    double d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const double rv = v.r[(int64_t)w.s * kSliceRows + t];
        d1 += rv * rv;
    }
    const double t1 = block_sum(d1, sh);
    if (threadIdx.x == 0) v.partials[blockIdx.x] = t1;
}

void launch_pcg_norm(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_pcg_norm, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

__global__ __launch_bounds__(192) void k_pcg_dots(DeviceMatrix m, CgVectors v)
{
    __shared__ double sh[3];
    if (v.s->done != 0) return;
    const int G = gridDim.x, t = threadIdx.x;
    double d0 = 0.0, d2 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int64_t row = (int64_t)w.s * kSliceRows + t;
        const double zv = v.z[row];
        d0 += v.r[row] * zv;
        d2 += zv * v.q[row];
    }
    const double t0 = block_sum(d0, sh);
    const double t2 = block_sum(d2, sh);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t0;
        v.partials[G + blockIdx.x] = t2;
    }
}

void launch_pcg_dots(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_pcg_dots, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

__global__ __launch_bounds__(256) void k_copy_gated(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n2,
                                                    const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

void launch_copy(const double *src, double *dst, int64_t n, const CgScalars *gate, hipStream_t st)
{
    const int64_t n2 = n / 2, blocks = (n2 + 255) / 256; // vector lengths are multiples of 192
    hipLaunchKernelGGL(k_copy_gated, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                       reinterpret_cast<const double2 *>(src), reinterpret_cast<double2 *>(dst), n2, gate);
}

__global__ __launch_bounds__(256) void k_add(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n2)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 a = src[i];
        double2 d = dst[i];
        d.x += a.x;
        d.y += a.y;
        dst[i] = d;
    }
}

void launch_add(const double *src, double *dst, int64_t n, hipStream_t st)
{
    const int64_t n2 = n / 2, blocks = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_add, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                       reinterpret_cast<const double2 *>(src), reinterpret_cast<double2 *>(dst), n2);
}

__global__ __launch_bounds__(256) void k_sub(const double2 *__restrict__ a, const double2 *__restrict__ b, double2 *__restrict__ out, int64_t n2)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 x = a[i], y = b[i];
        out[i] = make_double2(x.x - y.x, x.y - y.y);
    }
}

void launch_sub(const double *a, const double *b, double *out, int64_t n, hipStream_t st)
{
    const int64_t n2 = n / 2, blocks = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_sub, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                       reinterpret_cast<const double2 *>(a), reinterpret_cast<const double2 *>(b), reinterpret_cast<double2 *>(out), n2);
}

// ---- K cycle -----------------------------------------------------------------------------------------------

constexpr int kKcycGroups = 128; // stage-1 workgroups of a K-cycle dot product

// (ticket != nullptr -- row-partitioned levels: the workgroup that finishes LAST adds the partial sums up, in the order and with
//  the adder k_kcyc_finish uses (the same bits), and leaves the rank's three sums in sums_out for the all-reduce: the
//  one-workgroup launch between this kernel and the collective is gone, 4.3 us of every coefficient step)
__global__ __launch_bounds__(256) void k_kcyc_dots(const double *__restrict__ a0, const double *__restrict__ b0,
                                                   const double *__restrict__ a1, const double *__restrict__ b1,
                                                   const double *__restrict__ a2, const double *__restrict__ b2, int64_t n,
                                                   double *scratch, const CgScalars *gate, unsigned int *ticket = nullptr,
                                                   double *sums_out = nullptr, int phase = 0)
{
    __shared__ double sh[4];
    __shared__ int last_one;
    if (gate != nullptr && gate->done != 0) return;
    // contiguous chunk per workgroup, fixed order: deterministic
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        s0 += a0[i] * b0[i];
        s1 += a1[i] * b1[i];
        if (a2 != nullptr) s2 += a2[i] * b2[i];
    }
    const double t0 = block_sum(s0, sh);
    const double t1 = block_sum(s1, sh);
    const double t2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = t0;
        scratch[kKcycGroups + blockIdx.x] = t1;
        if (a2 != nullptr) scratch[2 * kKcycGroups + blockIdx.x] = t2; // (two-product callers hand over 2 x kKcycGroups doubles)
    }
    if (ticket == nullptr) return;
    if (threadIdx.x == 0) {
        __threadfence(); // the partial sums above before the ticket
        last_one = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last_one) return;
    __threadfence(); // (acquire for every thread's loads of the others' partial sums)
    double s[3];
    for (int a = 0; a < 3; a++) { // k_kcyc_finish's sums: lanes [0, groups) of a block_sum, the lanes beyond add zeros
        const double part = ((int)threadIdx.x < (int)gridDim.x && (a < 2 || a2 != nullptr))
                                ? __hip_atomic_load(&scratch[a * kKcycGroups + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0.0;
        s[a] = block_sum(part, sh);
    }
    if (threadIdx.x == 0) {
        sums_out[0] = s[0];
        sums_out[1] = s[1];
        sums_out[2] = phase == 1 ? 0.0 : s[2];
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (for the next coefficient step)
    }
}

// (sums_out != nullptr: the three sums go there and no coefficient step is taken -- row-partitioned levels all-reduce them
//  first and finish with groups = 1, stride = 1 on the reduced words)
__global__ __launch_bounds__(128) void k_kcyc_finish(int phase, int groups, int stride, const double *__restrict__ scratch,
                                                     KcycScalars *ks, const CgScalars *gate, double *sums_out)
{
    __shared__ double sh[4];
    if (gate != nullptr && gate->done != 0) return;
    double s[3];
    for (int a = 0; a < 3; a++) {
        const double part = (int)threadIdx.x < groups ? scratch[a * stride + threadIdx.x] : 0.0;
        s[a] = block_sum(part, sh);
    }
    if (threadIdx.x != 0) return;
    if (sums_out != nullptr) {
        sums_out[0] = s[0];
        sums_out[1] = s[1];
        sums_out[2] = phase == 1 ? 0.0 : s[2];
        return;
    }
    if (phase == 1) {
        const double rho1 = s[0], a1 = s[1];
        ks->rho1 = rho1;
        ks->a1 = a1;
        ks->t = rho1 > 0.0 ? a1 / rho1 : 0.0;
    } else {
        const double g = s[0], b2 = s[1], a2 = s[2];
        const double rho1 = ks->rho1, a1 = ks->a1;
        double w1 = ks->t, w2 = 0.0;
        if (rho1 > 0.0) {
            const double rho2 = b2 - g * g / rho1;
            if (rho2 > 0.0) {
                w1 = a1 / rho1 - g * a2 / (rho1 * rho2);
                w2 = a2 / rho2;
            }
        }
        ks->w1 = w1;
        ks->w2 = w2;
    }
}

void launch_kcyc_dots(int phase, const double *a0, const double *b0, const double *a1, const double *b1, const double *a2,
                      const double *b2, int64_t n6, KcycScalars *ks, double *scratch, const CgScalars *gate, hipStream_t st)
{
    int groups = (int)((n6 + 4095) / 4096);
    if (groups > kKcycGroups) groups = kKcycGroups;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(k_kcyc_dots, dim3(groups), dim3(256), 0, st, a0, b0, a1, b1, a2, b2, n6, scratch, gate);
    hipLaunchKernelGGL(k_kcyc_finish, dim3(1), dim3(128), 0, st, phase, groups, kKcycGroups, scratch, ks, gate, (double *)nullptr);
}

void launch_kcyc_dots_local(int phase, const double *a0, const double *b0, const double *a1, const double *b1, const double *a2,
                            const double *b2, int64_t n6, double *scratch, double *sums, const CgScalars *gate, hipStream_t st)
{
    int groups = (int)((n6 + 4095) / 4096);
    if (groups > kKcycGroups) groups = kKcycGroups;
    if (groups < 1) groups = 1;
    // (the ticket of the last-workgroup reduction: the word behind the three sums and a spare -- sums[4], zero since the level's
    //  vectors were allocated, set back to zero by the workgroup that draws the last ticket)
    hipLaunchKernelGGL(k_kcyc_dots, dim3(groups), dim3(256), 0, st, a0, b0, a1, b1, a2, b2, n6, scratch, gate,
                       reinterpret_cast<unsigned int *>(sums + 4), sums, phase);
}

void launch_kcyc_coefficients(int phase, const double *sums, KcycScalars *ks, const CgScalars *gate, hipStream_t st)
{
    hipLaunchKernelGGL(k_kcyc_finish, dim3(1), dim3(128), 0, st, phase, 1, 1, sums, ks, gate, (double *)nullptr);
}

// (sums != nullptr -- row-partitioned levels: the all-reduced sums of the step; every thread forms the coefficient from them as
//  k_kcyc_finish does, the first one keeps it in *ks for the second step: no launch of its own for three divisions)
__global__ __launch_bounds__(256) void k_kcyc_r2(const double *__restrict__ rc, const double *__restrict__ v1,
                                                 double *__restrict__ r2, int64_t n, KcycScalars *ks, const CgScalars *gate,
                                                 const double *sums = nullptr)
{
    if (gate != nullptr && gate->done != 0) return;
    double t;
    if (sums != nullptr) {
        const double rho1 = sums[0], a1 = sums[1];
        t = rho1 > 0.0 ? a1 / rho1 : 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ks->rho1 = rho1;
            ks->a1 = a1;
            ks->t = t;
        }
    } else {
        t = ks->t;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        r2[i] = rc[i] - t * v1[i];
}

// Small levels (a few thousand rows): the three launches of a K-cycle coefficient step -- stage-1 dots, finish, vector
// update -- as one single-workgroup kernel; each of them costs the 4.5 us of a dependent launch, whatever its size.
__global__ __launch_bounds__(1024) void k_kcyc_step1_small(const double *__restrict__ c1, const double *__restrict__ v1,
                                                            const double *__restrict__ rc, double *__restrict__ r2, int n,
                                                            KcycScalars *ks, const CgScalars *gate)
{
    __shared__ double sh[16];
    __shared__ double tt;
    if (gate != nullptr && gate->done != 0) return;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double c = c1[i];
        s0 += c * v1[i];
        s1 += c * rc[i];
    }
    const double rho1 = block_sum(s0, sh), a1 = block_sum(s1, sh);
    if (threadIdx.x == 0) {
        const double t = rho1 > 0.0 ? a1 / rho1 : 0.0;
        ks->rho1 = rho1;
        ks->a1 = a1;
        ks->t = t;
        tt = t;
    }
    __syncthreads();
    const double t = tt;
    for (int i = threadIdx.x; i < n; i += blockDim.x) r2[i] = rc[i] - t * v1[i];
}

__global__ __launch_bounds__(1024) void k_kcyc_step2_small(const double *__restrict__ c1, const double *__restrict__ c2,
                                                            const double *__restrict__ v1, const double *__restrict__ v2,
                                                            const double *__restrict__ r2, double *__restrict__ x, int n,
                                                            KcycScalars *ks, const CgScalars *gate)
{
    __shared__ double sh[16];
    __shared__ double ww[2];
    if (gate != nullptr && gate->done != 0) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double c = c2[i];
        s0 += c * v1[i];
        s1 += c * v2[i];
        s2 += c * r2[i];
    }
    const double g = block_sum(s0, sh), b2 = block_sum(s1, sh), a2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        const double rho1 = ks->rho1, a1 = ks->a1;
        double w1 = ks->t, w2 = 0.0;
        if (rho1 > 0.0) {
            const double rho2 = b2 - g * g / rho1;
            if (rho2 > 0.0) {
                w1 = a1 / rho1 - g * a2 / (rho1 * rho2);
                w2 = a2 / rho2;
            }
        }
        ks->w1 = w1;
        ks->w2 = w2;
        ww[0] = w1;
        ww[1] = w2;
    }
    __syncthreads();
    const double w1 = ww[0], w2 = ww[1];
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = w1 * c1[i] + w2 * c2[i];
}

void launch_kcyc_step1_small(const double *c1, const double *v1, const double *rc, double *r2, int64_t n6, KcycScalars *ks,
                             const CgScalars *gate, hipStream_t st)
{
    hipLaunchKernelGGL(k_kcyc_step1_small, dim3(1), dim3(1024), 0, st, c1, v1, rc, r2, (int)n6, ks, gate);
}

void launch_kcyc_step2_small(const double *c1, const double *c2, const double *v1, const double *v2, const double *r2, double *x,
                             int64_t n6, KcycScalars *ks, const CgScalars *gate, hipStream_t st)
{
    hipLaunchKernelGGL(k_kcyc_step2_small, dim3(1), dim3(1024), 0, st, c1, c2, v1, v2, r2, x, (int)n6, ks, gate);
}

int launch_two_dots(const double *a0, const double *b0, const double *a1, const double *b1, int64_t n, double *scratch, hipStream_t st)
{
    int groups = (int)((n + 4095) / 4096);
    if (groups > kKcycGroups) groups = kKcycGroups;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(k_kcyc_dots, dim3(groups), dim3(256), 0, st, a0, b0, a1, b1, (const double *)nullptr, (const double *)nullptr, n,
                       scratch, (const CgScalars *)nullptr);
    return groups;
}

void launch_kcyc_r2(const double *rc, const double *v1, double *r2, int64_t n6, KcycScalars *ks, const CgScalars *gate,
                    hipStream_t st, const double *sums)
{
    const int64_t blocks = (n6 + 255) / 256;
    hipLaunchKernelGGL(k_kcyc_r2, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, rc, v1, r2, n6, ks, gate, sums);
}

__global__ __launch_bounds__(256) void k_kcyc_combine(const double *__restrict__ c1, const double *__restrict__ c2,
                                                      double *__restrict__ x, int64_t n, KcycScalars *ks,
                                                      const CgScalars *gate, const double *sums = nullptr)
{
    if (gate != nullptr && gate->done != 0) return;
    double w1, w2;
    if (sums != nullptr) { // (see k_kcyc_r2; the expressions of k_kcyc_finish, phase 2)
        const double g = sums[0], b2 = sums[1], a2 = sums[2];
        const double rho1 = ks->rho1, a1 = ks->a1;
        w1 = ks->t;
        w2 = 0.0;
        if (rho1 > 0.0) {
            const double rho2 = b2 - g * g / rho1;
            if (rho2 > 0.0) {
                w1 = a1 / rho1 - g * a2 / (rho1 * rho2);
                w2 = a2 / rho2;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ks->w1 = w1;
            ks->w2 = w2;
        }
    } else {
        w1 = ks->w1;
        w2 = ks->w2;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = w1 * c1[i] + w2 * c2[i];
}

void launch_kcyc_combine(const double *c1, const double *c2, double *x, int64_t n6, KcycScalars *ks,
                         const CgScalars *gate, hipStream_t st, const double *sums)
{
    const int64_t blocks = (n6 + 255) / 256;
    hipLaunchKernelGGL(k_kcyc_combine, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, c1, c2, x, n6, ks, gate, sums);
}

} // namespace femshell

// =====================================================================================================================
// Multigrid setup on the device (amg_kernels.hpp): one lane per block of the result, gathers over the plan's lists
// =====================================================================================================================
namespace femshell {

namespace {

// element (i, j) of the block in slot `slot` of a sliced block ELL value array (slice bases are multiples of 32)
__device__ __forceinline__ int64_t ell_index(int64_t slot, int i, int j)
{
    const int n = (int)(slot & 31);
    return (slot - n) * 36 + ((int64_t)((j >> 1) * 6 + i) * kSliceNodes + n) * 2 + (j & 1);
}
// (upper_only: a diagonal block of K that holds its upper triangle only -- DeviceMatrix::diag_upper)
__device__ __forceinline__ void ell_load_sym(const double *vals, int64_t slot, double b[36])
{
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) {
            const double v = vals[ell_index(slot, i, j)];
            b[6 * i + j] = v;
            b[6 * j + i] = v;
        }
}
__device__ __forceinline__ void ell_load(const double *vals, int64_t slot, double b[36], bool transposed)
{
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double v = vals[ell_index(slot, i, j)];
            if (transposed) b[6 * j + i] = v;
            else b[6 * i + j] = v;
        }
}
__device__ __forceinline__ void ell_store(double *vals, int64_t slot, const double b[36])
{
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) vals[ell_index(slot, i, j)] = b[6 * i + j];
}
// A P, the intermediate of a coarsening step that only the Galerkin product reads, keeps every block as 36 consecutive
// doubles (slot-major): a block is then 288 contiguous bytes instead of 18 words in 18 different 512-byte groups of the
// sliced layout, which is what the product's gather of whole blocks wants (8.7 -> see DESIGN.md section 5)
__device__ __forceinline__ void blk_load_contig(const double *vals, int64_t slot, double b[36])
{
    const double2 *src = reinterpret_cast<const double2 *>(vals + slot * 36);
#pragma unroll
    for (int w = 0; w < 18; w++) {
        const double2 v = src[w];
        b[2 * w] = v.x;
        b[2 * w + 1] = v.y;
    }
}
__device__ __forceinline__ void blk_store_contig(double *vals, int64_t slot, const double b[36])
{
    double2 *dst = reinterpret_cast<double2 *>(vals + slot * 36);
#pragma unroll
    for (int w = 0; w < 18; w++) dst[w] = make_double2(b[2 * w], b[2 * w + 1]);
}
// c += a * b (row-major 6x6), a^T * b with ta
__device__ __forceinline__ void blk_mac(const double a[36], const double b[36], double c[36], bool ta)
{
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double aik = ta ? a[6 * k + i] : a[6 * i + k];
#pragma unroll
            for (int j = 0; j < 6; j++) c[6 * i + j] += aik * b[6 * k + j];
        }
}
// slot k of row (sl, n) of an ELL matrix
__device__ __forceinline__ int64_t ell_slot(const EllView &M, int sl, int k, int n) { return M.slice_base[sl] + (int64_t)k * kSliceNodes + n; }

// the neighbours of row a of K in a fixed order: its own slots (padding skipped), then its in-list (symmetric storage)
template <class F> __device__ __forceinline__ void for_each_neighbour(const DeviceMatrix &A, int a, F f)
{
    const int sl = a / kSliceNodes, n = a % kSliceNodes;
    const int64_t base = A.slice_base[sl];
    const int W = A.slice_width[sl];
    for (int k = 0; k < W; k++) {
        const int64_t slot = base + (int64_t)k * kSliceNodes + n;
        const int c = (k == 0) ? a : A.cols[slot];
        if (k > 0 && c == a) continue; // padding slot
        f(c, slot, false, slot);
    }
    if (A.symmetric) {
        const int Wi = A.in_width[sl];
        const int64_t ib = A.in_base[sl];
        for (int k = 0; k < Wi; k++) {
            const int64_t e = ib + (int64_t)k * kSliceNodes + n;
            const int32_t slot = A.in_slots[e];
            if (slot >= 0) f(A.in_rows[e], (int64_t)slot, true, e);
        }
    }
}

} // namespace

__device__ __forceinline__ int ell_slice_of(const EllView &M, int64_t t)
{
    int lo = 0, hi = M.n_slices - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (M.slice_base[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// slot of column J in row r of M, -1 if absent
__device__ __forceinline__ int64_t ell_find(const EllView &M, int r, int J)
{
    const int sl = r / kSliceNodes, n = r % kSliceNodes;
    const int cnt = M.count[r];
    for (int k = 0; k < cnt; k++) {
        const int64_t slot = M.slice_base[sl] + (int64_t)k * kSliceNodes + n;
        if (M.cols[slot] == J) return slot;
    }
    return -1;
}

__global__ __launch_bounds__(128) void k_amg_prolongator(DeviceMatrix A, const int32_t *__restrict__ agg, const double *__restrict__ Q,
                                                         double omega, const uint8_t *__restrict__ pmap_own,
                                                         const uint8_t *__restrict__ pmap_in, EllView P)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.total) return;
    // thread t <-> slot t of P
    const int sl = ell_slice_of(P, t); // slices have different widths: bisection over slice_base
    const int64_t off = t - P.slice_base[sl];
    const int kP = (int)(off / kSliceNodes), n = (int)(off % kSliceNodes);
    const int a = sl * kSliceNodes + n;
    double out[36];
#pragma unroll
    for (int e = 0; e < 36; e++) out[e] = 0.0;
    if (a < P.n_rows && kP < P.count[a]) {
        const int J = P.cols[t];
        double acc[36];
#pragma unroll
        for (int e = 0; e < 36; e++) acc[e] = 0.0;
        for_each_neighbour(A, a, [&](int c, int64_t slot, bool transposed, int64_t map_index) {
            const uint8_t target = transposed ? pmap_in[map_index] : pmap_own[map_index];
            if (target != kP) return;
            double blk[36], q[36];
            if (A.diag_upper && c == a && !transposed) ell_load_sym(A.vals, slot, blk);
            else ell_load(A.vals, slot, blk, transposed);
#pragma unroll
            for (int e = 0; e < 36; e++) q[e] = Q[(int64_t)c * 36 + e];
            blk_mac(blk, q, acc, false);
        });
        // D^-1 from the packed upper triangle of the inverse diagonal block
        double dinv[36];
        const double *mi = A.minv + (int64_t)(a / kSliceNodes) * kMinvWords * kSliceNodes + (a % kSliceNodes);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) dinv[6 * i + j] = mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
        double sm[36];
#pragma unroll
        for (int e = 0; e < 36; e++) sm[e] = 0.0;
        blk_mac(dinv, acc, sm, false);
        const bool own = J == agg[a];
#pragma unroll
        for (int e = 0; e < 36; e++) out[e] = (own ? Q[(int64_t)a * 36 + e] : 0.0) - omega * sm[e];
    }
    ell_store(P.vals, t, out);
}

__global__ __launch_bounds__(128) void k_amg_ap(DeviceMatrix A, EllView P, EllView AP)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= AP.total) return;
    const int sl = ell_slice_of(AP, t);
    const int64_t off = t - AP.slice_base[sl];
    const int kA = (int)(off / kSliceNodes), n = (int)(off % kSliceNodes);
    const int a = sl * kSliceNodes + n;
    double acc[36];
#pragma unroll
    for (int e = 0; e < 36; e++) acc[e] = 0.0;
    if (a < AP.n_rows && kA < AP.count[a]) {
        const int J = AP.cols[t];
        for_each_neighbour(A, a, [&](int c, int64_t slot, bool transposed, int64_t) {
            const int64_t ps = ell_find(P, c, J);
            if (ps < 0) return;
            double blk[36], pb[36];
            if (A.diag_upper && c == a && !transposed) ell_load_sym(A.vals, slot, blk);
            else ell_load(A.vals, slot, blk, transposed);
            ell_load(P.vals, ps, pb, false);
            blk_mac(blk, pb, acc, false);
        });
    }
    blk_store_contig(AP.vals, t, acc);
}

__global__ __launch_bounds__(128) void k_amg_restriction(EllView P, const int64_t *__restrict__ rptr, const int32_t *__restrict__ rrow,
                                                         const uint8_t *__restrict__ rk, EllView R)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.total) return;
    const int sl = ell_slice_of(R, t);
    const int64_t off = t - R.slice_base[sl];
    const int kR = (int)(off / kSliceNodes), n = (int)(off % kSliceNodes);
    const int I = sl * kSliceNodes + n;
    double out[36];
#pragma unroll
    for (int e = 0; e < 36; e++) out[e] = 0.0;
    if (I < R.n_rows && kR < (int)(rptr[I + 1] - rptr[I])) {
        const int64_t q = rptr[I] + kR;
        const int i = rrow[q];
        ell_load(P.vals, ell_slot(P, i / kSliceNodes, rk[q], i % kSliceNodes), out, true);
    }
    ell_store(R.vals, t, out);
}

// (diag_key: the column key of row I's diagonal block is I + diag_key -- row-partitioned levels carry global coarse ids as
//  column keys through the setup, amg_dist.cpp)
__global__ __launch_bounds__(128) void k_amg_galerkin(EllView P, EllView AP, const int64_t *__restrict__ rptr,
                                                      const int32_t *__restrict__ rrow, const uint8_t *__restrict__ rk, EllView Ac,
                                                      int diag_key)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Ac.total) return;
    const int sl = ell_slice_of(Ac, t);
    const int64_t off = t - Ac.slice_base[sl];
    const int kC = (int)(off / kSliceNodes), n = (int)(off % kSliceNodes);
    const int I = sl * kSliceNodes + n;
    double acc[36];
#pragma unroll
    for (int e = 0; e < 36; e++) acc[e] = 0.0;
    if (I < Ac.n_rows && kC < Ac.count[I]) {
        const int J = Ac.cols[t];
        for (int64_t q = rptr[I]; q < rptr[I + 1]; q++) {
            const int i = rrow[q];
            const int64_t as = ell_find(AP, i, J);
            if (as < 0) continue;
            double pb[36], ab[36];
            ell_load(P.vals, ell_slot(P, i / kSliceNodes, rk[q], i % kSliceNodes), pb, false);
            blk_load_contig(AP.vals, as, ab);
            blk_mac(pb, ab, acc, true); // P_iI^T (A P)_iJ
        }
        if (J == I + diag_key) // coarse dofs without fine support (zero column of P): unit diagonal keeps the level matrix SPD
#pragma unroll
            for (int v = 0; v < 6; v++)
                if (acc[7 * v] == 0.0) acc[7 * v] = 1.0;
    }
    ell_store(Ac.vals, t, acc);
}

// The Galerkin product on the matrix cores.  For a coarse row I the contraction sum_i P_iI^T (A P)_iJ over the ~20 fine
// rows i that see aggregate I and the ~12 coarse columns J of the row is a dense tall-skinny product: X^T (6 x 6k) times Y
// (6k x 6m), Y holding the blocks (A P)_iJ and zeros where a fine row does not reach a column.  One wave per coarse row
// runs it as v_mfma_f64_16x16x4_f64 tiles: M = the 6 coarse dofs of I (10 of the 16 rows idle), K = the 6 dofs of fine
// row i in two steps of 4, N = 16 columns of the panel per tile.  Operand maps (cdna_hip_programming.md, f64 form):
// lane l holds A[row l&15][k l>>4] and B[k l>>4][col l&15]; result reg g of lane l is D[row (l>>4) + 4g][col l&15].
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kGalerkinTiles = 6; // 96 panel columns = 16 coarse columns per pass

__global__ __launch_bounds__(64) void k_amg_galerkin_mfma(EllView P, EllView AP, const int64_t *__restrict__ rptr,
                                                          const int32_t *__restrict__ rrow, const uint8_t *__restrict__ rk, EllView Ac)
{
    __shared__ int32_t jcol[256];
    __shared__ int64_t apslot[16];
    const int I = blockIdx.x, lane = threadIdx.x;
    const int sl = I / kSliceNodes, n = I % kSliceNodes;
    const int W = Ac.slice_width[sl];
    const int cnt = I < Ac.n_rows ? Ac.count[I] : 0;
    const int64_t base = Ac.slice_base[sl];
    for (int k = lane; k < cnt; k += 64) jcol[k] = Ac.cols[base + (int64_t)k * kSliceNodes + n];
    // padding slots of the row hold zero blocks
    for (int e = lane; e < (W - cnt) * 36; e += 64) {
        const int k = cnt + e / 36, ij = e % 36;
        Ac.vals[ell_index(base + (int64_t)k * kSliceNodes + n, ij / 6, ij % 6)] = 0.0;
    }
    __syncthreads();
    const int r = lane & 15, kq = lane >> 4;
    for (int g0 = 0; g0 < cnt; g0 += 16) { // 16 coarse columns = 96 panel columns per pass
        const int gcnt = min(16, cnt - g0);
        const int ntiles = (6 * gcnt + 15) / 16;
        v4d acc[kGalerkinTiles];
#pragma unroll
        for (int t = 0; t < kGalerkinTiles; t++) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int64_t q = rptr[I]; q < rptr[I + 1]; q++) {
            const int i = rrow[q];
            const int64_t pslot = ell_slot(P, i / kSliceNodes, rk[q], i % kSliceNodes);
            __syncthreads(); // the previous fine row's readers are done with apslot
            if (lane < gcnt) apslot[lane] = ell_find(AP, i, jcol[g0 + lane]);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int d = 4 * ks + kq; // dof of the fine row (k index of this step)
                const double a = (r < 6 && d < 6) ? P.vals[ell_index(pslot, d, r)] : 0.0; // X^T[r][d] = P_iI[d][r]
#pragma unroll
                for (int t = 0; t < kGalerkinTiles; t++) {
                    if (t < ntiles) {
                        const int colg = 16 * t + r, sj = colg / 6, cdof = colg % 6;
                        double b = 0.0;
                        if (d < 6 && sj < gcnt) {
                            const int64_t as = apslot[sj];
                            if (as >= 0) b = AP.vals[as * 36 + 6 * d + cdof];
                        }
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        // rows 0..5 of the tiles are the block rows; row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int t = 0; t < kGalerkinTiles; t++) {
            if (t < ntiles) {
                const int colg = 16 * t + r, sj = colg / 6, cdof = colg % 6;
#pragma unroll
                for (int reg = 0; reg < 2; reg++) {
                    const int row = kq + 4 * reg;
                    if (row < 6 && sj < gcnt) {
                        double v = acc[t][reg];
                        // coarse dofs without fine support (zero column of P): unit diagonal keeps the level matrix SPD
                        if (jcol[g0 + sj] == I && row == cdof && v == 0.0) v = 1.0;
                        Ac.vals[ell_index(base + (int64_t)(g0 + sj) * kSliceNodes + n, row, cdof)] = v;
                    }
                }
            }
        }
    }
}

// ---- tentative prolongator
// row d of the 6x6 near-null block of `node`
__device__ __forceinline__ void near_null_row(const NearNullSrc &S, int32_t node, int d, double out[6])
{
    if (S.B != nullptr) {
        const double2 *b = reinterpret_cast<const double2 *>(S.B + (int64_t)node * 36 + 6 * d);
        const double2 b0 = b[0], b1 = b[1], b2 = b[2];
        out[0] = b0.x; out[1] = b0.y; out[2] = b1.x; out[3] = b1.y; out[4] = b2.x; out[5] = b2.y;
        return;
    }
#pragma unroll
    for (int m = 0; m < 6; m++) out[m] = (m == d) ? 1.0 : 0.0;
    if (d < 3) {
        // u = omega x r: rotation about x -> (0,-z,y), about y -> (z,0,-x), about z -> (-y,x,0)
        const double x = S.xyz[3ll * node] - S.cx, y = S.xyz[3ll * node + 1] - S.cy, z = S.xyz[3ll * node + 2] - S.cz;
        out[3] = d == 1 ? -z : (d == 2 ? y : 0.0);
        out[4] = d == 0 ? z : (d == 2 ? -x : 0.0);
        out[5] = d == 0 ? -y : (d == 1 ? x : 0.0);
    } else if (S.normals != nullptr) {
        const double *nv = S.normals + 3ll * node;
        const double ni = nv[d - 3];
#pragma unroll
        for (int j = 0; j < 3; j++) out[3 + j] -= ni * nv[j];
    }
    if (S.dmask != nullptr && ((S.dmask[node] >> d) & 1u)) {
#pragma unroll
        for (int m = 0; m < 6; m++) out[m] = 0.0;
    }
}

__device__ __forceinline__ double wave_sum_all(double v) // the sum of the wave in every lane (lane 0's association order)
{
    v = wave_sum(v);
    return __shfl(v, 0, 64);
}

constexpr int kQrRowsPerLane = 4; // rows of an aggregate held in registers: 256 rows = 42 nodes; larger ones work in memory

// kInMemory = false: every lane keeps up to kQrRowsPerLane rows of the aggregate's 6k x 6 matrix in registers.
// kInMemory = true (aggregates of more than 42 nodes, or all with FEMSHELL_AMG_QR=memory): the rows live in Q itself.
template <bool kInMemory>
__global__ __launch_bounds__(64) void k_amg_tentative_qr(NearNullSrc S, const int32_t *__restrict__ aptr,
                                                         const int32_t *__restrict__ order, int32_t na, double *__restrict__ Q,
                                                         double *__restrict__ Bc, int all)
{
    const int lane = threadIdx.x;
    for (int32_t a = blockIdx.x; a < na; a += gridDim.x) {
        const int32_t p0 = aptr[a];
        const int rows = 6 * (aptr[a + 1] - p0);
        const bool big = rows > 64 * kQrRowsPerLane || all != 0;
        if (big != kInMemory) continue; // (uniform per wave) the other instantiation takes this aggregate
        double R[36];
#pragma unroll
        for (int e = 0; e < 36; e++) R[e] = 0.0;
        if (!kInMemory) {
            double m[kQrRowsPerLane][6];
            int64_t dst[kQrRowsPerLane];
#pragma unroll
            for (int q = 0; q < kQrRowsPerLane; q++) {
                const int r = lane + 64 * q;
                dst[q] = -1;
#pragma unroll
                for (int j = 0; j < 6; j++) m[q][j] = 0.0;
                if (r < rows) {
                    const int32_t node = order[p0 + r / 6];
                    near_null_row(S, node, r % 6, m[q]);
                    dst[q] = (int64_t)node * 36 + 6 * (r % 6);
                }
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < kQrRowsPerLane; q++) s += m[q][j] * m[q][j];
                const double n0 = sqrt(wave_sum_all(s));
#pragma unroll
                for (int pass = 0; pass < 2; pass++)
#pragma unroll
                    for (int i = 0; i < j; i++) {
                        double cpart = 0.0;
#pragma unroll
                        for (int q = 0; q < kQrRowsPerLane; q++) cpart += m[q][i] * m[q][j];
                        const double cc = wave_sum_all(cpart);
#pragma unroll
                        for (int q = 0; q < kQrRowsPerLane; q++) m[q][j] -= cc * m[q][i];
                        R[6 * i + j] += cc;
                    }
                s = 0.0;
#pragma unroll
                for (int q = 0; q < kQrRowsPerLane; q++) s += m[q][j] * m[q][j];
                const double nj = sqrt(wave_sum_all(s));
                const bool keep = n0 > 0.0 && nj > 1e-8 * n0; // else: dependent column, no coarse dof here
                R[6 * j + j] = keep ? nj : 0.0;
#pragma unroll
                for (int q = 0; q < kQrRowsPerLane; q++) m[q][j] = keep ? m[q][j] / nj : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kQrRowsPerLane; q++)
                if (dst[q] >= 0) {
                    double2 *o = reinterpret_cast<double2 *>(Q + dst[q]);
                    o[0] = make_double2(m[q][0], m[q][1]);
                    o[1] = make_double2(m[q][2], m[q][3]);
                    o[2] = make_double2(m[q][4], m[q][5]);
                }
        } else {
            // rows in Q (global memory, L2-resident for one aggregate); only this wave touches them
            for (int r = lane; r < rows; r += 64) {
                const int32_t node = order[p0 + r / 6];
                double v[6];
                near_null_row(S, node, r % 6, v);
                double *o = Q + (int64_t)node * 36 + 6 * (r % 6);
#pragma unroll
                for (int j = 0; j < 6; j++) o[j] = v[j];
            }
            auto row_ptr = [&](int r) { return Q + (int64_t)order[p0 + r / 6] * 36 + 6 * (r % 6); };
            for (int j = 0; j < 6; j++) {
                double s = 0.0;
                for (int r = lane; r < rows; r += 64) { const double v = row_ptr(r)[j]; s += v * v; }
                const double n0 = sqrt(wave_sum_all(s));
                for (int pass = 0; pass < 2; pass++)
                    for (int i = 0; i < j; i++) {
                        double cpart = 0.0;
                        for (int r = lane; r < rows; r += 64) { const double *o = row_ptr(r); cpart += o[i] * o[j]; }
                        const double cc = wave_sum_all(cpart);
                        for (int r = lane; r < rows; r += 64) { double *o = row_ptr(r); o[j] -= cc * o[i]; }
                        R[6 * i + j] += cc;
                    }
                s = 0.0;
                for (int r = lane; r < rows; r += 64) { const double v = row_ptr(r)[j]; s += v * v; }
                const double nj = sqrt(wave_sum_all(s));
                const bool keep = n0 > 0.0 && nj > 1e-8 * n0;
                R[6 * j + j] = keep ? nj : 0.0;
                for (int r = lane; r < rows; r += 64) { double *o = row_ptr(r); o[j] = keep ? o[j] / nj : 0.0; }
            }
        }
        if (lane == 0) {
            double *o = Bc + (int64_t)a * 36;
#pragma unroll
            for (int e = 0; e < 36; e++) o[e] = R[e];
        }
    }
}

void launch_amg_tentative_qr(const NearNullSrc &B, const int32_t *aptr, const int32_t *order, int32_t na, int32_t largest,
                             double *Q, double *Bc, bool rows_in_memory, hipStream_t st)
{
    if (na <= 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(na, 1 << 20);
    const int all = rows_in_memory ? 1 : 0;
    if (!rows_in_memory)
        hipLaunchKernelGGL(k_amg_tentative_qr<false>, dim3(grid), dim3(64), 0, st, B, aptr, order, na, Q, Bc, all);
    if (rows_in_memory || 6 * largest > 64 * kQrRowsPerLane)
        hipLaunchKernelGGL(k_amg_tentative_qr<true>, dim3(grid), dim3(64), 0, st, B, aptr, order, na, Q, Bc, all);
}

void launch_amg_prolongator(const DeviceMatrix &A, const int32_t *agg, const double *Q, double omega, const uint8_t *pmap_own,
                            const uint8_t *pmap_in, const EllView &P, hipStream_t st)
{
    hipLaunchKernelGGL(k_amg_prolongator, dim3((unsigned)((P.total + 127) / 128)), dim3(128), 0, st, A, agg, Q, omega, pmap_own,
                       pmap_in, P);
}

void launch_amg_ap(const DeviceMatrix &A, const EllView &P, const EllView &AP, hipStream_t st)
{
    hipLaunchKernelGGL(k_amg_ap, dim3((unsigned)((AP.total + 127) / 128)), dim3(128), 0, st, A, P, AP);
}

void launch_amg_restriction(const EllView &P, const int64_t *rptr, const int32_t *rrow, const uint8_t *rk, const EllView &R,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_amg_restriction, dim3((unsigned)((R.total + 127) / 128)), dim3(128), 0, st, P, rptr, rrow, rk, R);
}

void launch_amg_galerkin(const EllView &P, const EllView &AP, const int64_t *rptr, const int32_t *rrow, const uint8_t *rk,
                         const EllView &Ac, hipStream_t st, bool mfma, int diag_key)
{
    if (mfma && diag_key == 0) // one wave per coarse row (padding rows of the last slice included: they get zero blocks)
        hipLaunchKernelGGL(k_amg_galerkin_mfma, dim3((unsigned)(Ac.n_slices * kSliceNodes)), dim3(64), 0, st, P, AP, rptr, rrow, rk, Ac);
    else
        hipLaunchKernelGGL(k_amg_galerkin, dim3((unsigned)((Ac.total + 127) / 128)), dim3(128), 0, st, P, AP, rptr, rrow, rk, Ac,
                           diag_key);
}

// ---- rows of an ELL operator on their way to another rank (amg_dist.cpp): W entries of 37 doubles per node -- the column
// key (-1: no block) and the block, row-major
__global__ __launch_bounds__(128) void k_pack_ell_rows(EllView M, int contig, const int32_t *__restrict__ nodes, int32_t count, int W,
                                                       double *__restrict__ buf)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)count * W) return;
    const int r = nodes[t / W], k = (int)(t % W);
    double *o = buf + t * 37;
    if (k < M.count[r]) {
        const int64_t slot = ell_slot(M, r / kSliceNodes, k, r % kSliceNodes);
        double b[36];
        if (contig) blk_load_contig(M.vals, slot, b);
        else ell_load(M.vals, slot, b, false);
        o[0] = (double)M.cols[slot];
#pragma unroll
        for (int e = 0; e < 36; e++) o[1 + e] = b[e];
    } else {
        o[0] = -1.0;
#pragma unroll
        for (int e = 0; e < 36; e++) o[1 + e] = 0.0;
    }
}

void launch_pack_ell_rows(const EllView &M, bool contig, const int32_t *nodes, int32_t count, int W, double *buf, hipStream_t st)
{
    const int64_t n = (int64_t)count * W;
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_ell_rows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, M, contig ? 1 : 0, nodes, count, W, buf);
}

__global__ void k_extract_keys(const double *__restrict__ buf, int64_t entries, int32_t *__restrict__ keys)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < entries) keys[t] = (int32_t)buf[t * 37];
}

void launch_extract_keys(const double *buf, int64_t entries, int32_t *keys, hipStream_t st)
{
    if (entries == 0) return;
    hipLaunchKernelGGL(k_extract_keys, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, buf, entries, keys);
}

// the received rows become the rows first_row ... of M: the host has compacted their keys into M.cols / M.count (entry k of
// the buffer with a key >= 0 is block number [its rank among the row's valid entries]); every slot of these rows is written
__global__ __launch_bounds__(128) void k_unpack_ell_rows(const double *__restrict__ buf, int32_t count, int W, EllView M, int contig,
                                                         int32_t first_row)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)count * W) return;
    const int g = (int)(t / W), k = (int)(t % W);
    const int r = first_row + g;
    const int sl = r / kSliceNodes, n = r % kSliceNodes;
    if (k >= M.slice_width[sl]) return;
    const int64_t slot = ell_slot(M, sl, k, n);
    double b[36];
#pragma unroll
    for (int e = 0; e < 36; e++) b[e] = 0.0;
    if (k < M.count[r]) {
        // the k-th valid entry of the row in the buffer
        int seen = 0;
        for (int q = 0; q < W; q++) {
            const double *e = buf + ((int64_t)g * W + q) * 37;
            if (e[0] >= 0.0) {
                if (seen == k) {
#pragma unroll
                    for (int v = 0; v < 36; v++) b[v] = e[1 + v];
                    break;
                }
                seen++;
            }
        }
    }
    if (contig) blk_store_contig(M.vals, slot, b);
    else ell_store(M.vals, slot, b);
}

void launch_unpack_ell_rows(const double *buf, int32_t count, int W, const EllView &M, bool contig, int32_t first_row, hipStream_t st)
{
    const int64_t n = (int64_t)count * W;
    if (n == 0) return;
    hipLaunchKernelGGL(k_unpack_ell_rows, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, buf, count, W, M, contig ? 1 : 0, first_row);
}

// ---- patch smoother (amg_patch.hpp): clusters of rigidly coupled nodes --------------------------------------------------

namespace {
__device__ __forceinline__ void load_dinv(const DeviceMatrix &A, int a, double d[36])
{
    const double *mi = A.minv + (int64_t)(a / kSliceNodes) * kMinvWords * kSliceNodes + (a % kSliceNodes);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) d[6 * i + j] = mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
}
} // namespace

// one lane per owned row: the stored blocks (a, c) with another owned row c -- each pair once (symmetric storage stores it once;
// full storage: c > a) -- whose sigma_max^2 exceeds tau2 go to the edge list (order arbitrary: the host sorts)
// (counter[0]: edges above tau; counter[1]: edges above the trigger level hi2 -- nearly coincident nodes --; counter[2]: pairs looked at)
__global__ __launch_bounds__(64) void k_patch_sigma(DeviceMatrix A, double tau2, double hi2, PatchEdge *edges, unsigned int *counter, unsigned int cap)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A.n_own) return;
    unsigned int pairs = 0, high = 0;
    const int sl = a / kSliceNodes, n = a % kSliceNodes;
    const int64_t base = A.slice_base[sl];
    const int W = A.slice_width[sl];
    double Da[36];
    bool have = false;
    for (int k = 1; k < W; k++) {
        const int64_t slot = base + (int64_t)k * kSliceNodes + n;
        const int c = A.cols[slot];
        if (c == a || c >= A.n_own || (!A.symmetric && c < a)) continue;
        if (!have) {
            load_dinv(A, a, Da);
            have = true;
        }
        double blk[36], Dc[36];
        ell_load(A.vals, slot, blk, false);
        load_dinv(A, c, Dc);
        const double s2 = patch_sigma2(Da, blk, Dc, tau2);
        pairs++;
        if (s2 > hi2) high++;
        if (s2 > tau2) {
            const unsigned int at = atomicAdd(counter, 1u);
            if (at < cap) {
                PatchEdge e;
                e.a = a < c ? a : c;
                e.c = a < c ? c : a;
                e.sigma2 = s2;
                edges[at] = e;
            }
        }
    }
    if (high) atomicAdd(counter + 1, high);
    if (pairs) atomicAdd(counter + 2, pairs);
}

void launch_patch_sigma(const DeviceMatrix &A, double tau, double trigger_sigma, PatchEdge *edges, unsigned int *counter, unsigned int cap,
                        hipStream_t st)
{
    if (A.n_own <= 0) return;
    hipLaunchKernelGGL(k_patch_sigma, dim3((unsigned)((A.n_own + 63) / 64)), dim3(64), 0, st, A, tau * tau, trigger_sigma * trigger_sigma, edges,
                       counter, cap);
}

// the dense diagonal block A_cc of every cluster (row-major (6 m)^2 at moff[c]) and the inverse diagonal blocks of its members
// (36 doubles per member position): one thread per (cluster, i, j)
__global__ __launch_bounds__(128) void k_patch_gather(DeviceMatrix A, PatchView pv, double *Bc, double *dinv_of_member)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t / (kPatchMaxNodes * kPatchMaxNodes)), ij = (int)(t % (kPatchMaxNodes * kPatchMaxNodes));
    if (c >= pv.n_clusters) return;
    const int m = pv.ptr[c + 1] - pv.ptr[c], i = ij / kPatchMaxNodes, j = ij % kPatchMaxNodes;
    if (i >= m || j >= m) return;
    const int a = pv.nodes[pv.ptr[c] + i], b = pv.nodes[pv.ptr[c] + j];
    double blk[36];
#pragma unroll
    for (int e = 0; e < 36; e++) blk[e] = 0.0;
    if (i == j) {
        const int64_t slot = A.slice_base[a / kSliceNodes] + (a % kSliceNodes);
        if (A.diag_upper) ell_load_sym(A.vals, slot, blk);
        else ell_load(A.vals, slot, blk, false);
        double d[36];
        load_dinv(A, a, d);
#pragma unroll
        for (int e = 0; e < 36; e++) dinv_of_member[(int64_t)(pv.ptr[c] + i) * 36 + e] = d[e];
    } else {
        // A_ab: a stored block of row a, or the transpose of a stored block of row b
        bool found = false;
        for (int pass = 0; pass < 2 && !found; pass++) {
            const int r = pass == 0 ? a : b, want = pass == 0 ? b : a;
            const int sl = r / kSliceNodes, n = r % kSliceNodes;
            const int W = A.slice_width[sl];
            for (int k = 1; k < W; k++) {
                const int64_t slot = A.slice_base[sl] + (int64_t)k * kSliceNodes + n;
                if (A.cols[slot] == want) {
                    ell_load(A.vals, slot, blk, pass == 1);
                    found = true;
                    break;
                }
            }
        }
    }
    const int N = 6 * m;
    double *B = Bc + pv.moff[c];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int q = 0; q < 6; q++) B[(int64_t)(6 * i + r) * N + 6 * j + q] = blk[6 * r + q];
}

void launch_patch_gather(const DeviceMatrix &A, const PatchView &pv, double *Bc, double *dinv_of_member, hipStream_t st)
{
    const int64_t n = (int64_t)pv.n_clusters * kPatchMaxNodes * kPatchMaxNodes;
    if (n == 0) return;
    hipLaunchKernelGGL(k_patch_gather, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, A, pv, Bc, dinv_of_member);
}

// z_c += coef M_c r_c on the clustered nodes, for up to two targets (the Chebyshev direction, as floats where the level keeps it so,
// and the iterate): one wave per cluster, lane t < 6 m owns dof t of the cluster
__global__ __launch_bounds__(64) void k_patch_correct(PatchView pv, const double *__restrict__ r, double coef, double *d, int d_float,
                                                      double *x, const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    const int c = blockIdx.x, t = threadIdx.x;
    const int m = pv.ptr[c + 1] - pv.ptr[c], N = 6 * m;
    const int node = t < N ? pv.nodes[pv.ptr[c] + t / 6] : 0;
    const int64_t row = (int64_t)node * 6 + t % 6;
    const double rv = t < N ? r[row] : 0.0;
    const double *Mrow = pv.M + pv.moff[c] + (int64_t)(t < N ? t : 0) * N;
    double acc = 0.0;
    for (int s = 0; s < N; s++) acc += Mrow[s] * __shfl(rv, s, 64);
    if (t >= N) return;
    const double delta = coef * acc;
    if (d != nullptr) {
        if (d_float) reinterpret_cast<float *>(d)[row] = (float)((double)reinterpret_cast<float *>(d)[row] + delta);
        else d[row] += delta;
    }
    if (x != nullptr) x[row] += delta;
}

void launch_patch_correct(const PatchView &pv, const double *r, double coef, double *d, bool d_float, double *x, const CgScalars *gate,
                          hipStream_t st)
{
    if (pv.n_clusters == 0) return;
    hipLaunchKernelGGL(k_patch_correct, dim3((unsigned)pv.n_clusters), dim3(64), 0, st, pv, r, coef, d, d_float ? 1 : 0, x, gate);
}

// The smoothing of the prolongator with the cluster blocks: P = P0 - omega B^-1 A P0 = (the point-block result of k_amg_prolongator)
// - omega M (A P0) on the rows of clustered nodes.  One thread per (member position, slot of P's row): for the aggregate J of that
// slot, sum over the cluster's members j of M_ij times row j of A P0 at J = sum over the neighbours k of j in aggregate J of A_jk Q_k.
__global__ __launch_bounds__(64) void k_patch_prolongator(DeviceMatrix A, const int32_t *__restrict__ agg, const double *__restrict__ Q,
                                                          double omega, EllView P, PatchView pv, int width)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pos = (int)(t / width), kP = (int)(t % width);
    if (pos >= pv.n_members) return;
    const int a = pv.nodes[pos], c = pv.cluster_of[pos];
    if (kP >= P.count[a] || (pv.in_p != nullptr && !pv.in_p[c])) return;
    const int64_t pslot = ell_slot(P, a / kSliceNodes, kP, a % kSliceNodes);
    const int J = P.cols[pslot];
    const int m = pv.ptr[c + 1] - pv.ptr[c], N = 6 * m, i = pos - pv.ptr[c];
    const double *M = pv.M + pv.moff[c];
    double out[36];
#pragma unroll
    for (int e = 0; e < 36; e++) out[e] = 0.0;
    for (int j = 0; j < m; j++) {
        const int b = pv.nodes[pv.ptr[c] + j];
        double T[36];
#pragma unroll
        for (int e = 0; e < 36; e++) T[e] = 0.0;
        bool any = false;
        for_each_neighbour(A, b, [&](int k, int64_t slot, bool transposed, int64_t) {
            if (agg[k] != J) return;
            double blk[36], q[36];
            if (A.diag_upper && k == b && !transposed) ell_load_sym(A.vals, slot, blk);
            else ell_load(A.vals, slot, blk, transposed);
#pragma unroll
            for (int e = 0; e < 36; e++) q[e] = Q[(int64_t)k * 36 + e];
            blk_mac(blk, q, T, false);
            any = true;
        });
        if (!any) continue;
        double Mij[36];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = 0; q < 6; q++) Mij[6 * r + q] = M[(int64_t)(6 * i + r) * N + 6 * j + q];
        blk_mac(Mij, T, out, false);
    }
    double cur[36];
    ell_load(P.vals, pslot, cur, false);
#pragma unroll
    for (int e = 0; e < 36; e++) cur[e] -= omega * out[e];
    ell_store(P.vals, pslot, cur);
}

void launch_patch_prolongator(const DeviceMatrix &A, const int32_t *agg, const double *Q, double omega, const EllView &P, const PatchView &pv,
                              int width, hipStream_t st)
{
    const int64_t n = (int64_t)pv.n_members * width;
    if (n == 0) return;
    hipLaunchKernelGGL(k_patch_prolongator, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, A, agg, Q, omega, P, pv, width);
}

// partial sums of z.z over the first n entries, one per workgroup of the grid G (the layout launch_minv_apply_norm writes)
__global__ __launch_bounds__(256) void k_sqnorm_partials(const double *__restrict__ z, int64_t n, double *__restrict__ partials)
{
    __shared__ double sh[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, b = (int64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    double acc = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) acc += z[i] * z[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

void launch_sqnorm_partials(const double *z, int64_t n, double *partials, int G, hipStream_t st)
{
    hipLaunchKernelGGL(k_sqnorm_partials, dim3((unsigned)G), dim3(256), 0, st, z, n, partials);
}

// out[j] = part[j * G] + part[j * G + 1] + ... in INDEX ORDER, j = 0, 1 -- the sums the host took over the partial norms of a power
// iteration after bringing all 2 G of them over (1 MB at 4M triangles, 6.7 ms as a pageable copy).  Wave j of the one workgroup
// stages 1024 partials at a time in LDS with coalesced loads; its first lane adds them one after the other.
__global__ __launch_bounds__(128) void k_sums_in_order(const double *__restrict__ part, int G, double *__restrict__ out)
{
    __shared__ double buf[2][1024];
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *p = part + (size_t)j * G;
    double s = 0.0;
    for (int base = 0; base < G; base += 1024) {
        const int m = min(1024, G - base);
        for (int i = lane; i < m; i += 64) buf[j][i] = p[base + i];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0)
            for (int i = 0; i < m; i++) s += buf[j][i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) out[j] = s;
}
void launch_sums_in_order(const double *part, int G, double *out2, hipStream_t st)
{
    hipLaunchKernelGGL(k_sums_in_order, dim3(1), dim3(128), 0, st, part, G, out2);
}

} // namespace femshell
