// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libfemshell.
//
// All kernels are HBM-bandwidth-bound FP64 streaming kernels; none uses MFMA (the blocks
// are 6x6 and there is a single right-hand side, so there is no GEMM-shaped contraction).
// Layout rules they share (plan.hpp): a slice = 32 node rows = 192 scalar rows; per-slice
// kernels map one lane to one scalar row so that every matrix/vector stream is read with
// consecutive lanes on consecutive 16-byte (matrix) or 8-byte (vector) words; workgroup b
// works on slice (b%8)*ceil(S/8) + b/8 so that each XCD (blocks are dealt round-robin over
// the 8 XCDs) sweeps one contiguous eighth of the rows and its L2 keeps the x entries shared
// by neighbouring slices.
#include "kernels.hpp"
#include "plan.hpp"

#include <cstdlib>

namespace femshell {

static_assert(kSliceNodes == 32 && kSliceRows == 192, "kernels assume 32-node slices");

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

static int assemble_grid(const DeviceMatrix &m)
{
    static const int cap = [] {
        const char *e = getenv("FEMSHELL_ASM_GRID");
        return e ? atoi(e) : 2048;
    }();
    const int g = 8 * ((m.n_slices + 7) / 8);
    return g < cap ? g : cap;
}

constexpr int kMaxGrid = 2560; // 256 CUs x 10 resident 192-thread workgroups

int slice_grid(const DeviceMatrix &m)
{
    const int g = 8 * ((m.n_slices + 7) / 8);
    return g < kMaxGrid ? g : kMaxGrid;
}

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

// =====================================================================================
// Assembly: replaces the element loop of assemble_elasticity (fem-shell.cpp:1197-1232).
// One lane owns one 6x6 block slot of K (slot k of node n of the slice) and sums the
// contributions K_e(ia,ib) of the elements in the slot's gather list, in element order.
// Slot 0 is the diagonal block (about six elements), the other slots are edge blocks
// (two elements), so putting the diagonal slots of a slice into one wave keeps the other
// waves divergence-free.  Every block is written exactly once: no atomics, no zero-fill
// pass, bitwise reproducible.
// Dirichlet dofs follow libMesh's constrain_element_matrix_and_vector (fem-shell.cpp:1227):
// rows and columns of fixed dofs are zero, the diagonal entry is the number of elements
// touching the node.
// =====================================================================================
template <int kWavesPerSimd>
__global__ __launch_bounds__(256, kWavesPerSimd) void k_assemble(DeviceMatrix m, MatConst mc)
{
    // LDS: [output tile | ownership mask | element records | partial-sum staging]
    extern __shared__ double lds[];
    double2 *lds_tile = reinterpret_cast<double2 *>(lds);                                  // kOutSlots*32*36 doubles
    uint32_t *lds_mask = reinterpret_cast<uint32_t *>(lds + kOutSlots * kSliceNodes * 36); // 64 words
    double *lds_rec = lds + kOutSlots * kSliceNodes * 36 + 32;
    double *lds_stage = lds_rec + (size_t)m.max_slice_elems * kRecDoubles;
    const int tid = threadIdx.x;

    SliceWalk w(m.n_slices);
    if (!w.valid()) return;
    // software pipeline over slices: the node ids of the next slice's elements are fetched while
    // the current slice computes, so that a slice exposes one dependent load (the coordinates)
    int e0 = m.slice_elem_ptr[w.s], ne = m.slice_elem_ptr[w.s + 1] - e0;
    int4 nd = make_int4(0, 0, 0, -1);
    if (tid < ne) nd = m.slice_elem_nodes[e0 + tid];

    for (; w.valid(); w.next()) {
        const int s = w.s;
        const int64_t base = m.slice_base[s];
        const int W = m.slice_width[s];
        const int i0 = m.item_ptr[s], ni = m.item_ptr[s + 1] - i0;
        uint4 item = make_uint4(0, 0, 0, 0);
        if (tid < ni) item = m.items[i0 + tid]; // in flight during phase A

        // ---- phase A: one record per element touching the slice
        for (int i = tid; i < ne; i += blockDim.x) {
            const int4 c = (i == tid) ? nd : m.slice_elem_nodes[e0 + i];
            double rec[kRecDoubles];
            bool ok = false;
            if (c.w < 0) {
                double X[9];
                const double *pa = m.xyz + 3 * (int64_t)c.x, *pb = m.xyz + 3 * (int64_t)c.y,
                             *pc = m.xyz + 3 * (int64_t)c.z;
                X[0] = pa[0]; X[1] = pa[1]; X[2] = pa[2];
                X[3] = pb[0]; X[4] = pb[1]; X[5] = pb[2];
                X[6] = pc[0]; X[7] = pc[1]; X[8] = pc[2];
                ok = tri3_record(X, mc, rec);
            } else {
#pragma unroll
                for (int q = 0; q < kRecDoubles; q++) rec[q] = 0.0;
            }
            if (!ok) atomicCAS(m.status, 0, e0 + i + 1);
            double2 *dst = reinterpret_cast<double2 *>(lds_rec + (size_t)i * kRecDoubles);
#pragma unroll
            for (int q = 0; q < kRecDoubles / 2; q++) dst[q] = make_double2(rec[2 * q], rec[2 * q + 1]);
        }
        // prefetch the next slice's element node ids
        {
            const int s2 = s + w.step;
            if (s2 < w.last) {
                e0 = m.slice_elem_ptr[s2];
                ne = m.slice_elem_ptr[s2 + 1] - e0;
                if (tid < ne) nd = m.slice_elem_nodes[e0 + tid];
            }
        }
        __syncthreads();

        // ---- phase B: one lane per work item (at most kItemPairs element contributions), in
        //      rounds of 256 items; each round's finished blocks leave through the LDS tile so
        //      that every store instruction covers 1 KiB of consecutive addresses
        double2 *out = reinterpret_cast<double2 *>(m.vals + base * 36);
        const bool multi = ni > (int)blockDim.x; // several rounds: the tile is only partly owned per round
        for (int r0 = 0; r0 < ni; r0 += blockDim.x) {
            const int it = r0 + tid;
            const bool live = it < ni;
            if (r0 > 0) {
                item = make_uint4(0, 0, 0, 0);
                if (live) item = m.items[i0 + it];
            }
            if (multi && tid < 64) lds_mask[tid] = 0u;
            const int slot_in_slice = (int)(item.x & 0xffffu), chunk = (int)((item.x >> 16) & 0xffu),
                      nchunks = (int)(item.x >> 24);
            const int cnt = (int)(item.z >> 16);
            double blk[36];
#pragma unroll
            for (int i = 0; i < 36; i++) blk[i] = 0.0;
            for (int q = 0; q < cnt; q++) {
                const uint32_t pr = (q == 0) ? (item.y & 0xffffu) : (q == 1 ? (item.y >> 16) : (item.z & 0xffffu));
                tri3_block_add_rec(lds_rec + (size_t)(pr >> 4) * kRecDoubles, (int)((pr >> 2) & 3u), (int)(pr & 3u), mc,
                                   blk);
            }
            const bool owner = live && chunk == 0 && nchunks > 0; // nchunks == 0: padding item
            // column node and Dirichlet masks of the slot (needed after the reduction)
            int col = 0, valence = 0;
            uint32_t mrow = 0, mcol = 0;
            if (owner) {
                const int64_t slot = base + slot_in_slice;
                col = m.cols[slot];
                mrow = m.dmask[s * kSliceNodes + (slot_in_slice & 31)];
                mcol = m.dmask[col];
                valence = m.pair_ptr[slot + 1] - m.pair_ptr[slot];
            }
            if (live && chunk > 0) {
                double2 *st = reinterpret_cast<double2 *>(lds_stage + (size_t)item.w * 36);
#pragma unroll
                for (int i = 0; i < 18; i++) st[i] = make_double2(blk[2 * i], blk[2 * i + 1]);
            }
            __syncthreads();
            if (owner) {
                // chunks > 0 sort after chunk 0, so with several rounds they may not have run yet:
                // plan.cpp keeps all chunks of a slot in one round when a slice has several rounds
                for (int c = 1; c < nchunks; c++) {
                    const double2 *st = reinterpret_cast<const double2 *>(lds_stage + (size_t)(item.w + c - 1) * 36);
#pragma unroll
                    for (int i = 0; i < 18; i++) {
                        const double2 v = st[i];
                        blk[2 * i] += v.x;
                        blk[2 * i + 1] += v.y;
                    }
                }
                if (mrow | mcol) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++)
                            if (((mrow >> i) & 1u) | ((mcol >> j) & 1u)) blk[6 * i + j] = 0.0;
                    if (col == s * kSliceNodes + (slot_in_slice & 31)) {
#pragma unroll
                        for (int i = 0; i < 6; i++)
                            if ((mrow >> i) & 1u) blk[7 * i] = (double)valence;
                    }
                }
                if (multi) atomicOr(&lds_mask[slot_in_slice >> 5], 1u << (slot_in_slice & 31));
            }
            const int my_k = slot_in_slice >> 5, my_n = slot_in_slice & 31;
            for (int k0 = 0; k0 < W; k0 += kOutSlots) {
                if (owner && my_k >= k0 && my_k < k0 + kOutSlots) {
                    double2 *t = lds_tile + (size_t)(my_k - k0) * 3 * kSliceRows + my_n * 6;
#pragma unroll
                    for (int jp = 0; jp < 3; jp++)
#pragma unroll
                        for (int i = 0; i < 6; i++)
                            t[jp * kSliceRows + i] = make_double2(blk[6 * i + 2 * jp], blk[6 * i + 2 * jp + 1]);
                }
                __syncthreads();
                const int nk = min(kOutSlots, W - k0);
                const int words = nk * 3 * kSliceRows; // double2 words of this pass
                double2 *dst = out + (size_t)k0 * 3 * kSliceRows;
                if (!multi) {
                    for (int q = tid; q < words; q += blockDim.x) dst[q] = lds_tile[q];
                } else {
                    for (int q = tid; q < words; q += blockDim.x) {
                        const int kk = k0 + q / (3 * kSliceRows), nn = (q % kSliceRows) / 6;
                        if ((lds_mask[kk] >> nn) & 1u) dst[q] = lds_tile[q];
                    }
                }
                __syncthreads();
            }
        }
    }
}

void launch_assemble(const DeviceMatrix &m, const MatConst &mc, hipStream_t st)
{
    const size_t lds = (size_t)m.lds_bytes;
    static const int variant = [] {
        const char *e = getenv("FEMSHELL_ASM_WAVES"); // tuning knob: waves per SIMD the kernel is compiled for
        return e ? atoi(e) : 2;
    }();
    const int g = assemble_grid(m);
    switch (variant) {
    case 1: hipLaunchKernelGGL(k_assemble<1>, dim3(g), dim3(256), lds, st, m, mc); break;
    case 3: hipLaunchKernelGGL(k_assemble<3>, dim3(g), dim3(256), lds, st, m, mc); break;
    case 4: hipLaunchKernelGGL(k_assemble<4>, dim3(g), dim3(256), lds, st, m, mc); break;
    default: hipLaunchKernelGGL(k_assemble<2>, dim3(g), dim3(256), lds, st, m, mc); break;
    }
}

// Right-hand side: contribRHS (fem-shell.cpp:1118-1153) is a masked copy -- every node's load
// enters once, fixed dofs get 0 (fem-shell.cpp:1227).
__global__ void k_rhs(DeviceMatrix m, const double *loads, double *F)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)m.n_pad * 6) return;
    const int node = (int)(i / 6), v = (int)(i % 6);
    const bool fixed = (m.dmask[node] >> v) & 1u;
    F[i] = (fixed || node >= m.n_own) ? 0.0 : loads[i];
}

void launch_rhs(const DeviceMatrix &m, const double *loads, double *F, hipStream_t st)
{
    const int64_t n = (int64_t)m.n_pad * 6;
    hipLaunchKernelGGL(k_rhs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, loads, F);
}

// Unconstrained element matrices in the reference's variable-major ordering
// (fem-shell.cpp:1105-1109); one lane per node block.  Parity/debug export only.
__global__ __launch_bounds__(128) void k_element_matrices(DeviceMatrix m, MatConst mc, int first, int count, double *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)count * 9) return;
    const int e = (int)(t / 9), pr = (int)(t % 9), ia = pr / 3, ib = pr % 3;
    const int32_t *c = m.tri + 3 * (int64_t)(first + e);
    double X[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int d = 0; d < 3; d++) X[3 * i + d] = m.xyz[3 * (int64_t)c[i] + d];
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    if (!tri3_block_add(X, ia, ib, mc, acc)) atomicCAS(m.status, 0, kStatusDirect + first + e);
    double *Ke = out + (int64_t)e * 324;
#pragma unroll
    for (int al = 0; al < 6; al++)
#pragma unroll
        for (int be = 0; be < 6; be++) Ke[(3 * al + ia) * 18 + 3 * be + ib] = acc[6 * al + be];
}

void launch_element_matrices(const DeviceMatrix &m, const MatConst &mc, int32_t first, int32_t count,
                             double *Ke_out, hipStream_t st)
{
    const int64_t n = (int64_t)count * 9;
    hipLaunchKernelGGL(k_element_matrices, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, m, mc, first,
                       count, Ke_out);
}

// =====================================================================================
// Block-Jacobi setup: invert the 6x6 diagonal block of every owned node (Gauss-Jordan, no
// pivoting: the blocks are SPD) into the row-interleaved layout the CG update kernel streams.
// =====================================================================================
__global__ void k_block_jacobi(DeviceMatrix m)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m.n_pad) return;
    const int s = a / kSliceNodes, n = a % kSliceNodes;
    double *mi = m.minv + (int64_t)s * 6 * kSliceRows + n * 6;
    if (a >= m.n_own) {
#pragma unroll
        for (int j = 0; j < 6; j++)
#pragma unroll
            for (int i = 0; i < 6; i++) mi[j * kSliceRows + i] = 0.0;
        return;
    }
    const double2 *src = reinterpret_cast<const double2 *>(m.vals + m.slice_base[s] * 36);
    double A[6][6], B[6][6];
#pragma unroll
    for (int jp = 0; jp < 3; jp++)
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double2 v = src[jp * kSliceRows + n * 6 + i];
            A[i][2 * jp] = v.x;
            A[i][2 * jp + 1] = v.y;
        }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) B[i][j] = (i == j) ? 1.0 : 0.0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const double piv = A[c][c];
        if (!(piv > 0.0)) ok = false;
        const double d = 1.0 / piv;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            A[c][j] *= d;
            B[c][j] *= d;
        }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = A[r][c];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                A[r][j] -= f * A[c][j];
                B[r][j] -= f * B[c][j];
            }
        }
    }
    if (!ok) {
        atomicCAS(m.status, 0, -(a + 1));
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) B[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int i = 0; i < 6; i++) mi[j * kSliceRows + i] = B[i][j];
}

void launch_block_jacobi(const DeviceMatrix &m, hipStream_t st)
{
    hipLaunchKernelGGL(k_block_jacobi, dim3((m.n_pad + 127) / 128), dim3(128), 0, st, m);
}

// =====================================================================================
// SpMV y = K x on the sliced block ELL layout, one lane per scalar row.  Per block slot a
// lane issues three 16-byte loads of K (consecutive lanes -> consecutive 16-byte words, 1 KiB
// per wave instruction) and three 16-byte loads of x (the six lanes of a node read the same
// words; neighbouring slices share them through L2).  Optionally fuses the partial sums of
// x.y needed by CG (p.Ap).
// =====================================================================================
__global__ __launch_bounds__(192) void k_spmv(DeviceMatrix m, const double *__restrict__ x,
                                              double *__restrict__ y, double *__restrict__ partials,
                                              const CgScalars *s)
{
    __shared__ double sh[3];
    if (s != nullptr && s->done != 0) return;
    const int t = threadIdx.x;
    double dotv = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t base = m.slice_base[sl];
        const int W = m.slice_width[sl];
        const int32_t *c = m.cols + base + t / 6;
        const double2 *v = reinterpret_cast<const double2 *>(m.vals + base * 36) + t;
        double acc = 0.0;
        for (int k = 0; k < W; k++) {
            const int col = c[k * kSliceNodes];
            const double2 *xv = reinterpret_cast<const double2 *>(x + 6 * (int64_t)col);
            const double2 a0 = v[(k * 3 + 0) * kSliceRows];
            const double2 a1 = v[(k * 3 + 1) * kSliceRows];
            const double2 a2 = v[(k * 3 + 2) * kSliceRows];
            const double2 x0 = xv[0], x1 = xv[1], x2 = xv[2];
            acc += a0.x * x0.x;
            acc += a0.y * x0.y;
            acc += a1.x * x1.x;
            acc += a1.y * x1.y;
            acc += a2.x * x2.x;
            acc += a2.y * x2.y;
        }
        const int64_t row = (int64_t)sl * kSliceRows + t;
        y[row] = acc;
        if (partials != nullptr) dotv += acc * x[row];
    }
    if (partials != nullptr) {
        const double tot = block_sum(dotv, sh);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

void launch_spmv(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                 hipStream_t st)
{
    hipLaunchKernelGGL(k_spmv, dim3(slice_grid(m)), dim3(192), 0, st, m, x, y, partials, s);
}

// =====================================================================================
// CG vector kernels (one lane per scalar row, one workgroup per slice)
// =====================================================================================

// z_row = sum_j Minv[row][j] * r[node*6+j], r of the slice staged in LDS
__device__ __forceinline__ double apply_minv(const DeviceMatrix &m, int sl, int t, const double *rs)
{
    const double *mi = m.minv + (int64_t)sl * 6 * kSliceRows + t;
    const int nb = (t / 6) * 6;
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) z += mi[j * kSliceRows] * rs[nb + j];
    return z;
}

__global__ __launch_bounds__(192) void k_cg_init(DeviceMatrix m, CgVectors v)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    const int G = gridDim.x, t = threadIdx.x;
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const double bv = v.b[row];
        __syncthreads();
        rs[t] = bv;
        __syncthreads();
        const double z = apply_minv(m, sl, t, rs);
        v.x[row] = 0.0;
        v.r[row] = bv;
        v.z[row] = z;
        v.p[row] = z;
        d0 += bv * z;
        d1 += bv * bv;
    }
    const double t0 = block_sum(d0, sh);
    const double t1 = block_sum(d1, sh);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t0;
        v.partials[G + blockIdx.x] = t1;
    }
}

void launch_cg_init(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_cg_init, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

// x += alpha p ; r -= alpha q ; z = M^-1 r ; partial sums of r.z and r.r
__global__ __launch_bounds__(192) void k_cg_update(DeviceMatrix m, CgVectors v)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    if (v.s->done != 0) return;
    const int G = gridDim.x, t = threadIdx.x;
    const double alpha = v.s->alpha;
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const double pv = v.p[row], qv = v.q[row];
        v.x[row] += alpha * pv;
        const double rn = v.r[row] - alpha * qv;
        v.r[row] = rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double z = apply_minv(m, sl, t, rs);
        v.z[row] = z;
        d0 += rn * z;
        d1 += rn * rn;
    }
    const double t0 = block_sum(d0, sh);
    const double t1 = block_sum(d1, sh);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t0;
        v.partials[G + blockIdx.x] = t1;
    }
}

void launch_cg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_cg_update, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

// p = z + beta p over the owned (padded) rows, 16 bytes per lane
__global__ __launch_bounds__(256) void k_cg_direction(CgVectors v, int64_t n2)
{
    if (v.s->done != 0) return;
    const double beta = v.s->beta;
    const double2 *z = reinterpret_cast<const double2 *>(v.z);
    double2 *p = reinterpret_cast<double2 *>(v.p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 zv = z[i];
        double2 pv = p[i];
        pv.x = zv.x + beta * pv.x;
        pv.y = zv.y + beta * pv.y;
        p[i] = pv;
    }
}

void launch_cg_direction(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    const int64_t n2 = (int64_t)m.n_pad * 3;
    const int64_t blocks = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_cg_direction, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, v, n2);
}

// single workgroup: deterministic reduction of the per-workgroup partial sums, then the
// scalar recurrence step
__global__ __launch_bounds__(256) void k_cg_scalar(CgVectors v, int G, int do_reduce, int nsums, int phase,
                                                   double rtol)
{
    __shared__ double sh[4];
    CgScalars *s = v.s;
    if (phase != CG_PHASE_INIT && s->done != 0) return;
    if (do_reduce) {
        for (int a = 0; a < nsums; a++) {
            double t = 0.0;
            for (int i = threadIdx.x; i < G; i += blockDim.x) t += v.partials[(int64_t)a * G + i];
            const double tot = block_sum(t, sh);
            if (threadIdx.x == 0) s->red[a] = tot;
        }
    }
    if (threadIdx.x != 0) return;
    if (phase == CG_PHASE_INIT) {
        s->rz = s->red[0];
        s->bb = s->red[1];
        s->rr = s->red[1];
        s->tol2 = rtol > 0.0 ? rtol * rtol * s->red[1] : 0.0;
        s->alpha = 0.0;
        s->beta = 0.0;
        s->iters = 0;
        s->done = (s->red[1] == 0.0) ? 1 : 0;
    } else if (phase == CG_PHASE_ALPHA) {
        const double pq = s->red[0];
        if (!(pq > 0.0)) s->done = -1;
        else s->alpha = s->rz / pq;
    } else if (phase == CG_PHASE_BETA) {
        const double rzn = s->red[0], rr = s->red[1];
        s->rr = rr;
        const int it = s->iters + 1;
        s->iters = it;
        if (v.hist != nullptr && it <= v.hist_cap) v.hist[it - 1] = rr / s->bb;
        if (rr <= s->tol2) s->done = 1;
        else {
            s->beta = rzn / s->rz;
            s->rz = rzn;
        }
    }
}

void launch_cg_scalar(const DeviceMatrix &m, const CgVectors &v, bool reduce, int nsums, CgPhase phase,
                      double rtol, hipStream_t st)
{
    hipLaunchKernelGGL(k_cg_scalar, dim3(1), dim3(256), 0, st, v, slice_grid(m), reduce ? 1 : 0, nsums, (int)phase,
                       rtol);
}

__global__ void k_pack(const double *p, const int32_t *nodes, int32_t count, double *buf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)count * 6) return;
    buf[i] = p[(int64_t)nodes[i / 6] * 6 + i % 6];
}

void launch_pack(const double *p, const int32_t *send_nodes, int32_t count, double *sendbuf, hipStream_t st)
{
    const int64_t n = (int64_t)count * 6;
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, send_nodes, count, sendbuf);
}

} // namespace femshell
