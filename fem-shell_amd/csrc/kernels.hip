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
#include "device_common.hpp"

#include <type_traits>
#include "assemble_kernel.hpp"

#include <cstdlib>

namespace femshell {

static_assert(kSliceNodes == 32 && kSliceRows == 192, "kernels assume 32-node slices");

static int assemble_grid(const DeviceMatrix &m)
{
    static const int cap = [] {
        const char *e = getenv("FEMSHELL_ASM_GRID");
        return e ? atoi(e) : 2048;
    }();
    const int g = 8 * ((m.n_slices + 7) / 8);
    return g < cap ? g : cap;
}

int slice_grid(const DeviceMatrix &m)
{
    static const int cap = [] {
        const char *e = getenv("FEMSHELL_SLICE_GRID"); // tuning knob
        return e ? atoi(e) : 65536; // one slice per workgroup up to 2M node rows: best SpMV rate; more rows share workgroups
    }();
    const int g = 8 * ((m.n_slices + 7) / 8);
    return g < cap ? g : cap;
}

void launch_assemble(const DeviceMatrix &m, const MatConst &mc, hipStream_t st)
{
    const size_t lds = (size_t)m.lds_bytes;
    static const int variant = [] {
        const char *e = getenv("FEMSHELL_ASM_WAVES"); // tuning knob: waves per SIMD the kernel is compiled for
        return e ? atoi(e) : 2;
    }();
    const int g = assemble_grid(m);
    auto launch = [&](auto kernel) {
        if (lds > 64 * 1024) // beyond the default dynamic-LDS limit
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(g), dim3(256), lds, st, m, mc);
    };
    if (m.pipe) { // two workgroups per CU, b and b + G/2 on the same one: their roles complement each other
        static const int pipe_cap = [] { // two workgroups per CU of this device (FEMSHELL_ASM_PIPE_GRID overrides)
            const char *e = getenv("FEMSHELL_ASM_PIPE_GRID");
            if (e) return atoi(e);
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            return 8 * ((2 * cus + 7) / 8);
        }();
        const int gp = g < pipe_cap ? g : pipe_cap;
        auto launch_pipe = [&](auto kernel) {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kernel, dim3(gp), dim3(256), lds, st, m, mc);
        };
        if (m.n_lquad > 0) launch_pipe(k_assemble_pipe<0, true>);
        else launch_pipe(k_assemble_pipe<0, false>);
        return;
    }
    if (m.n_lquad > 0) {
        launch(k_assemble<2, 0, true>);
        return;
    }
    switch (variant) {
    case 1: launch(k_assemble<1, 0, false>); break;
    case 3: launch(k_assemble<3, 0, false>); break;
    case 4: launch(k_assemble<4, 0, false>); break;
    default: launch(k_assemble<2, 0, false>); break;
    }
}

// Constraint word of every assembly work item (DeviceMatrix::item_flags): the assembly kernel fetches it together
// with the item, one slice ahead, instead of chasing cols -> dmask[col] in its block phase.
// (In kernel statistics this kernel can show 15-27 ms per call at 4M triangles: it is the first kernel behind the uploads
// of femshell_set_mesh / femshell_set_dirichlet -- launched twice in a row it takes 17 ms and 78 us, and 115 us behind the
// uploads of femshell_set_loads alone (tools/lab/flags_prof.sh, idle_prof.sh).  Idle gaps of up to half a second in front
// of a kernel cost it nothing.)
__global__ __launch_bounds__(256) void k_item_flags(DeviceMatrix m)
{
    for (int s = blockIdx.x; s < m.n_slices; s += gridDim.x) {
        const int64_t base = m.slice_base[s];
        const int i0 = m.item_ptr[s], ni = m.item_ptr[s + 1] - i0;
        for (int it = threadIdx.x; it < ni; it += blockDim.x) {
            const uint32_t x = m.items[i0 + it].x;
            const int slot_in_slice = (int)(x & 0xffffu), chunk = (int)((x >> 16) & 0xffu), nchunks = (int)(x >> 24);
            uint32_t f = 0u;
            if (chunk == 0 && nchunks > 0) {
                const int64_t slot = base + slot_in_slice;
                const int row = s * kSliceNodes + (slot_in_slice & 31), col = m.cols[slot];
                const uint32_t valence = (uint32_t)(m.pair_ptr[slot + 1] - m.pair_ptr[slot]);
                f = (uint32_t)m.dmask[row] | ((uint32_t)m.dmask[col] << 6) | ((valence < 255u ? valence : 255u) << 12) |
                    (col == row ? 1u << 20 : 0u);
            }
            if (m.pipe) { // the pipelined kernel reads the word from the item itself: one load and 4 bytes per item less
                uint32_t *w = &const_cast<uint4 *>(m.items)[i0 + it].w;
                *w = (*w & ((1u << kPipeFlagShift) - 1u)) | (f << kPipeFlagShift);
            } else {
                m.item_flags[i0 + it] = f;
            }
        }
    }
}

void launch_item_flags(const DeviceMatrix &m, int64_t n_items, hipStream_t st)
{
    if (n_items == 0 || m.n_slices == 0) return;
    hipLaunchKernelGGL(k_item_flags, dim3(m.n_slices < 8192 ? m.n_slices : 8192), dim3(256), 0, st, m);
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
// Elements [0,n_ltri) are triangles (9 blocks each), the rest quads (16 blocks each).
__global__ __launch_bounds__(128) void k_element_matrices(DeviceMatrix m, MatConst mc, int first, int count, double *out)
{
    const bool quads = first >= m.n_ltri;
    const int nn = quads ? 4 : 3, nb = nn * nn;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)count * nb) return;
    const int e = (int)(t / nb), pr = (int)(t % nb), ia = pr / nn, ib = pr % nn;
    double rec[kRecDoublesQuad];
    bool ok;
    if (!quads) {
        const int32_t *c = m.tri + 3 * (int64_t)(first + e);
        double X[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int d = 0; d < 3; d++) X[3 * i + d] = m.xyz[3 * (int64_t)c[i] + d];
        ok = tri3_record(X, mc, rec);
    } else {
        const int32_t *c = m.quad + 4 * (int64_t)(first + e - m.n_ltri);
        double X[12];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int d = 0; d < 3; d++) X[3 * i + d] = m.xyz[3 * (int64_t)c[i] + d];
        ok = quad4_record(X, mc, rec);
    }
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    if (ok) block_add_rec<true>(rec, ia, ib, mc, acc);
    else report_status(m.status, kStatusDirect + first + e);
    const int N = 6 * nn;
    double *Ke = out + (int64_t)e * N * N;
#pragma unroll
    for (int al = 0; al < 6; al++)
#pragma unroll
        for (int be = 0; be < 6; be++) Ke[(nn * al + ia) * N + nn * be + ib] = acc[6 * al + be];
}

void launch_element_matrices(const DeviceMatrix &m, const MatConst &mc, int32_t first, int32_t count,
                             double *Ke_out, hipStream_t st)
{
    const int64_t n = (int64_t)count * (first >= m.n_ltri ? 16 : 9);
    hipLaunchKernelGGL(k_element_matrices, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, m, mc, first,
                       count, Ke_out);
}

// =====================================================================================
// Block-Jacobi setup: invert the 6x6 diagonal block of every owned node (Cholesky: the blocks are SPD)
// into the packed layout the CG vector kernels stream.
// =====================================================================================
__global__ void k_block_jacobi(DeviceMatrix m)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= m.n_pad) return;
    const int s = a / kSliceNodes, n = a % kSliceNodes;
    double *mi = m.minv + (int64_t)s * kMinvWords * kSliceNodes + n;
    if (a >= m.n_own) {
#pragma unroll
        for (int e = 0; e < kMinvWords; e++) mi[e * kSliceNodes] = 0.0;
        return;
    }
    const double2 *src = reinterpret_cast<const double2 *>(m.vals + m.slice_base[s] * 36);
    double A[6][6], B[6][6];
#pragma unroll
    for (int jp = 0; jp < 3; jp++)
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (m.diag_upper && 2 * jp + 1 < i) continue; // (never written: the mirror image below fills it in)
            const double2 v = src[(jp * 6 + i) * kSliceNodes + n];
            A[i][2 * jp] = v.x;
            A[i][2 * jp + 1] = v.y;
        }
    if (m.diag_upper) {
#pragma unroll
        for (int i = 1; i < 6; i++)
#pragma unroll
            for (int j = 0; j < i; j++) A[i][j] = A[j][i];
    }
    // Cholesky A = L L^T (the blocks are SPD), then A^-1 = L^-T L^-1: symmetric by construction and backward
    // stable without pivoting, which matters on sliver elements (block condition numbers around 1e9)
    double L[6][6], Li[6][6];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double d = A[c][c];
#pragma unroll
        for (int k = 0; k < c; k++) d -= L[c][k] * L[c][k];
        if (!(d > 0.0)) {
            ok = false;
            d = 1.0;
        }
        const double lcc = sqrt(d), il = 1.0 / lcc;
        L[c][c] = lcc;
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            double v = A[r][c];
#pragma unroll
            for (int k = 0; k < c; k++) v -= L[r][k] * L[c][k];
            L[r][c] = v * il;
        }
    }
    // Li = L^-1 (lower triangular) by forward substitution
#pragma unroll
    for (int c = 0; c < 6; c++) {
        Li[c][c] = 1.0 / L[c][c];
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            double v = 0.0;
#pragma unroll
            for (int k = c; k < r; k++) v -= L[r][k] * Li[k][c];
            Li[r][c] = v / L[r][r];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j2 = i; j2 < 6; j2++) {
            double v = 0.0;
#pragma unroll
            for (int k = j2; k < 6; k++) v += Li[k][i] * Li[k][j2];
            B[i][j2] = v;
        }
    if (!ok) {
        report_status(m.status, -(a + 1));
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j2 = i; j2 < 6; j2++) B[i][j2] = (i == j2) ? 1.0 : 0.0;
    }
    // the inverse of a symmetric block is symmetric: its upper triangle is stored (and applied), 21 words per node
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) mi[minv_word(i, j) * kSliceNodes] = B[i][j];
}

__global__ void k_status_flags(const int32_t *__restrict__ status, double *__restrict__ agree)
{
    const int32_t st = *status;
    agree[0] = st > 0 ? 1.0 : 0.0;
    agree[1] = st < 0 ? 1.0 : 0.0;
}

void launch_status_flags(const int32_t *status, double *agree, hipStream_t st)
{
    hipLaunchKernelGGL(k_status_flags, dim3(1), dim3(1), 0, st, status, agree);
}

void launch_block_jacobi(const DeviceMatrix &m, hipStream_t st)
{
    hipLaunchKernelGGL(k_block_jacobi, dim3((m.n_pad + 127) / 128), dim3(128), 0, st, m);
}

// =====================================================================================
// SpMV y = K x on the sliced block ELL layout, one lane per scalar row: lane t of a slice holds block row
// i = t / 32 of node n = t % 32, so that the words of K it needs are the t-th of every 192-word group and
// consecutive lanes read consecutive 16-byte words (1 KiB per wave instruction).  The x entries of the slice's
// block columns (48 bytes per slot and node) are staged through LDS once per slice -- the six lanes of a node sit
// in three different waves in this mapping -- and read back as broadcasts.  Optionally fuses the partial sums of
// x.y needed by CG (p.Ap).
// =====================================================================================
// kChunk block slots are handled together: all their K loads are issued back to back; the loads of the first
// chunk are issued before the x staging (spmv_load), so that their latency overlaps it.
// (kF32: the words come from a single-precision copy of the values in the same layout, v32 -- smoothing products of the
//  multigrid cycle only; the arithmetic stays FP64.  The chunk keeps the words AS LOADED and converts them where they are used:
//  until round 5 spmv_load converted to double on the spot, so the wave waited for its loads before it began to stage x -- the
//  float variant of the product took 16.3 us on the 584-slice level of the 4M hierarchy where the FP64 one took 11.4.)
template <int kChunk, bool kF32 = false> struct SpmvChunk {
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<kF32, v2f_, v2d_>::type Word;
    Word a[kChunk][3];
};
template <int kChunk, bool kF32 = false, bool kNT = true>
__device__ __forceinline__ void spmv_load(SpmvChunk<kChunk, kF32> &c, const double2 *__restrict__ v, int k0, int W,
                                          const float2 *__restrict__ v32 = nullptr)
{
    typedef typename SpmvChunk<kChunk, kF32>::Word Word;
#pragma unroll
    for (int q = 0; q < kChunk; q++) {
        if (k0 + q < W) {
            // an operator that is read once per launch and does not fit the caches: non-temporal loads leave them to x (kNT);
            // operators of a few ten megabytes -- the coarse levels of the cycle, multiplied again a few microseconds later --
            // are read with plain loads and found in the L2 / the Infinity Cache by the next product
            const Word *vv = kF32 ? reinterpret_cast<const Word *>(v32 + (size_t)(k0 + q) * 3 * kSliceRows)
                                  : reinterpret_cast<const Word *>(v + (size_t)(k0 + q) * 3 * kSliceRows);
            if (kNT) {
                c.a[q][0] = __builtin_nontemporal_load(vv);
                c.a[q][1] = __builtin_nontemporal_load(vv + kSliceRows);
                c.a[q][2] = __builtin_nontemporal_load(vv + 2 * kSliceRows);
            } else {
                c.a[q][0] = vv[0];
                c.a[q][1] = vv[kSliceRows];
                c.a[q][2] = vv[2 * kSliceRows];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 3; t++) c.a[q][t] = (Word){0, 0};
        }
    }
}
template <int kChunk, bool kF32>
__device__ __forceinline__ double spmv_fma(const SpmvChunk<kChunk, kF32> &c, const double2 *__restrict__ xs, int k0, int W, double acc)
{
#pragma unroll
    for (int q = 0; q < kChunk; q++) {
        if (k0 + q < W) {
            const double2 *xx = xs + (size_t)(k0 + q) * 3 * kSliceNodes;
            const double2 x0 = xx[0], x1 = xx[1], x2 = xx[2];
            acc += (double)c.a[q][0].x * x0.x;
            acc += (double)c.a[q][0].y * x0.y;
            acc += (double)c.a[q][1].x * x1.x;
            acc += (double)c.a[q][1].y * x1.y;
            acc += (double)c.a[q][2].x * x2.x;
            acc += (double)c.a[q][2].y * x2.y;
        }
    }
    return acc;
}

// (cheb: the epilogue of launch_spmv_cheb -- y is r_out, base_vec r_in with sign -1, x the direction d_in)
struct ChebEpilogue {
    double *d_out = nullptr; // nullptr: plain product
    double *xsol = nullptr;
    double a = 0.0, c = 0.0;
    // launch_spmv_axpy_keep: the product itself, K x without the base vector, is stored as well (as floats: prod_float)
    double *prod_out = nullptr;
    int prod_float = 0;
    // the FIRST step of a smoothing instead of a later one: d_out = c D^-1 r_out (no previous direction: d_in is the vector the
    // product multiplied, not a direction), x += d_out -- or x = d_out from a zero guess (start == 2)
    int start = 0;
};

template <int kChunk, bool kF32 = false, bool kNT = true>
__global__ __launch_bounds__(192) void k_spmv(DeviceMatrix m, const double *__restrict__ x,
                                              double *__restrict__ y, double *__restrict__ partials,
                                              const CgScalars *s, const int32_t *__restrict__ order, int count,
                                              const double *base_vec, double sign, int panel, ChebEpilogue cheb)
{
    extern __shared__ double2 xs_all[]; // panel slots x 32 nodes x 3 words: x of the slice's block columns
    __shared__ double sh[3];
    __shared__ double rs[kSliceRows]; // Chebyshev epilogue: the slice's new residual, node-major
    if (s != nullptr && s->done != 0) return;
    const int t = threadIdx.x;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *xs = xs_all + 3 * (t & 31);
    double dotv = 0.0;
    for (SliceWalk w(count); w.valid(); w.next()) {
        const int sl = order != nullptr ? order[w.s] : w.s;
        const int64_t base = m.slice_base[sl];
        const int W = m.slice_width[sl];
        double acc = 0.0;
        double2 xw = make_double2(0.0, 0.0);
        // slices wider than the LDS panel (restriction operators of coarse multigrid levels: a coarse node collects
        // from every fine node its basis function touches) go through it in several passes
        for (int p0 = 0; p0 == 0 || p0 < W; p0 += panel) {
            const int Wp = W - p0 < panel ? W - p0 : panel;
            const double2 *v = reinterpret_cast<const double2 *>(m.vals + base * 36) + (size_t)p0 * 3 * kSliceRows + t;
            const int32_t *cols = m.cols + base + (int64_t)p0 * kSliceNodes;
            const float2 *v32 = kF32 ? reinterpret_cast<const float2 *>(m.vals32 + base * 36) + (size_t)p0 * 3 * kSliceRows + t : nullptr;
            // (the float variant keeps two chunks in flight: the second chunk's words travel during the staging of x as well, every
            //  later chunk while its predecessor is multiplied -- 96 registers of words; the FP64 variant has room for one)
            SpmvChunk<kChunk, kF32> ch, ch2;
            spmv_load<kChunk, kF32, kNT>(ch, v, 0, Wp, v32);
            if (kF32 && kChunk < Wp) spmv_load<kChunk, kF32, kNT>(ch2, v, kChunk, Wp, v32);
            __syncthreads(); // the previous panel's readers are done with xs_all
            for (int e = t; e < Wp * kSliceNodes; e += kSliceRows) {
                const double2 *xv = x2 + 3 * (int64_t)cols[e];
                const double2 x0 = xv[0], x1 = xv[1], x2w = xv[2];
                xs_all[3 * e] = x0;
                xs_all[3 * e + 1] = x1;
                xs_all[3 * e + 2] = x2w;
            }
            __syncthreads();
            acc = spmv_fma<kChunk, kF32>(ch, xs, 0, Wp, acc);
            if (kF32) {
                // slots in ascending order as before: ch (0), ch2 (kChunk), ch (2 kChunk), ch2 (3 kChunk), ...
                for (int k0 = kChunk; k0 < Wp; k0 += 2 * kChunk) {
                    if (k0 + kChunk < Wp) spmv_load<kChunk, kF32, kNT>(ch, v, k0 + kChunk, Wp, v32);
                    acc = spmv_fma<kChunk, kF32>(ch2, xs, k0, Wp, acc);
                    if (k0 + kChunk < Wp) {
                        if (k0 + 2 * kChunk < Wp) spmv_load<kChunk, kF32, kNT>(ch2, v, k0 + 2 * kChunk, Wp, v32);
                        acc = spmv_fma<kChunk, kF32>(ch, xs, k0 + kChunk, Wp, acc);
                    }
                }
            } else {
                for (int k0 = kChunk; k0 < Wp; k0 += kChunk) {
                    spmv_load<kChunk, kF32, kNT>(ch, v, k0, Wp, v32);
                    acc = spmv_fma<kChunk, kF32>(ch, xs, k0, Wp, acc);
                }
            }
            // x[row] is in LDS during the first panel: slot 0 is the diagonal block, its column is the lane's own node
            if (p0 == 0 && partials != nullptr) xw = xs[t >> 6]; // word (t / 32) / 2 of the node's six entries
        }
        const int64_t row = (int64_t)sl * kSliceRows + (t & 31) * 6 + (t >> 5);
        // base_vec: y = base + sign * K x (residual b - K x, prolongation x + P x_c); may alias y
        const double yv = base_vec != nullptr ? base_vec[row] + sign * acc : acc;
        y[row] = yv;
        if (cheb.prod_out != nullptr) {
            if (cheb.prod_float) reinterpret_cast<float *>(cheb.prod_out)[row] = (float)acc;
            else cheb.prod_out[row] = acc;
        }
        if (cheb.d_out != nullptr) {
            // (MEASURED, round 4: fetching the epilogue's operands -- D^-1 row, d, x, base vector -- ahead of the product, in an
            //  instantiation of its own, costs 24 registers and a wave per SIMD there and was 2 % SLOWER on the 4M solves
            //  although the small levels are latency chains: profiles/r04_spmv_epilogue_prefetch_ab.txt.  Left as it is.)
            // d_out = a d_in + c D^-1 r_out on the lane's row; the six residual entries of its node sit in six lanes of
            // three waves (lane = node + 32 dof): exchanged through LDS
            const int n = t & 31, i = t >> 5;
            double mrow[6];
            if (m.minv32 != nullptr) {
                const float *mi = m.minv32 + (int64_t)sl * kMinvWords * kSliceNodes + n;
#pragma unroll
                for (int j = 0; j < 6; j++) mrow[j] = (double)mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
            } else {
                const double *mi = m.minv + (int64_t)sl * kMinvWords * kSliceNodes + n;
#pragma unroll
                for (int j = 0; j < 6; j++) mrow[j] = mi[minv_word(i < j ? i : j, i < j ? j : i) * kSliceNodes];
            }
            const double dv = cheb.start ? 0.0 : x[row], xv = cheb.start == 2 ? 0.0 : cheb.xsol[row];
            __syncthreads();
            rs[n * 6 + i] = yv;
            __syncthreads();
            double z = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) z += mrow[j] * rs[n * 6 + j];
            if (cheb.start) {
                // (the expression of k_cheb_start, so that the compiler contracts it the same way -- x = fma(c, z, x) -- and the
                //  fused start gives the bits of the separate pass)
                const double dn = cheb.c * z;
                cheb.d_out[row] = dn;
                cheb.xsol[row] = xv + dn;
            } else {
                const double dn = cheb.a * dv + cheb.c * z;
                cheb.d_out[row] = dn;
                cheb.xsol[row] = xv + dn;
            }
        }
        if (partials != nullptr) dotv += acc * (((t >> 5) & 1) ? xw.y : xw.x);
    }
    if (partials != nullptr) {
        const double tot = block_sum(dotv, sh);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

// =====================================================================================
// Symmetric storage (plan.hpp): of every off-diagonal pair of owned nodes only the block K_ac of the lower row a is
// stored.  Phase 1 (k_spmv_sym), one lane per NODE row: the lane streams the 36 words of each of its blocks once and
// uses them twice -- y_a += K_ac x_c for its own row and u = K_ac^T x_a for row c, written next to the slot (48
// bytes).  With a lane per node both products are lane-local: no reduction across lanes or waves.  Phase 2
// (k_sym_gather), one lane per scalar row: y_c += sum of the u of the blocks (a, c), in the fixed order of the plan's
// in-lists -- deterministic, no atomics.  Traffic on the 4M-triangle panel: 2.31 GB of blocks + 0.29 GB of u written
// and read once, against 4.03 GB of blocks with full storage.  Two refinements: of the symmetric diagonal block only the
// words of the upper triangle are read (2.12 GB of blocks), and a product whose row c lies in the lane's own slice waits
// in LDS for the end of the slice instead of going through HBM (plan.hpp loc_index / loc_list: 0.19 GB of u).
// The fused dot x.Kx of CG needs no second phase: x.Kx = sum_a x_a.(direct part of y_a) + sum over stored
// off-diagonal blocks of x_c.u.
// =====================================================================================
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
// (kNT: non-temporal loads -- an operator that is streamed once per launch and larger than the caches leaves them to the vectors;
//  kNT = false, plain loads: operators small enough for the 256 MB Infinity Cache to serve the NEXT product of the same cycle --
//  a level-1 operator of the 4M hierarchy is multiplied sixteen times per outer iteration)
template <bool kF32, bool kNT = true> __device__ __forceinline__ v2d load_word(const double2 *v, const float2 *v32, size_t off)
{
    if (kF32) {
        const v2f *p = reinterpret_cast<const v2f *>(v32) + off;
        const v2f w = kNT ? __builtin_nontemporal_load(p) : *p;
        v2d r;
        r.x = (double)w.x;
        r.y = (double)w.y;
        return r;
    }
    const v2d *p = reinterpret_cast<const v2d *>(v) + off;
    return kNT ? __builtin_nontemporal_load(p) : *p;
}

// the 18 words (jp, i) of block slot k: wd[jp * 6 + i] = columns 2jp, 2jp+1 of row i.  diag: only the words of the upper triangle
// are needed (slot 0 of a symmetric-storage row); the others stay unset
// kVal: 0 = FP64 values, 1 = the float copy (m.vals32)
template <int kVal, bool kDiag, bool kNT = true>
__device__ __forceinline__ void load_block_words(const double2 *v, const float2 *v32, int k, v2d wd[18])
{
#pragma unroll
    for (int e = 0; e < 18; e++)
        if (!kDiag || 2 * (e / 6) + 1 >= e % 6) wd[e] = load_word<(kVal == 1), kNT>(v, v32, ((size_t)k * 18 + e) * kSliceNodes);
}

// kVal: the blocks come from m.vals (0) or from m.vals32 (1: single precision, same layout); the arithmetic stays FP64
// (A bfloat16 copy -- 72 B per block -- was built and measured in round 5 and is gone again: the smoother's copy needs about 20
//  significant bits.  The residuals a cycle restricts are increments of products with that copy, and an error of 2^-8 ||A|| ||d||
//  in them is amplified by the coarse solves: the 4M-triangle panel did not converge at all, and the float copy rounded to 18 / 17 /
//  16 bits takes 129 / 186 / 394 iterations on the cylinder instead of 97: profiles/r05_smoother_significant_bits.txt,
//  FEMSHELL_AMG_SMOOTH_SIGBITS.)
// kVec (with kVal >= 1 only; DeviceMatrix::vec32): 1 = y and the transposed products are stored as floats, 2 = x is read as floats too
// (The float-storing variants compile to 182-194 registers, two waves per SIMD where the FP64 product has three.  MEASURED, round
//  4: held to three waves -- amdgpu_waves_per_eu(3, 3), 168 registers, five dwords spilled -- the 4M solves take the same time
//  within the run-to-run scatter of 1 %: profiles/r04_spmv_sym_waves_ab.txt, four alternating rounds.  Left to the compiler.)
template <int kVal, int kVec, bool kNT = true>
__global__ __launch_bounds__(64) void k_spmv_sym(DeviceMatrix m, const double *__restrict__ x, double *__restrict__ y,
                                                 double *__restrict__ partials, const CgScalars *s,
                                                 const int32_t *__restrict__ order, int count)
{
    if (s != nullptr && s->done != 0) return;
    const int lane = threadIdx.x, half = lane >> 5, n = lane & 31;
    double dotv = 0.0;
    extern __shared__ double2 lds_products[]; // [half][max_loc][3]: transposed products that stay inside a slice
    const bool has_local = m.loc_index != nullptr;
    double2 *lu = lds_products + (size_t)half * m.max_loc * 3;
    const int n_pairs = (count + 1) >> 1;
    for (SliceWalk w(n_pairs); w.valid(); w.next()) {
        const int q = 2 * w.s + half;
        const bool live = q < count;
        const int sl = live ? (order != nullptr ? order[q] : q) : 0;
        const int64_t base = m.slice_base[sl];
        const int W = live ? m.slice_width[sl] : 0;
        const int a = sl * kSliceNodes + n;
        double xa[6], ya[6];
        load_node6(x, a, kVec == 2, xa);
#pragma unroll
        for (int i = 0; i < 6; i++) ya[i] = 0.0;
        const double2 *v = reinterpret_cast<const double2 *>(m.vals + base * 36) + n;
        const float2 *v32 = kVal == 1 ? reinterpret_cast<const float2 *>(m.vals32 + base * 36) + n : nullptr;
        double2 *tb = reinterpret_cast<double2 *>(m.tbuf + base * 6);
        float2 *tbf = reinterpret_cast<float2 *>(m.tbuf) + base * 3; // (kVec >= 1: float (slot * 6 + j) of the same buffer)
        const uint8_t *li = has_local ? m.loc_index + base + n : nullptr;
        if (W > 0) {
            // slot 0 is the diagonal block K_aa, which is symmetric: only the 12 of its 18 words that hold the upper
            // triangle are read (each word is 512 contiguous bytes of the slice, so the other six never leave HBM);
            // element (i, j) below the diagonal is taken from (j, i).  Same order of the sum over j as in the loop
            // below, so a block whose halves mirror each other exactly (k_assemble's do) gives the same bits.
            v2d wd[18];
            load_block_words<kVal, true, kNT>(v, v32, 0, wd);
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const int r = j >= i ? i : j, cl = j >= i ? j : i; // (r, cl): the element of the upper triangle
                    const v2d kw = wd[(cl >> 1) * 6 + r];
                    ya[i] += ((cl & 1) ? kw.y : kw.x) * xa[j];
                }
        }
        for (int k = 1; k < W; k++) {
            const int c = m.cols[base + (int64_t)k * kSliceNodes + n];
            v2d wd[18];
            load_block_words<kVal, false, kNT>(v, v32, k, wd); // word (jp = e/6, i = e%6)
            double xc[6];
            load_node6(x, c, kVec == 2, xc);
            double u[6];
#pragma unroll
            for (int j = 0; j < 6; j++) u[j] = 0.0;
#pragma unroll
            for (int jp = 0; jp < 3; jp++)
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const v2d kw = wd[jp * 6 + i];
                    ya[i] += kw.x * xc[2 * jp];
                    ya[i] += kw.y * xc[2 * jp + 1];
                    u[2 * jp] += kw.x * xa[i];
                    u[2 * jp + 1] += kw.y * xa[i];
                }
            // the transpose acts on row c when c is another owned row (ghost columns belong to another rank,
            // padding slots point at the own row)
            if (c != a && c < m.n_pad) {
                const int local = has_local ? (int)li[(size_t)k * kSliceNodes] : 255;
                // (row c is a row of this slice: the product waits in LDS for the end of the slice, else next to the slot)
                if (kVec >= 1 && local == 255) {
                    float2 *t = tbf + ((size_t)k * kSliceNodes + n) * 3;
                    t[0] = make_float2((float)u[0], (float)u[1]);
                    t[1] = make_float2((float)u[2], (float)u[3]);
                    t[2] = make_float2((float)u[4], (float)u[5]);
                } else {
                    double2 *t = local != 255 ? lu + local * 3 : tb + ((size_t)k * kSliceNodes + n) * 3;
                    t[0] = make_double2(u[0], u[1]);
                    t[1] = make_double2(u[2], u[3]);
                    t[2] = make_double2(u[4], u[5]);
                }
                if (partials != nullptr)
                    dotv += xc[0] * u[0] + xc[1] * u[1] + xc[2] * u[2] + xc[3] * u[3] + xc[4] * u[4] + xc[5] * u[5];
            }
        }
        if (live && partials != nullptr) // (before the in-slice products join: x_c.u counted them above)
            dotv += xa[0] * ya[0] + xa[1] * ya[1] + xa[2] * ya[2] + xa[3] * ya[3] + xa[4] * ya[4] + xa[5] * ya[5];
        if (has_local) {
            // the transposed products of this slice's own rows, in the order of the in-list
            __syncthreads(); // (one wave per workgroup)
            const int Wi = live ? m.in_width[sl] : 0;
            const uint8_t *ll = m.loc_list + m.in_base[sl] + n;
            for (int k = 0; k < Wi; k++) {
                const int idx = ll[(size_t)k * kSliceNodes];
                if (idx != 255) {
                    const double2 t0 = lu[idx * 3], t1 = lu[idx * 3 + 1], t2 = lu[idx * 3 + 2];
                    ya[0] += t0.x; ya[1] += t0.y; ya[2] += t1.x; ya[3] += t1.y; ya[4] += t2.x; ya[5] += t2.y;
                }
            }
            __syncthreads(); // the next slice overwrites the products
        }
        if (live) {
            if (kVec >= 1) {
                float2 *yo = reinterpret_cast<float2 *>(y) + 3 * (int64_t)a;
                yo[0] = make_float2((float)ya[0], (float)ya[1]);
                yo[1] = make_float2((float)ya[2], (float)ya[3]);
                yo[2] = make_float2((float)ya[4], (float)ya[5]);
            } else {
                double2 *yo = reinterpret_cast<double2 *>(y) + 3 * (int64_t)a;
                yo[0] = make_double2(ya[0], ya[1]);
                yo[1] = make_double2(ya[2], ya[3]);
                yo[2] = make_double2(ya[4], ya[5]);
            }
        }
    }
    if (partials != nullptr) {
        const double tot = wave_sum(dotv);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

// (kQ32: y and the transposed products were stored as floats by a smoothing product, DeviceMatrix::vec32; out is FP64)
template <bool kQ32>
__global__ __launch_bounds__(192) void k_sym_gather(DeviceMatrix m, const double *y, double *out, const double *base_vec, double sign,
                                                    const CgScalars *s)
{
    if (s != nullptr && s->done != 0) return;
    const int t = threadIdx.x, n = t / 6, j = t % 6;
    const float *yf = reinterpret_cast<const float *>(y), *tf = reinterpret_cast<const float *>(m.tbuf);
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int Wi = m.in_width[sl];
        const int64_t ib = m.in_base[sl];
        const int64_t row = (int64_t)sl * kSliceRows + t;
        double acc = kQ32 ? (double)yf[row] : y[row];
        const double bv = base_vec != nullptr ? base_vec[row] : 0.0;
        // the slot indices of the first entries together, then their products together (as in k_cg_update): one entry at a
        // time is two dependent memory round trips per entry; the order of the additions is the same
        int32_t slot4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) slot4[k] = (k < Wi) ? m.gat_slots[ib + (int64_t)k * kSliceNodes + n] : -1;
        double t4[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            t4[k] = slot4[k] < 0 ? 0.0 : (kQ32 ? (double)tf[(int64_t)slot4[k] * 6 + j] : m.tbuf[(int64_t)slot4[k] * 6 + j]);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (slot4[k] >= 0) acc += t4[k];
        for (int k = 4; k < Wi; k++) {
            const int32_t slot = m.gat_slots[ib + (int64_t)k * kSliceNodes + n];
            if (slot >= 0) acc += kQ32 ? (double)tf[(int64_t)slot * 6 + j] : m.tbuf[(int64_t)slot * 6 + j];
        }
        out[row] = base_vec != nullptr ? bv + sign * acc : acc;
    }
}

static bool operator_fits_the_caches(const DeviceMatrix &m);

static void spmv_sym_phase1(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                            const int32_t *order, int count, int grid, hipStream_t st, bool f32 = false)
{
    const size_t lds = m.loc_index != nullptr ? (size_t)2 * m.max_loc * 48 : 0;
    // operators whose single-precision values fit the Infinity Cache are read with plain loads: the next smoothing product of the
    // same visit finds them there (operator_fits_the_caches, below)
    if (f32 && m.vals32 != nullptr && m.vec32 == 2 && operator_fits_the_caches(m)) {
        hipLaunchKernelGGL((k_spmv_sym<1, 2, false>), dim3(grid), dim3(64), lds, st, m, x, y, partials, s, order, count);
        return;
    }
    if (f32 && m.vals32 != nullptr) {
        if (m.vec32 == 2) hipLaunchKernelGGL((k_spmv_sym<1, 2>), dim3(grid), dim3(64), lds, st, m, x, y, partials, s, order, count);
        else if (m.vec32 == 1) hipLaunchKernelGGL((k_spmv_sym<1, 1>), dim3(grid), dim3(64), lds, st, m, x, y, partials, s, order, count);
        else hipLaunchKernelGGL((k_spmv_sym<1, 0>), dim3(grid), dim3(64), lds, st, m, x, y, partials, s, order, count);
    } else {
        hipLaunchKernelGGL((k_spmv_sym<0, 0>), dim3(grid), dim3(64), lds, st, m, x, y, partials, s, order, count);
    }
}

__global__ __launch_bounds__(256) void k_to_f32(const double *__restrict__ src, float *__restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}

void launch_to_f32(const double *src, float *dst, int64_t n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_to_f32, dim3(4096), dim3(256), 0, st, src, dst, n);
}

// experiment knob (FEMSHELL_AMG_SMOOTH_SIGBITS): the float copy rounded to `sig` significant bits in place (round to nearest even on
// the bit pattern) -- how many bits does the smoother's copy of a level operator need?
__global__ __launch_bounds__(256) void k_round_sig(float *__restrict__ v, int64_t n, int sig)
{
    const int drop = 24 - sig;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t b = __float_as_uint(v[i]);
        const uint32_t half = (1u << (drop - 1)) - 1u;
        v[i] = __uint_as_float(((b + half + ((b >> drop) & 1u)) >> drop) << drop);
    }
}
void launch_round_sig(float *v, int64_t n, int sig, hipStream_t st)
{
    if (n > 0 && sig > 0 && sig < 24) hipLaunchKernelGGL(k_round_sig, dim3(4096), dim3(256), 0, st, v, n, sig);
}

// one lane per node (device_common.hpp; FEMSHELL_NODE_KERNELS=0: the kernels above)
static bool node_kernels_on()
{
    static const bool on = [] {
        const char *e = getenv("FEMSHELL_NODE_KERNELS");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
static int node_grid_of(const DeviceMatrix &m)
{
    const int g = 8 * ((node_pairs(m.n_slices) + 7) / 8), cap = slice_grid(m);
    return g < cap ? g : cap;
}

template <bool kQ32>
__global__ __launch_bounds__(64) void k_sym_gather_node(DeviceMatrix m, const double *y, double *out, const double *base_vec, double sign,
                                                        const CgScalars *s)
{
    if (s != nullptr && s->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double acc[6], bv[6];
        load_node6(y, node, kQ32, acc);
        if (base_vec != nullptr) load_node6(base_vec, node, false, bv);
        node_gather<kQ32>(m, sl, n, acc);
        if (base_vec != nullptr) {
#pragma unroll
            for (int j = 0; j < 6; j++) acc[j] = bv[j] + sign * acc[j];
        }
        store_node6(out, node, false, acc);
    }
}

void launch_sym_gather(const DeviceMatrix &m, double *y, const double *base_vec, double sign, const CgScalars *s, hipStream_t st,
                       bool q32, double *out)
{
    if (out == nullptr) out = y;
    if (node_kernels_on()) {
        if (q32) hipLaunchKernelGGL(k_sym_gather_node<true>, dim3(node_grid_of(m)), dim3(64), 0, st, m, y, out, base_vec, sign, s);
        else hipLaunchKernelGGL(k_sym_gather_node<false>, dim3(node_grid_of(m)), dim3(64), 0, st, m, y, out, base_vec, sign, s);
        return;
    }
    if (q32) hipLaunchKernelGGL(k_sym_gather<true>, dim3(slice_grid(m)), dim3(192), 0, st, m, y, out, base_vec, sign, s);
    else hipLaunchKernelGGL(k_sym_gather<false>, dim3(slice_grid(m)), dim3(192), 0, st, m, y, out, base_vec, sign, s);
}

// Double-double residual with symmetric storage: a lane per scalar row walks the blocks of its own row (row i of the
// block) and the blocks of its in-list (column i of the block, the transpose); used once per refinement pass.
__global__ __launch_bounds__(192) void k_residual_dd_sym(DeviceMatrix m, const double *__restrict__ x, const double *__restrict__ b,
                                                         double *__restrict__ r);

// Residual r = b - K x with the products and the row sums carried in double-double (error-free TwoProduct by FMA,
// TwoSum accumulation): on the thin-shell systems ||K|| ||x|| exceeds ||b|| by seven to nine orders of magnitude, so a
// residual evaluated in plain FP64 is rounding noise at 1e-7 ||b|| and restarting CG from it makes the answer worse.
// This kernel feeds the residual replacement of the multigrid-preconditioned solve (amg_solve.cpp).  Same data
// movement as k_spmv (HBM-bound at 0.2 flop/B; the five-fold arithmetic stays far below the FP64 ridge).
struct DD {
    double hi, lo;
};
// (with the default -ffp-contract=fast the compiler fuses acc.hi + a*x and a*x - bb into FMAs -- HIP's __dmul_rn /
// __dadd_rn are plain operators and `#pragma clang fp contract(off)` did not prevent it either; the product is
// therefore issued through inline assembly -- and the error terms below, which assume
// s = fl(acc.hi + fl(a x)), would be those of a different sum: measured, the "double-double" residual was then no
// better than the FP64 one)
__device__ __forceinline__ void dd_fma_acc(DD &acc, double a, double x)
{
    double p; // fl(a x) as an opaque instruction: neither pragmas nor the _rn intrinsics stop the backend from fusing
    asm("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(a), "v"(x));
    const double e = __fma_rn(a, x, -p);          // a*x = p + e exactly
    const double s = __dadd_rn(acc.hi, p);
    const double bb = __dsub_rn(s, acc.hi);
    const double err = __dadd_rn(__dsub_rn(acc.hi, __dsub_rn(s, bb)), __dsub_rn(p, bb)); // acc.hi + p = s + err exactly
    acc.hi = s;
    acc.lo = __dadd_rn(acc.lo, __dadd_rn(err, e));
}

__global__ __launch_bounds__(192) void k_residual_dd(DeviceMatrix m, const double *__restrict__ x, const double *__restrict__ b,
                                                     double *__restrict__ r)
{
    extern __shared__ double2 xs_all[];
    const int t = threadIdx.x;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t base = m.slice_base[sl];
        const int W = m.slice_width[sl];
        const double2 *v = reinterpret_cast<const double2 *>(m.vals + base * 36) + t;
        __syncthreads();
        for (int e = t; e < W * kSliceNodes; e += kSliceRows) {
            const double2 *xv = x2 + 3 * (int64_t)m.cols[base + e];
            xs_all[3 * e] = xv[0];
            xs_all[3 * e + 1] = xv[1];
            xs_all[3 * e + 2] = xv[2];
        }
        __syncthreads();
        const double2 *xs = xs_all + 3 * (t & 31);
        DD acc{0.0, 0.0};
        for (int k0 = 0; k0 < W; k0 += 4) {
            SpmvChunk<4> ch;
            spmv_load<4>(ch, v, k0, W);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (k0 + q < W) {
                    const double2 *xx = xs + (size_t)(k0 + q) * 3 * kSliceNodes;
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const double2 xw = xx[u];
                        dd_fma_acc(acc, ch.a[q][u].x, xw.x);
                        dd_fma_acc(acc, ch.a[q][u].y, xw.y);
                    }
                }
            }
        }
        const int64_t row = (int64_t)sl * kSliceRows + (t & 31) * 6 + (t >> 5);
        // r = b - (hi + lo), the difference b - hi taken exactly
        const double bv = b[row];
        const double s = __dsub_rn(bv, acc.hi);
        const double bb = __dsub_rn(s, bv);
        const double err = __dadd_rn(__dsub_rn(bv, __dsub_rn(s, bb)), __dsub_rn(-acc.hi, bb));
        r[row] = __dadd_rn(s, __dsub_rn(err, acc.lo));
    }
}

__global__ __launch_bounds__(192) void k_residual_dd_sym(DeviceMatrix m, const double *__restrict__ x, const double *__restrict__ b,
                                                         double *__restrict__ r)
{
    const int t = threadIdx.x, n = t / 6, i = t % 6;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t base = m.slice_base[sl];
        const int W = m.slice_width[sl];
        const int a = sl * kSliceNodes + n;
        DD acc{0.0, 0.0};
        for (int k = 0; k < W; k++) {
            const int c = (k == 0) ? a : m.cols[base + (int64_t)k * kSliceNodes + n];
            const double *blk = m.vals + base * 36 + (int64_t)k * 36 * kSliceNodes; // [(jp*6 + i)*32 + n]*2 + jj
            const double *xc = x + 6 * (int64_t)c;
            if (k == 0 && m.diag_upper) {
                // the diagonal block holds its upper triangle only: (i, j), j < i, is (j, i) -- same order of the sum over j
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const int r = j >= i ? i : j, cl = j >= i ? j : i;
                    dd_fma_acc(acc, blk[((size_t)((cl >> 1) * 6 + r) * kSliceNodes + n) * 2 + (cl & 1)], xc[j]);
                }
                continue;
            }
#pragma unroll
            for (int jp = 0; jp < 3; jp++) {
                const double *wd = blk + ((size_t)(jp * 6 + i) * kSliceNodes + n) * 2;
                dd_fma_acc(acc, wd[0], xc[2 * jp]);
                dd_fma_acc(acc, wd[1], xc[2 * jp + 1]);
            }
        }
        const int Wi = m.in_width[sl];
        const int64_t ib = m.in_base[sl];
        for (int k = 0; k < Wi; k++) {
            const int32_t slot = m.in_slots[ib + (int64_t)k * kSliceNodes + n];
            if (slot < 0) continue;
            const int src = m.in_rows[ib + (int64_t)k * kSliceNodes + n];
            const int ns = slot & 31;
            const double *blk = m.vals + (int64_t)(slot - ns) * 36; // the (slice, k) group of 32 blocks the slot sits in
            const double *xs = x + 6 * (int64_t)src;
            // column i of the block: entries K[i'][i], word (jp = i/2, i'), component i & 1
#pragma unroll
            for (int ip = 0; ip < 6; ip++)
                dd_fma_acc(acc, blk[((size_t)((i >> 1) * 6 + ip) * kSliceNodes + ns) * 2 + (i & 1)], xs[ip]);
        }
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const double bv = b[row];
        const double sdd = __dsub_rn(bv, acc.hi);
        const double bb = __dsub_rn(sdd, bv);
        const double err = __dadd_rn(__dsub_rn(bv, __dsub_rn(sdd, bb)), __dsub_rn(-acc.hi, bb));
        r[row] = __dadd_rn(sdd, __dsub_rn(err, acc.lo));
    }
}

void launch_residual_dd(const DeviceMatrix &m, const double *x, const double *b, double *r, hipStream_t st)
{
    if (m.symmetric) {
        hipLaunchKernelGGL(k_residual_dd_sym, dim3(slice_grid(m)), dim3(192), 0, st, m, x, b, r);
        return;
    }
    const size_t lds = (size_t)m.max_slice_width * kSliceNodes * 3 * sizeof(double2);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_residual_dd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_residual_dd, dim3(slice_grid(m)), dim3(192), lds, st, m, x, b, r);
}

// Is the operator of a size at which the next product of the same cycle finds it in the 256 MB Infinity Cache -- and at which that
// pays?  Upper bound of the bytes of its values from the padded slot count; plain loads between FEMSHELL_SPMV_CACHED_MIN_MB and
// FEMSHELL_SPMV_CACHED_MB (defaults 128 and 300; CACHED_MB=0: every operator is streamed with non-temporal loads as in rounds 1-4).
// MEASURED, round 5, alternating on one box (symmetric-storage smoothing products only): the level-1 operator of the 4M hierarchies
// (197 MB of floats, sixteen smoothing products per outer iteration) with plain loads: panel 0.6454 -> 0.6372 s, cylinder 0.6031 ->
// 0.5945 s; level 0 as well (1.2 GB): 0.6002 s -- it does not fit, and its lines evict the vectors; the 250k-triangle roof, whose
// level 0 is 72 MB: 0.0603 -> 0.0614 s -- an operator that small is gone from the L2s anyway and costs the vectors their place.
static bool operator_fits_the_caches(const DeviceMatrix &m)
{
    static const double max_mb = [] {
        const char *e = getenv("FEMSHELL_SPMV_CACHED_MB");
        return e ? atof(e) : 300.0;
    }();
    static const double min_mb = [] {
        const char *e = getenv("FEMSHELL_SPMV_CACHED_MIN_MB");
        return e ? atof(e) : 128.0;
    }();
    const double bytes_per_value = m.vals32 != nullptr ? 4.0 : 8.0;
    const double mb = (double)m.n_slices * m.max_slice_width * kSliceNodes * 36.0 * bytes_per_value * 1e-6;
    return mb <= max_mb && mb >= min_mb;
}

constexpr int kSpmvPanel = 64; // block slots of x staged in LDS at a time by k_spmv

static void spmv_dispatch(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                          const int32_t *order, int count, int grid, hipStream_t st, const double *base_vec = nullptr,
                          double sign = 1.0, ChebEpilogue cheb = ChebEpilogue())
{
    static const int chunk = [] {
        const char *e = getenv("FEMSHELL_SPMV_CHUNK"); // tuning knob: block slots loaded together
        return e ? atoi(e) : 8;
    }();
    const dim3 g(grid), b(192);
    // x of the block columns, at most kSpmvPanel slots at a time (96 KiB of the CU's 160)
    const int panel = m.max_slice_width < kSpmvPanel ? (m.max_slice_width > 0 ? m.max_slice_width : 1) : kSpmvPanel;
    const size_t lds = (size_t)panel * kSliceNodes * 3 * sizeof(double2);
    auto launch = [&](auto kernel) {
        if (lds > 64 * 1024) // beyond the default dynamic-LDS limit (slices wider than 42 blocks)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, g, b, lds, st, m, x, y, partials, s, order, count, base_vec, sign, panel, cheb);
    };
    // (MEASURED, round 5: plain loads here -- the latency-bound products of the small levels -- LOSE: with them the 4M panel keeps
    //  0.5 % of the 1.3 % the symmetric-storage products gain, and the 250k-triangle roof, all of whose operators fit, takes 0.0609 s
    //  instead of 0.0594 s.  FEMSHELL_SPMV_CACHED_FULL=1 selects them for A/B runs.)
    static const bool cached_full = getenv("FEMSHELL_SPMV_CACHED_FULL") && atoi(getenv("FEMSHELL_SPMV_CACHED_FULL")) == 1;
    const bool cached = cached_full && operator_fits_the_caches(m);
    if (m.vals32 != nullptr) { // a product of the multigrid cycle on a single-precision copy of the values (amg_solve.cpp)
        if (cached) launch(k_spmv<8, true, false>);
        else launch(k_spmv<8, true>);
        return;
    }
    if (cached && chunk == 8) {
        launch(k_spmv<8, false, false>);
        return;
    }
    switch (chunk) {
    case 1: launch(k_spmv<1>); break;
    case 2: launch(k_spmv<2>); break;
    case 4: launch(k_spmv<4>); break;
    default: launch(k_spmv<8>); break;
    }
}

void launch_spmv(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                 hipStream_t st)
{
    if (m.symmetric) {
        spmv_sym_phase1(m, x, y, partials, s, nullptr, m.n_slices, slice_grid(m), st);
        launch_sym_gather(m, y, nullptr, 1.0, s, st);
        return;
    }
    spmv_dispatch(m, x, y, partials, s, nullptr, m.n_slices, slice_grid(m), st);
}

void launch_spmv_cheb(const DeviceMatrix &m, const double *d_in, const double *r_in, double *r_out, double *d_out, double *x,
                      double a, double c, const CgScalars *s, hipStream_t st)
{
    ChebEpilogue e;
    e.d_out = d_out;
    e.xsol = x;
    e.a = a;
    e.c = c;
    spmv_dispatch(m, d_in, r_out, nullptr, s, nullptr, m.n_slices, slice_grid(m), st, r_in, -1.0, e);
}

void launch_spmv_start(const DeviceMatrix &m, const double *v_in, const double *r_in, double *r_out, double *d_out, double *x,
                       double inv_theta, const CgScalars *s, hipStream_t st)
{
    ChebEpilogue e;
    e.d_out = d_out;
    e.xsol = x;
    e.c = inv_theta;
    e.start = 1;
    spmv_dispatch(m, v_in, r_out, nullptr, s, nullptr, m.n_slices, slice_grid(m), st, r_in, -1.0, e);
}

static bool spmv_node_applies(const DeviceMatrix &m);
static void launch_spmv_node(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign, double *prod_out,
                             bool prod_float, const CgScalars *s, hipStream_t st);

void launch_spmv_axpy(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign,
                      const CgScalars *s, hipStream_t st)
{
    if (spmv_node_applies(m)) {
        launch_spmv_node(m, x, y, base_vec, sign, nullptr, false, s, st);
        return;
    }
    if (m.symmetric) { // (base_vec must not be y here: phase 1 overwrites y with the direct part)
        spmv_sym_phase1(m, x, y, nullptr, s, nullptr, m.n_slices, slice_grid(m), st);
        launch_sym_gather(m, y, base_vec, sign, s, st);
        return;
    }
    spmv_dispatch(m, x, y, nullptr, s, nullptr, m.n_slices, slice_grid(m), st, base_vec, sign);
}

// Full-storage product with one lane per NODE row (round 5): y = base_vec + sign * K x for operators with few blocks per row -- the
// prolongations, 2.5 blocks per fine node.  k_spmv stages the x of a slice's block columns through LDS behind two barriers per
// slice, which pays for wide rows; with two to four slots a slice is two barriers around a handful of loads, and the prolongation
// onto level 0 of the 4M-triangle panel moved its 1.14 GB at 3.6 TB/s (312 us).  Here a lane streams the words of its blocks as
// k_spmv_sym does and reads the six entries of each column node straight from the caches.  Per row the sum runs over the slots
// in ascending order and inside a block over the columns in ascending order, as in k_spmv: same bits.
template <bool kF32>
__global__ __launch_bounds__(64) void k_spmv_node(DeviceMatrix m, const double *__restrict__ x, double *y, const CgScalars *s,
                                                  const double *base_vec, double sign, double *prod_out, int prod_float)
{
    if (s != nullptr && s->done != 0) return;
    const int half = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t base = m.slice_base[sl];
        const int W = m.slice_width[sl];
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        const double2 *v = reinterpret_cast<const double2 *>(m.vals + base * 36) + n;
        const float2 *v32 = kF32 ? reinterpret_cast<const float2 *>(m.vals32 + base * 36) + n : nullptr;
        double ya[6], bv[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ya[i] = 0.0;
        if (base_vec != nullptr) load_node6(base_vec, node, false, bv);
        for (int k = 0; k < W; k++) {
            const int c = m.cols[base + (int64_t)k * kSliceNodes + n];
            v2d wd[18];
            load_block_words<(kF32 ? 1 : 0), false>(v, v32, k, wd);
            double xc[6];
            load_node6(x, c, false, xc);
#pragma unroll
            for (int jp = 0; jp < 3; jp++)
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const v2d kw = wd[jp * 6 + i];
                    ya[i] += kw.x * xc[2 * jp];
                    ya[i] += kw.y * xc[2 * jp + 1];
                }
        }
        if (prod_out != nullptr) store_node6(prod_out, node, prod_float != 0, ya);
        if (base_vec != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) ya[i] = bv[i] + sign * ya[i];
        }
        store_node6(y, node, false, ya);
    }
}

// rows narrower than this go through k_spmv_node (FEMSHELL_SPMV_NODE_WIDTH; 0 = never)
static int spmv_node_width()
{
    static const int w = [] {
        const char *e = getenv("FEMSHELL_SPMV_NODE_WIDTH");
        return e ? atoi(e) : 8;
    }();
    return w;
}

static bool spmv_node_applies(const DeviceMatrix &m) { return !m.symmetric && m.max_slice_width > 0 && m.max_slice_width <= spmv_node_width(); }

static void launch_spmv_node(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign, double *prod_out,
                             bool prod_float, const CgScalars *s, hipStream_t st)
{
    const dim3 g(node_grid_of(m)), b(64);
    if (m.vals32 != nullptr) hipLaunchKernelGGL(k_spmv_node<true>, g, b, 0, st, m, x, y, s, base_vec, sign, prod_out, prod_float ? 1 : 0);
    else hipLaunchKernelGGL(k_spmv_node<false>, g, b, 0, st, m, x, y, s, base_vec, sign, prod_out, prod_float ? 1 : 0);
}

void launch_spmv_axpy_keep(const DeviceMatrix &m, const double *x, double *y, const double *base_vec, double sign, double *prod_out,
                           bool prod_float, const CgScalars *s, hipStream_t st)
{
    if (spmv_node_applies(m)) {
        launch_spmv_node(m, x, y, base_vec, sign, prod_out, prod_float, s, st);
        return;
    }
    ChebEpilogue e;
    e.prod_out = prod_out;
    e.prod_float = prod_float ? 1 : 0;
    spmv_dispatch(m, x, y, nullptr, s, nullptr, m.n_slices, slice_grid(m), st, base_vec, sign, e);
}

// a full-storage product with any of the epilogues above over the slices order[begin, begin + count) only: the interior / boundary
// halves of a product whose halo exchange runs beside the interior half (row-partitioned multigrid levels, amg_solve.cpp)
void launch_spmv_epilogue_span(const DeviceMatrix &m, const double *x, double *y, const SpmvEpilogue &e, const int32_t *order, int begin,
                               int count, const CgScalars *s, hipStream_t st)
{
    if (count <= 0) return;
    ChebEpilogue c;
    c.d_out = e.d_out;
    c.xsol = e.xsol;
    c.a = e.a;
    c.c = e.c;
    c.start = e.start;
    c.prod_out = e.prod_out;
    c.prod_float = e.prod_float ? 1 : 0;
    const int g = 8 * ((count + 7) / 8), cap = slice_grid(m);
    spmv_dispatch(m, x, y, nullptr, s, order + begin, count, g < cap ? g : cap, st, e.base_vec, e.sign, c);
}

int span_grid(const DeviceMatrix &m, int count)
{
    const int g = 8 * ((count + 7) / 8), cap = slice_grid(m);
    return g < cap ? g : cap;
}

int launch_spmv_span(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s,
                     const int32_t *order, int begin, int count, int partial_offset, hipStream_t st)
{
    if (count <= 0) return 0;
    const int grid = span_grid(m, count);
    if (m.symmetric) { // phase 1 only: the caller runs launch_sym_gather once all spans are through
        // (m.vals32 set: a smoothing product of the multigrid cycle on the single-precision copy of the values)
        spmv_sym_phase1(m, x, y, partials != nullptr ? partials + partial_offset : nullptr, s, order + begin, count, grid, st,
                        m.vals32 != nullptr);
        return grid;
    }
    spmv_dispatch(m, x, y, partials != nullptr ? partials + partial_offset : nullptr, s, order + begin, count, grid, st);
    return grid;
}

// =====================================================================================
// CG vector kernels (one lane per scalar row, one workgroup per slice)
// =====================================================================================

__global__ __launch_bounds__(192) void k_cg_init(DeviceMatrix m, CgVectors v, int restart)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    const int G = gridDim.x, t = threadIdx.x;
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv(m, sl, t);
        const double bv = restart ? v.b[row] - v.q[row] : v.b[row]; // the residual to start from
        __syncthreads();
        rs[t] = bv;
        __syncthreads();
        const double z = apply_minv(mr, t, rs);
        if (!restart) v.x[row] = 0.0;
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

void launch_cg_init(const DeviceMatrix &m, const CgVectors &v, bool restart, hipStream_t st)
{
    hipLaunchKernelGGL(k_cg_init, dim3(slice_grid(m)), dim3(192), 0, st, m, v, restart ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_copy(const double2 *src, double2 *dst, int64_t n2)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

void launch_copy_x_to_p(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    const int64_t n2 = (int64_t)m.n_pad * 3;
    const int64_t blocks = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                       reinterpret_cast<const double2 *>(v.x), reinterpret_cast<double2 *>(v.p), n2);
}

// x += alpha p ; r -= alpha q ; z = M^-1 r ; partial sums of r.z and r.r
// (gather: symmetric storage, q holds the direct part of K p only; the row adds the transposed products of its
// in-list here instead of in a k_sym_gather pass of its own -- one read and one write of q and a launch less)
template <bool kGather>
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
        const MinvRow mr = load_minv(m, sl, t);
        const double pv = v.p[row], xv = v.x[row], rv = v.r[row];
        double qv = v.q[row];
        if (kGather) {
            const int Wi = m.in_width[sl], n = t / 6, j = t % 6;
            const int64_t ib = m.in_base[sl];
            // the slot indices of the first entries together, then their products together: one entry at a time is two
            // dependent memory round trips per entry (same order of the additions either way)
            int32_t slot4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) slot4[k] = (k < Wi) ? m.gat_slots[ib + (int64_t)k * kSliceNodes + n] : -1;
            double t4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) t4[k] = (slot4[k] >= 0) ? m.tbuf[(int64_t)slot4[k] * 6 + j] : 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (slot4[k] >= 0) qv += t4[k];
            for (int k = 4; k < Wi; k++) {
                const int32_t slot = m.gat_slots[ib + (int64_t)k * kSliceNodes + n];
                if (slot >= 0) qv += m.tbuf[(int64_t)slot * 6 + j];
            }
        }
        v.x[row] = xv + alpha * pv;
        const double rn = rv - alpha * qv;
        v.r[row] = rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double z = apply_minv(mr, t, rs);
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

// the same with one lane per node: no LDS, no barrier (launched with the grid of the per-slice kernels: the scalar step
// reduces slice_grid(m) partial sums per array)
template <bool kGather>
__global__ __launch_bounds__(64) void k_cg_update_node(DeviceMatrix m, CgVectors v)
{
    if (v.s->done != 0) return;
    const int G = gridDim.x, half = threadIdx.x >> 5, n = threadIdx.x & 31;
    const double alpha = v.s->alpha;
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(node_pairs(m.n_slices)); w.valid(); w.next()) {
        const int sl = 2 * w.s + half;
        if (sl >= m.n_slices) continue;
        const int64_t node = (int64_t)sl * kSliceNodes + n;
        double mv[kMinvWords], pv[6], xv[6], rv[6], qv[6], z[6];
        node_minv(m, sl, n, false, mv);
        load_node6(v.p, node, false, pv);
        load_node6(v.x, node, false, xv);
        load_node6(v.r, node, false, rv);
        load_node6(v.q, node, false, qv);
        if (kGather) node_gather<false>(m, sl, n, qv);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            xv[j] = xv[j] + alpha * pv[j];
            rv[j] = rv[j] - alpha * qv[j];
        }
        store_node6(v.x, node, false, xv);
        store_node6(v.r, node, false, rv);
        node_minv_apply(mv, rv, z);
        store_node6(v.z, node, false, z);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            d0 += rv[j] * z[j];
            d1 += rv[j] * rv[j];
        }
    }
    const double t0 = wave_sum(d0), t1 = wave_sum(d1);
    if (threadIdx.x == 0) {
        v.partials[blockIdx.x] = t0;
        v.partials[G + blockIdx.x] = t1;
    }
}

void launch_cg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st, bool gather)
{
    if (node_kernels_on()) {
        if (gather) hipLaunchKernelGGL(k_cg_update_node<true>, dim3(slice_grid(m)), dim3(64), 0, st, m, v);
        else hipLaunchKernelGGL(k_cg_update_node<false>, dim3(slice_grid(m)), dim3(64), 0, st, m, v);
        return;
    }
    if (gather) hipLaunchKernelGGL(k_cg_update<true>, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
    else hipLaunchKernelGGL(k_cg_update<false>, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

void launch_spmv_direct(const DeviceMatrix &m, const double *x, double *y, double *partials, const CgScalars *s, hipStream_t st,
                        bool single_precision_values)
{
    spmv_sym_phase1(m, x, y, partials, s, nullptr, m.n_slices, slice_grid(m), st, single_precision_values);
}

// ---- single-reduction recurrence (multi-rank solves): see kernels.hpp
__global__ __launch_bounds__(192) void k_cgcg_init(DeviceMatrix m, CgVectors v)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    const int G = gridDim.x, t = threadIdx.x;
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv(m, sl, t);
        const double bv = v.b[row];
        __syncthreads();
        rs[t] = bv;
        __syncthreads();
        const double z = apply_minv(mr, t, rs);
        v.x[row] = 0.0;
        v.r[row] = bv;
        v.z[row] = z;
        v.p[row] = 0.0;
        v.sv[row] = 0.0;
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

void launch_cgcg_init(const DeviceMatrix &m, const CgVectors &v, hipStream_t st)
{
    hipLaunchKernelGGL(k_cgcg_init, dim3(slice_grid(m)), dim3(192), 0, st, m, v);
}

// (kGather: as in k_cg_update.  step >= 0: the scalar step of the previous iteration is done here, by every workgroup
// for itself and published by workgroup 0 -- a launch less between the all-reduce and the next product; the arithmetic
// is that of cg_scalar_phase(CG_PHASE_FUSED_STEP))
template <bool kGather>
__global__ __launch_bounds__(192) void k_cgcg_update(DeviceMatrix m, CgVectors v, int step)
{
    __shared__ double rs[kSliceRows];
    __shared__ double sh[3];
    CgScalars *s = v.s;
    if (s->done != 0) return; // set by an earlier launch: the same in every workgroup
    const int G = gridDim.x, t = threadIdx.x;
    double alpha, beta;
    if (step >= 0) {
        const int par = step & 1;
        const double rz_old = s->ring_rz[par], alpha_old = s->ring_alpha[par];
        const double rzn = s->red[0], rr = s->red[1], zaz = s->red[2];
        int done = 0;
        alpha = 0.0;
        beta = 0.0;
        if (rr <= s->tol2) done = 1;
        else {
            beta = rzn / rz_old;
            const double denom = zaz - beta * rzn / alpha_old;
            if (!(denom > 0.0)) done = -1;
            else alpha = rzn / denom;
        }
        if (blockIdx.x == 0 && t == 0) {
            s->rr = rr;
            const int it = s->iters + 1;
            s->iters = it;
            if (v.hist != nullptr && it <= v.hist_cap) v.hist[it - 1] = rr / s->bb;
            if (done == 0) {
                s->beta = beta;
                s->alpha = alpha;
                s->rz = rzn;
                s->ring_rz[par ^ 1] = rzn;
                s->ring_alpha[par ^ 1] = alpha;
            }
            s->done = done; // read by the later launches only: this one has taken its decision from red[]
        }
        if (done != 0) return;
    } else {
        alpha = s->alpha;
        beta = s->beta;
    }
    double d0 = 0.0, d1 = 0.0;
    for (SliceWalk w(m.n_slices); w.valid(); w.next()) {
        const int sl = w.s;
        const int64_t row = (int64_t)sl * kSliceRows + t;
        const MinvRow mr = load_minv(m, sl, t);
        const double uv = v.z[row], pv = v.p[row], sv = v.sv[row], xv = v.x[row], rv = v.r[row];
        double wv = v.q[row];
        if (kGather) {
            const int Wi = m.in_width[sl], n = t / 6, j = t % 6;
            const int64_t ib = m.in_base[sl];
            for (int k = 0; k < Wi; k++) {
                const int32_t slot = m.gat_slots[ib + (int64_t)k * kSliceNodes + n];
                if (slot >= 0) wv += m.tbuf[(int64_t)slot * 6 + j];
            }
        }
        const double pn = uv + beta * pv, sn = wv + beta * sv;
        v.p[row] = pn;
        v.sv[row] = sn;
        v.x[row] = xv + alpha * pn;
        const double rn = rv - alpha * sn;
        v.r[row] = rn;
        __syncthreads();
        rs[t] = rn;
        __syncthreads();
        const double z = apply_minv(mr, t, rs);
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

void launch_cgcg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st, int step, bool gather)
{
    if (gather) hipLaunchKernelGGL(k_cgcg_update<true>, dim3(slice_grid(m)), dim3(192), 0, st, m, v, step);
    else hipLaunchKernelGGL(k_cgcg_update<false>, dim3(slice_grid(m)), dim3(192), 0, st, m, v, step);
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

// Deterministic two-stage reduction of the per-workgroup partial sums followed by the scalar recurrence
// step, in one launch: kReduceGroups workgroups each sum a contiguous chunk (fixed order), publish their
// result and take a ticket; the workgroup that draws the last ticket adds the stage-1 sums in index order
// and updates alpha / beta / the convergence flag.  Hand-off without fences (an agent-scope release writes
// back the XCD's L2, an acquire invalidates the CU's L1: microseconds each): the stage-1 sums are stored and
// loaded with sc1 (agent-scope relaxed atomics, served by L2), the storing lane waits for its stores (vmcnt(0))
// before its agent-scope ticket add, and the last arriver loads after that add has returned and a workgroup
// barrier (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire).  Which workgroup is last
// does not change the result: the final sum runs over the stage-1 sums in a fixed tree.
constexpr int kReduceGroups = 64;

constexpr double kRefineDrop = 1.0e-4; // residual reduction asked of a refinement pass (relative to its right-hand side) ...
// ... unless the pass knows better.  Its correction e is the displacement error of the iterate x it started from, and what the
// pass leaves of that error is about ||e|| / ||x|| times the drop of its residual (cg_amg).  With ||x||^2 at hand (pass_xx) the
// pass therefore stops when that estimate, with the ||e_k|| of the correction so far, is kRefineTarget of the tolerance: on the
// 4M panel (||e|| / ||x|| = 4.6e-8 behind a first phase to 1e-8) at a drop of 4.3e-4 instead of 1e-4 -- six iterations of 104 --
// and where the first phase left a larger error, deeper than 1e-4 instead of a second pass.  Never less than two digits, never
// more than six.
constexpr double kRefineTarget = 0.2, kRefineDropMin = 1.0e-6, kRefineDropMax = 1.0e-2;

__device__ __forceinline__ void cg_scalar_phase(const CgVectors &v, int phase, double rtol)
{
    CgScalars *s = v.s;
    if (phase == CG_PHASE_INIT) {
        s->rz = s->red[0];
        s->bb = s->red[1];
        s->rr = s->red[1];
        s->tol2 = rtol > 0.0 ? rtol * rtol * s->red[1] : 0.0;
        s->alpha = 0.0;
        s->beta = 0.0;
        s->iters = 0;
        s->done = (s->red[1] == 0.0) ? 1 : 0;
    } else if (phase == CG_PHASE_RESTART) {
        // explicit residual r = b - K x: red[0] = r.z, red[1] = r.r
        s->rz = s->red[0];
        s->rr = s->red[1];
        s->done = (s->red[1] <= s->tol2) ? 1 : 0;
    } else if (phase == CG_PHASE_ALPHA) {
        const double pq = s->red[0];
        if (!(pq > 0.0)) s->done = -1;
        else s->alpha = s->rz / pq;
    } else if (phase == CG_PHASE_FUSED_INIT) {
        // red = (r.z, r.r = b.b, z.Az) of the initial residual
        s->rz = s->red[0];
        s->bb = s->red[1];
        s->rr = s->red[1];
        s->tol2 = rtol > 0.0 ? rtol * rtol * s->red[1] : 0.0;
        s->beta = 0.0;
        s->alpha = 0.0;
        s->iters = 0;
        s->done = (s->red[1] == 0.0) ? 1 : 0;
        if (s->done == 0) {
            if (!(s->red[2] > 0.0)) s->done = -1;
            else s->alpha = s->red[0] / s->red[2];
        }
        s->ring_rz[0] = s->rz;
        s->ring_alpha[0] = s->alpha;
    } else if (phase == CG_PHASE_FUSED_STEP) {
        // red = (r.z, r.r, z.Az) of the new residual: beta = rz'/rz, alpha = rz' / (z.Az - beta rz'/alpha)
        const double rzn = s->red[0], rr = s->red[1], zaz = s->red[2];
        s->rr = rr;
        const int it = s->iters + 1;
        s->iters = it;
        if (v.hist != nullptr && it <= v.hist_cap) v.hist[it - 1] = rr / s->bb;
        if (rr <= s->tol2) s->done = 1;
        else {
            const double beta = rzn / s->rz;
            const double denom = zaz - beta * rzn / s->alpha;
            if (!(denom > 0.0)) s->done = -1;
            else {
                s->beta = beta;
                s->alpha = rzn / denom;
                s->rz = rzn;
            }
        }
    } else if (phase == CG_PHASE_FLEX_INIT) {
        s->pass_xx = 0.0;
        s->pass_rhs_rr = 0.0;
        s->bb = s->red[0];
        s->rr = s->red[0];
        s->tol2 = rtol > 0.0 ? rtol * rtol * s->red[0] : 0.0;
        s->alpha = 0.0;
        s->beta = 0.0;
        s->rz = 0.0;
        s->iters = 0;
        s->done = (s->red[0] == 0.0) ? 1 : 0;
    } else if (phase == CG_PHASE_FLEX_RESTART) {
        // a refinement pass starts: red[0] = r.r of the new right-hand side; b.b, the tolerance (relative to the
        // original right-hand side), the iteration count and the history carry on
        s->rr = s->red[0];
        s->alpha = 0.0;
        s->beta = 0.0;
        s->rz = 0.0;
        // (cg_amg passes rtol = 0: a pass stops on the drop of its own right-hand side alone; with rtol > 0 it would also
        //  stop at the tolerance of the solve as a whole)
        const double tol2_solve = rtol > 0.0 ? rtol * rtol * s->bb : 0.0;
        s->done = (s->red[0] <= tol2_solve) ? 1 : 0;
        // the correction equation needs four digits, not the full tolerance again: its solution is added to an iterate
        // whose error it reduces by that factor (2e-10 -> 1e-13 and below on the shell systems), and every further
        // digit costs iterations of the whole method
        s->tol2 = fmax(tol2_solve, kRefineDrop * kRefineDrop * s->red[0]);
        s->pass_rhs_rr = s->red[0];
    } else if (phase == CG_PHASE_FLEX_WARM) {
        s->rr = s->red[0];
        s->done = (s->red[0] <= s->tol2) ? 1 : 0;
    } else if (phase == CG_PHASE_FLEX_RZ0) {
        s->rz = s->red[0];
        if (!(s->red[0] > 0.0)) s->done = -1; // the preconditioner is not positive definite
    } else if (phase == CG_PHASE_FLEX_CONV) {
        const double rr = s->red[0];
        s->rr = rr;
        const int it = s->iters + 1;
        s->iters = it;
        if (v.hist != nullptr && it <= v.hist_cap) v.hist[it - 1] = rr / s->bb;
        if (s->pass_xx > 0.0 && s->pass_rhs_rr > 0.0 && s->red[1] > 0.0) {
            // a refinement pass with the adaptive rule: red[1] = e.e of the correction so far
            const double drop = kRefineTarget * s->pass_rtol * sqrt(s->pass_xx / s->red[1]);
            const double d = fmin(fmax(drop, kRefineDropMin), kRefineDropMax);
            s->tol2 = d * d * s->pass_rhs_rr;
        }
        if (rr <= s->tol2) s->done = 1;
    } else if (phase == CG_PHASE_FLEX_BETA) {
        const double rzn = s->red[0], zq = s->red[1];
        if (!(rzn > 0.0)) s->done = -1;
        else {
            s->beta = -s->alpha * zq / s->rz;
            s->rz = rzn;
        }
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

__global__ __launch_bounds__(256) void k_cg_scalar(CgVectors v, int G, int do_reduce, int nsums, int phase,
                                                   double rtol, int len3, int gate_phase)
{
    __shared__ double sh[4];
    CgScalars *s = v.s;
    // gate_phase: the phase this launch belongs to (a reduce-only launch in front of an all-reduce carries
    // phase NONE but must not be skipped when it serves an INIT / RESTART step on a finished solve)
    if (gate_phase != CG_PHASE_INIT && gate_phase != CG_PHASE_RESTART && gate_phase != CG_PHASE_FUSED_INIT &&
        gate_phase != CG_PHASE_FLEX_INIT && gate_phase != CG_PHASE_FLEX_RESTART && gate_phase != CG_PHASE_FLEX_WARM && s->done != 0)
        return; // same decision in every workgroup
    if (!do_reduce) {
        if (blockIdx.x == 0 && threadIdx.x == 0) cg_scalar_phase(v, phase, rtol);
        return;
    }
    const int nwg = gridDim.x;
    for (int a = 0; a < nsums; a++) {
        const int len = (a == 2) ? len3 : G; // the third array (single-reduction CG: the SpMV's) has its own length
        const int chunk = (len + nwg - 1) / nwg;
        const int lo = blockIdx.x * chunk, hi = min(len, lo + chunk);
        const double *pa = v.partials + (int64_t)a * G;
        double acc = 0.0;
        const int B = blockDim.x;
        for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * B) {
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = (i0 + q * B < hi) ? pa[i0 + q * B] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += t[q];
        }
        const double tot = block_sum(acc, sh);
        if (threadIdx.x == 0) __hip_atomic_store(&s->stage[a][blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int last_flag;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t ticket = __hip_atomic_fetch_add(&s->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (ticket == (uint32_t)nwg - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!last_flag) return;
    for (int a = 0; a < nsums; a++) {
        double part = 0.0;
        if ((int)threadIdx.x < nwg) part = __hip_atomic_load(&s->stage[a][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double tot = block_sum(part, sh);
        if (threadIdx.x == 0) s->red[a] = tot;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&s->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
        cg_scalar_phase(v, phase, rtol);
    }
}

void launch_cg_scalar(const DeviceMatrix &m, const CgVectors &v, bool reduce, int nsums, CgPhase phase,
                      double rtol, hipStream_t st, int n_partials, int len3, int gate_phase)
{
    const int G = n_partials > 0 ? n_partials : slice_grid(m);
    const int groups = reduce ? (G >= 4096 ? kReduceGroups : 1) : 1;
    hipLaunchKernelGGL(k_cg_scalar, dim3(groups), dim3(256), 0, st, v, G, reduce ? 1 : 0, nsums, (int)phase, rtol, len3,
                       gate_phase < 0 ? (int)phase : gate_phase);
}

__global__ void k_pack(const double *p, const int32_t *nodes, int32_t count, int32_t width, double *buf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)count * width) return;
    buf[i] = p[(int64_t)nodes[i / width] * width + i % width];
}

void launch_pack(const double *p, const int32_t *send_nodes, int32_t count, double *sendbuf, hipStream_t st, int width)
{
    const int64_t n = (int64_t)count * width;
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, send_nodes, count, (int32_t)width, sendbuf);
}

} // namespace femshell
