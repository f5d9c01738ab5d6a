// amg_dense.hip -- dense inverse of the coarsest operator of the multigrid hierarchy on the matrix cores (gfx950).
//
// Why it exists: the K cycle visits level l 2^(l-1) times per outer iteration, and on the last levels every visit is a
// chain of launches that each cost their 4.5 us whatever they compute -- on the 4M-triangle panel the 1231-node level and
// the 81-node level below it took 14 % of the solve for 0.04 % of the data.  Ending the hierarchy at the first level
// of at most ~1400 nodes (8400 dofs) with an exact solve replaces that subtree by one dense matrix-vector product per visit.
// The inverse of such an operator is 0.4 TFLOP of FP64 work, far out of reach of the host code that inverts the small
// coarsest operators (amg_setup.cpp dense_inverse) -- and it is the one place of this library where a dense,
// GEMM-shaped contraction of real size exists: it runs on v_mfma_f64_16x16x4_f64.
//
// Method: symmetric block sweep operator (Goodnight 1979) on the lower triangle, 64 x 64 tiles.  Sweeping block K,
//     B = A_KK^-1,   A_RK <- A_RK B,   A_RR <- A_RR - A_RK B A_KR,   A_KK <- -B          (R = all other rows),
// keeps the matrix symmetric, so only tiles (i >= j) are stored and updated; after all blocks the matrix is -A^-1.
// Per step: k_dense_pivot (one workgroup: scalar sweeps of the 64 x 64 pivot block in LDS), k_dense_panels (C = column
// block K gathered from the lower triangle, W = C B on the matrix cores), k_dense_update (every lower tile:
// A_ij -= W_i C_j^T, 2 x 64^3 flops per tile on the matrix cores, operands staged through LDS with a row stride of 66
// doubles: the 32 lanes of a ds_read_b64 group hit 32 different bank pairs).  n^3 flops in all, the lower triangle read
// and written once per step.
// Semi-definite operators: a scalar pivot that has lost eleven digits against the original diagonal entry is dropped
// (zero row and column of the inverse), a clearly negative one is a failure -- the rules of the host code.
#include "amg_device.hpp"
#include "device_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

namespace femshell {

namespace {

constexpr int kNB = 64;   // tile edge
constexpr int kLdp = 66;  // LDS row stride of a staged 64 x 64 operand (doubles)
constexpr int kHalf = 32, kLdh = 34; // k_dense_update stages its operands in two halves of K
typedef double v4d __attribute__((ext_vector_type(4)));

// D (n_pad x n_pad, row-major, zero-initialised) <- the blocks of the host BSR matrix
__global__ __launch_bounds__(256) void k_dense_scatter(const int64_t *__restrict__ ptr, const int32_t *__restrict__ col,
                                                       const double *__restrict__ val, int32_t nr, double *__restrict__ D, int64_t ld)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // (block, entry)
    const int64_t q = e / 36;
    if (q >= ptr[nr]) return;
    int lo = 0, hi = nr - 1; // block row of q by bisection
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int r = (int)(e % 36) / 6, c = (int)(e % 36) % 6;
    D[(6ll * lo + r) * ld + 6ll * col[q] + c] = val[e];
}

// lower triangle <- mean of the two triangles (the Galerkin products are symmetric up to rounding), unit diagonal on the
// padding rows, copy of the diagonal for the pivot test
__global__ __launch_bounds__(256) void k_dense_symmetrize(double *__restrict__ D, int64_t ld, int n, int n_pad, double *__restrict__ diag0)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(e / n_pad), c = (int)(e % n_pad);
    if (r >= n_pad || c > r) return;
    if (r >= n) {
        D[(int64_t)r * ld + c] = r == c ? 1.0 : 0.0;
        if (r == c) diag0[r] = 1.0;
        return;
    }
    const double v = 0.5 * (D[(int64_t)r * ld + c] + D[(int64_t)c * ld + r]);
    D[(int64_t)r * ld + c] = v;
    if (r == c) diag0[r] = v;
}

// B = (pivot block K)^-1; status[0] = 1 on a clearly negative pivot, status[1] counts the dropped directions.  One
// workgroup; thread (bi, bj) keeps the 4 x 4 sub-block (4 bi .., 4 bj ..) of the 64 x 64 block in registers and the block is
// swept four pivots at a time: the four rows of a step (= its four columns: the block stays symmetric) travel through a
// double-buffered LDS panel, every thread sweeps the 4 x 4 pivot sub-block itself (scalar symmetric sweeps with the pivot
// test) and applies the rank-4 update to its registers: 16 steps with one barrier each.  (Swept in LDS one pivot at a time
// with three barriers per pivot the kernel took as long as the trailing update of the whole matrix, 111 us per step.)
__global__ __launch_bounds__(256) void k_dense_pivot(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ diag0,
                                                     double *__restrict__ B, int32_t *status)
{
    __shared__ double rowbuf[2][4][kNB];
    __shared__ double a0[kNB];
    const int tid = threadIdx.x, k0 = K * kNB, bi = tid >> 4, bj = tid & 15;
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 4 * bi + a, c = 4 * bj + b;
            s[a][b] = r >= c ? D[(int64_t)(k0 + r) * ld + k0 + c] : D[(int64_t)(k0 + c) * ld + k0 + r];
        }
    if (tid < kNB) a0[tid] = diag0[k0 + tid];
    int n_dead = 0, n_failed = 0;
    for (int g = 0; g < kNB / 4; g++) {
        const int cur = g & 1;
        if (bi == g) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) rowbuf[cur][k][4 * bj + b] = s[k][b];
        }
        __syncthreads();
        // the 4 x 4 pivot sub-block, swept to minus its inverse on the live directions (every thread does the same)
        double m[4][4];
        bool dead[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 4; l++) m[k][l] = rowbuf[cur][k][4 * g + l];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double d = m[k][k], a = a0[4 * g + k];
            const bool failed = !(a > 0.0) || d < -1e-4 * fabs(a);
            dead[k] = failed || d <= 1e-11 * a;
            n_failed += failed ? 1 : 0;
            n_dead += dead[k] ? 1 : 0;
            const double inv = dead[k] ? 0.0 : 1.0 / d;
            double col[4];
#pragma unroll
            for (int l = 0; l < 4; l++) col[l] = m[l][k];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i == k || j == k) continue;
                    m[i][j] -= col[i] * col[j] * inv; // (dead: inv = 0, nothing moves)
                }
#pragma unroll
            for (int l = 0; l < 4; l++) {
                if (l == k) continue;
                const double v = col[l] * inv; // dead: zero row and column
                m[l][k] = v;
                m[k][l] = v;
            }
            m[k][k] = -inv;
        }
        // rows of the step as they were before it, for the thread's rows and columns: R[k][.] with R = rows 4g .. 4g+3
        double ri[4][4], rj[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ri[k][q] = rowbuf[cur][k][4 * bi + q];
                rj[k][q] = rowbuf[cur][k][4 * bj + q];
            }
        // block sweep with B4 = -m:  S_ij <- S_ij - R_i^T B4 R_j;  rows of the step <- B4 R_j;  columns <- (B4 R_i)^T;  pivot block <- m
        double ti[4][4]; // ti[a][l] = sum_k ri[k][a] B4[k][l]
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int l = 0; l < 4; l++) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) v -= ri[k][a] * m[k][l];
                ti[a][l] = v;
            }
        if (bi == g && bj == g) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) s[a][b] = m[a][b];
        } else if (bi == g) { // row panel: B4 R_j
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; l++) v -= m[a][l] * rj[l][b];
                    s[a][b] = v;
                }
        } else if (bj == g) { // column panel: (B4 R_i)^T = R_i^T B4
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) s[a][b] = ti[a][b];
        } else {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = s[a][b];
#pragma unroll
                    for (int l = 0; l < 4; l++) v -= ti[a][l] * rj[l][b];
                    s[a][b] = v;
                }
        }
    }
    if (tid == 0 && n_dead) {
        if (n_failed) status[0] = 1;
        atomicAdd(&status[1], n_dead);
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) B[(4 * bi + a) * kNB + 4 * bj + b] = -s[a][b]; // the sweeps leave -inverse
}

template <int kDepth, int kStride>
__device__ __forceinline__ void quadrant_xyt(const double *Xs, const double *Ys, int r0, int c0, int lane, v4d acc[2][2])
{
    const int rr = lane & 15, kq = lane >> 4;
#pragma unroll 4
    for (int k = 0; k < kDepth; k += 4) {
        double a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            a[t] = Xs[(r0 + 16 * t + rr) * kStride + k + kq];
            b[t] = Ys[(c0 + 16 * t + rr) * kStride + k + kq];
        }
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj < 2; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
}

// C_i = A(i, K) gathered from the lower triangle (tile (i, K) below the pivot block, tile (K, i) transposed above it),
// W_i = C_i B.  One workgroup per block row; block row K itself is skipped (its tile becomes -B).
__global__ __launch_bounds__(256) void k_dense_panels(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                      double *__restrict__ Cp, double *__restrict__ Wp)
{
    extern __shared__ double lds_dense[]; // two staged operands (67.6 KB: beyond the static limit)
    double *Cs = lds_dense, *Bs = lds_dense + kNB * kLdp;
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i == K) return;
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int r = e / kNB, c = e % kNB;
        // (above the pivot block the read is a transposed one: column-wise in memory, once per step and block row)
        const double v = i > K ? D[(int64_t)(i * kNB + r) * ld + K * kNB + c] : D[(int64_t)(K * kNB + c) * ld + i * kNB + r];
        Cs[r * kLdp + c] = v;
        Bs[r * kLdp + c] = B[e]; // symmetric: B^T = B, so W = C B = C (B^T)^T has the X Y^T form
        Cp[(int64_t)(i * kNB + r) * kNB + c] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    quadrant_xyt<kNB, kLdp>(Cs, Bs, r0, c0, lane, acc);
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int r = r0 + 16 * ti + (lane >> 4) + 4 * g, c = c0 + 16 * tj + (lane & 15);
                Wp[(int64_t)(i * kNB + r) * kNB + c] = acc[ti][tj][g];
            }
}

// one workgroup per lower tile (i >= j): the sweep of block K
__global__ __launch_bounds__(256, 4) void k_dense_update(double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                      const double *__restrict__ Cp, const double *__restrict__ Wp)
{
    extern __shared__ double lds_dense[]; // 64 x 66 doubles: the transposed copy of the row-K tiles; else two 64 x 34 halves
    double *Ws = lds_dense, *Cs = lds_dense + kNB * kLdh;
    const int tid = threadIdx.x;
    // linear tile index -> (i, j), i >= j
    const int t = blockIdx.x;
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    while (i * (i + 1) / 2 > t) i--;
    const int j = t - i * (i + 1) / 2;
    double *tile = D + (int64_t)(i * kNB) * ld + j * kNB;
    if (i == K && j == K) {
        for (int e = tid; e < kNB * kNB; e += 256) tile[(int64_t)(e / kNB) * ld + e % kNB] = -B[e];
        return;
    }
    if (j == K) { // below the pivot block: A_iK <- W_i
        for (int e = tid; e < kNB * kNB; e += 256) tile[(int64_t)(e / kNB) * ld + e % kNB] = Wp[(int64_t)(i * kNB) * kNB + e];
        return;
    }
    if (i == K) { // left of the pivot block: A_Kj <- W_j^T
        for (int e = tid; e < kNB * kNB; e += 256) Ws[(e / kNB) * kLdp + e % kNB] = Wp[(int64_t)(j * kNB) * kNB + e];
        __syncthreads();
        for (int e = tid; e < kNB * kNB; e += 256) tile[(int64_t)(e / kNB) * ld + e % kNB] = Ws[(e % kNB) * kLdp + e / kNB];
        return;
    }
    const int wave = tid >> 6, lane = tid & 63, r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
    // the tile's own values travel while the operands are staged
    v4d acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int r = r0 + 16 * ti + (lane >> 4) + 4 * g, c = c0 + 16 * tj + (lane & 15);
                acc[ti][tj][g] = -tile[(int64_t)r * ld + c]; // accumulate W C^T - A, store its negative
            }
    // K in two halves of 32 (row stride 34 doubles, again 2 mod 32: conflict-free): 35 KB of LDS per workgroup, four
    // workgroups per CU instead of two with whole operands
    const double *wsrc = Wp + (int64_t)(i * kNB) * kNB, *csrc = Cp + (int64_t)(j * kNB) * kNB;
    for (int h = 0; h < 2; h++) {
        if (h) __syncthreads(); // the first half's readers are through
        for (int e = tid; e < kNB * kHalf / 2; e += 256) { // 16-byte words: 16 per row and half
            const int r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
            const double2 wv = *reinterpret_cast<const double2 *>(wsrc + r * kNB + h * kHalf + k);
            const double2 cv = *reinterpret_cast<const double2 *>(csrc + r * kNB + h * kHalf + k);
            *reinterpret_cast<double2 *>(Ws + r * kLdh + k) = wv;
            *reinterpret_cast<double2 *>(Cs + r * kLdh + k) = cv;
        }
        __syncthreads();
        quadrant_xyt<kHalf, kLdh>(Ws, Cs, r0, c0, lane, acc);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int r = r0 + 16 * ti + (lane >> 4) + 4 * g, c = c0 + 16 * tj + (lane & 15);
                tile[(int64_t)r * ld + c] = -acc[ti][tj][g];
            }
}

// out (n x ldo, both triangles) <- -(lower triangle of D)
template <class T> __global__ __launch_bounds__(256) void k_dense_finish(const double *__restrict__ D, int64_t ld, int n, T *__restrict__ out, int64_t ldo)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(e / ldo), c = (int)(e % ldo);
    if (r >= n) return;
    double v = 0.0;
    if (c < n) v = -(r >= c ? D[(int64_t)r * ld + c] : D[(int64_t)c * ld + r]);
    out[(int64_t)r * ldo + c] = (T)v;
}

// y = Ainv b, one wave per row (rows are streamed with consecutive lanes on consecutive 16-byte words, four of them in
// flight per lane); rows [n, n_pad6) of y are set to zero
template <class T> struct Pair;
template <> struct Pair<double> { typedef double2 type; };
template <> struct Pair<float> { typedef float2 type; };
template <class T>
__global__ __launch_bounds__(256) void k_dense_gemv_big(const T *__restrict__ A, int64_t lda, const double *__restrict__ b,
                                                        double *__restrict__ y, int n, int n_pad6, const CgScalars *gate)
{
    if (gate != nullptr && gate->done != 0) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad6) return;
    double acc = 0.0;
    if (row < n) {
        typedef typename Pair<T>::type P2;
        const P2 *a = reinterpret_cast<const P2 *>(A + (int64_t)row * lda);
        const double2 *bb = reinterpret_cast<const double2 *>(b);
        const int n2 = (n + 1) / 2; // lda is even and the padding column is zero; b is padded with zeros too
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = lane;
        for (; j + 192 < n2; j += 256) {
            const P2 a0 = a[j], a1 = a[j + 64], a2 = a[j + 128], a3 = a[j + 192];
            const double2 b0 = bb[j], b1 = bb[j + 64], b2 = bb[j + 128], b3 = bb[j + 192];
            s0 += (double)a0.x * b0.x + (double)a0.y * b0.y;
            s1 += (double)a1.x * b1.x + (double)a1.y * b1.y;
            s2 += (double)a2.x * b2.x + (double)a2.y * b2.y;
            s3 += (double)a3.x * b3.x + (double)a3.y * b3.y;
        }
        for (; j < n2; j += 64) {
            const P2 a0 = a[j];
            const double2 b0 = bb[j];
            s0 += (double)a0.x * b0.x + (double)a0.y * b0.y;
        }
        acc = wave_sum((s0 + s1) + (s2 + s3));
    }
    if (lane == 0) y[row] = acc;
}

} // namespace

int amg_dense_inverse_device(femshell_ctx *c, const Bsr &A, bool single_precision, DevBuf<double> *inv64, DevBuf<float> *inv32,
                             int64_t *lda_out, AmgDenseStats *stats)
{
    hipStream_t st = c->stream;
    const int n = 6 * A.nr, n_pad = (n + kNB - 1) / kNB * kNB, nt = n_pad / kNB;
    const int64_t ld = n_pad;
    DevBuf<double> D, diag0, B, Cp, Wp;
    DevBuf<int64_t> dptr;
    DevBuf<int32_t> dcol, dstatus;
    DevBuf<double> dval;
    FS_HIP(D.alloc((size_t)n_pad * n_pad));
    FS_HIP(D.zero(st));
    FS_HIP(diag0.alloc(n_pad));
    FS_HIP(B.alloc(kNB * kNB));
    FS_HIP(Cp.alloc((size_t)n_pad * kNB));
    FS_HIP(Wp.alloc((size_t)n_pad * kNB));
    FS_HIP(dptr.upload(A.ptr, st));
    FS_HIP(dcol.upload(A.col, st));
    FS_HIP(dval.upload(A.val, st));
    FS_HIP(dstatus.alloc(2));
    FS_HIP(dstatus.zero(st));
    hipEvent_t e0, e1;
    FS_HIP(hipEventCreate(&e0));
    FS_HIP(hipEventCreate(&e1));
    FS_HIP(hipEventRecord(e0, st));
    const int64_t nent = (int64_t)A.val.size();
    hipLaunchKernelGGL(k_dense_scatter, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, dptr.p, dcol.p, dval.p, A.nr, D.p, ld);
    hipLaunchKernelGGL(k_dense_symmetrize, dim3((unsigned)(((int64_t)n_pad * n_pad + 255) / 256)), dim3(256), 0, st, D.p, ld, n, n_pad, diag0.p);
    const int tiles = nt * (nt + 1) / 2;
    const size_t lds = 2 * (size_t)kNB * kLdp * sizeof(double);
    const size_t lds_update = std::max((size_t)kNB * kLdp, 2 * (size_t)kNB * kLdh) * sizeof(double);
    FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_panels), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int K = 0; K < nt; K++) {
        hipLaunchKernelGGL(k_dense_pivot, dim3(1), dim3(256), 0, st, D.p, ld, K, diag0.p, B.p, dstatus.p);
        hipLaunchKernelGGL(k_dense_panels, dim3(nt), dim3(256), lds, st, D.p, ld, K, B.p, Cp.p, Wp.p);
        hipLaunchKernelGGL(k_dense_update, dim3(tiles), dim3(256), lds_update, st, D.p, ld, K, B.p, Cp.p, Wp.p);
    }
    const int64_t ldo = (n + 1) / 2 * 2;
    const unsigned gfin = (unsigned)(((int64_t)n * ldo + 255) / 256);
    if (single_precision) {
        FS_HIP(inv32->alloc((size_t)n * ldo));
        hipLaunchKernelGGL(k_dense_finish<float>, dim3(gfin), dim3(256), 0, st, D.p, ld, n, inv32->p, ldo);
    } else {
        FS_HIP(inv64->alloc((size_t)n * ldo));
        hipLaunchKernelGGL(k_dense_finish<double>, dim3(gfin), dim3(256), 0, st, D.p, ld, n, inv64->p, ldo);
    }
    FS_HIP(hipEventRecord(e1, st));
    int32_t hstatus[2] = {0, 0};
    FS_HIP(hipMemcpyAsync(hstatus, dstatus.p, sizeof hstatus, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    FS_HIP(hipGetLastError());
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *lda_out = ldo;
    if (stats) {
        stats->n = n;
        stats->ms = ms;
        stats->mfma_flops = (double)nt * ((double)tiles + nt) * 2.0 * kNB * kNB * kNB; // issued on the matrix cores
        stats->useful_flops = (double)n * n * n;                                       // n^3 of a symmetric inversion
        stats->dropped = hstatus[1];
        stats->bytes = (double)nt * (double)tiles * 2.0 * kNB * kNB * 8.0;             // lower triangle read + written per step
    }
    if (hstatus[0] != 0) return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: coarsest operator is not positive definite");
    return FEMSHELL_OK;
}

void launch_dense_gemv_big(const double *A64, const float *A32, int64_t lda, const double *b, double *y, int32_t n, int32_t n_pad6,
                           const CgScalars *gate, hipStream_t st)
{
    const dim3 g((unsigned)((n_pad6 + 3) / 4)), blk(256);
    if (A32 != nullptr) hipLaunchKernelGGL(k_dense_gemv_big<float>, g, blk, 0, st, A32, lda, b, y, n, n_pad6, gate);
    else hipLaunchKernelGGL(k_dense_gemv_big<double>, g, blk, 0, st, A64, lda, b, y, n, n_pad6, gate);
}

} // namespace femshell
