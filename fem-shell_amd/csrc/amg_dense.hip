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
#include <cstdlib>
#include <string>
#include <vector>

namespace femshell {

namespace {

constexpr int kNB = 64;   // tile edge
constexpr int kSW = 128;  // sweep width: two tiles per block sweep (halves the passes over the lower triangle)
constexpr int kLdp = 66;  // LDS row stride of a transposed 64 x 64 tile (doubles)
constexpr int kHalf = 32, kLdh = 34; // the matrix-core kernels stage their operands in chunks of 32 columns of K
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

// The sweeps read and write the lower triangle only, and the lower triangle of a Galerkin operator is as good a symmetric
// matrix as the mean of its two triangles (they differ in the 16th digit): nothing to symmetrise, only the diagonal to copy
// for the pivot test and a unit diagonal to put on the padding rows (0.31 ms of transposed reads less at 7386 dofs).
__global__ __launch_bounds__(256) void k_dense_prepare(double *__restrict__ D, int64_t ld, int n, int n_pad, double *__restrict__ diag0)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_pad) diag0[e] = e < n ? D[e * ld + e] : 1.0;
    // padding rows: (n_pad - n) x n_pad entries
    const int64_t q = e;
    if (q < (int64_t)(n_pad - n) * n_pad) {
        const int r = n + (int)(q / n_pad), c = (int)(q % n_pad);
        if (c <= r) D[(int64_t)r * ld + c] = r == c ? 1.0 : 0.0;
    }
}

// out (n x ldo, both triangles) <- -(lower triangle of D), tile by tile: the tile goes out as it is and, transposed through
// LDS, as its mirror image (coalesced on both sides; the element-wise version read the upper half column-wise)
template <class T>
__global__ __launch_bounds__(256) void k_dense_finish_tiled(const double *__restrict__ D, int64_t ld, int n, T *__restrict__ out, int64_t ldo)
{
    __shared__ double tile[kNB][kNB + 1];
    const int t = blockIdx.x, tid = threadIdx.x;
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    while (i * (i + 1) / 2 > t) i--;
    const int j = t - i * (i + 1) / 2;
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int r = e / kNB, c = e % kNB;
        const int gr = i * kNB + r, gc = j * kNB + c;
        const double v = (gr < n && gc < n) ? -D[(int64_t)gr * ld + gc] : 0.0;
        tile[r][c] = v;
        if (gr < n && gc < ldo && (i != j || c <= r)) out[(int64_t)gr * ldo + gc] = (T)v; // (gc in [n, ldo): the zero padding column)
    }
    __syncthreads();
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int c = e / kNB, r = e % kNB; // out row = tile column
        const int gr = j * kNB + c, gc = i * kNB + r;
        if (gr < n && gc < ldo && (i != j || r > c)) out[(int64_t)gr * ldo + gc] = (T)((gc < n) ? tile[r][c] : 0.0);
    }
}

// ---- the 64 x 64 building block: s <- -(s^-1) on the live directions ---------------------------------------------
// Thread (bi, bj) of a 256-thread workgroup keeps the 4 x 4 sub-block (4 bi .., 4 bj ..) of a symmetric 64 x 64 block in
// registers; the block is swept four pivots at a time: the four rows of a step (= its four columns) travel through a
// double-buffered LDS panel, every thread sweeps the 4 x 4 pivot sub-block itself (scalar symmetric sweeps with the pivot
// test against the ORIGINAL diagonal a0) and applies the rank-4 update to its registers: 16 steps, one barrier each.
// (Swept in LDS one pivot at a time with three barriers per pivot this took as long as the trailing update of the whole
// matrix, 111 us per step at 7386 dofs.)
__device__ __forceinline__ void sweep64(double s[4][4], const double *a0, double (*rowbuf)[4][kNB], int bi, int bj, int &n_dead,
                                        int &n_failed)
{
    for (int g = 0; g < kNB / 4; g++) {
        const int cur = g & 1;
        if (bi == g) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) rowbuf[cur][k][4 * bj + b] = s[k][b];
        }
        __syncthreads();
        double m[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 4; l++) m[k][l] = rowbuf[cur][k][4 * g + l];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double d = m[k][k], a = a0[4 * g + k];
            const bool failed = !(a > 0.0) || d < -1e-4 * fabs(a);
            const bool dead = failed || d <= 1e-11 * a;
            n_failed += failed ? 1 : 0;
            n_dead += dead ? 1 : 0;
            // reciprocal by the hardware estimate and two Newton steps (an IEEE division is a forty-instruction dependent
            // chain, and four of them sit on the critical path of every step)
            double inv = 0.0;
            if (!dead) {
                inv = __builtin_amdgcn_rcp(d);
                inv = inv * (2.0 - d * inv);
                inv = inv * (2.0 - d * inv);
            }
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
        double ri[4][4], rj[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                ri[k][q] = rowbuf[cur][k][4 * bi + q];
                rj[k][q] = rowbuf[cur][k][4 * bj + q];
            }
        // block sweep with B4 = -m:  S_ij <- S_ij - R_i^T B4 R_j;  rows of the step <- B4 R_j;  columns <- R_i^T B4;  pivot block <- m
        double ti[4][4];
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
        } else if (bi == g) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; l++) v -= m[a][l] * rj[l][b];
                    s[a][b] = v;
                }
        } else if (bj == g) {
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
    __syncthreads(); // rowbuf may be reused
}

// 32 x 32 quadrant (r0, c0) of X Y^T for two staged operands of kDepth columns (row stride kStride): acc[ti][tj], tile (ti, tj)
// of the quadrant.  Operand maps of v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; result
// register g of lane l is D[(l >> 4) + 4 g][l & 15].  B[k][n] = Y[n][k]: both operands are read as (row l & 15, k l >> 4).
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

// element loop of the 32 x 32 quadrant a wave owns in the matrix-core layout: f(ti, tj, g, row, col)
template <class F> __device__ __forceinline__ void for_quadrant(int r0, int c0, int lane, F f)
{
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
            for (int g = 0; g < 4; g++) f(ti, tj, g, r0 + 16 * ti + (lane >> 4) + 4 * g, c0 + 16 * tj + (lane & 15));
}

// B (128 x 128) = inverse of the pivot block K = [[A, C^T], [C, D]] of the sweep (tiles (2K,2K), (2K+1,2K), (2K+1,2K+1)),
// restricted to the live directions: B_A = A^-1, W = C B_A, S = D - W C^T, B_S = S^-1, then
//   B = [[B_A + W^T B_S W, -W^T B_S], [-B_S W, B_S]].
// One workgroup: the two 64 x 64 inverses by sweep64 (registers, thread per 4 x 4), the four 64^3 products on the matrix
// cores from LDS (row stride 66: conflict-free; a product with thread-per-4x4 FMAs from LDS took 98 us per sweep, more than
// the trailing update it feeds).  LDS: P0 = B_A, P1 = C -> S -> B_S, P2 = W -> X21^T, P3 = W^T.
// status[0] = 1 on a clearly negative pivot, status[1] counts the dropped directions.
// (panels: the four 64 x 66 work blocks P0 .. P3 -- LDS in k_dense_pivot; in the look-ahead of k_dense_update, whose
//  workgroups have 35 KB of LDS each, a scratch buffer in HBM that stays in the caches: the products then read their operands
//  through global loads, slower than from LDS but beside the trailing update instead of in front of it.  small: 2 x 4 x 64
//  doubles of row panel + 128 diagonal entries, LDS in both.)
__device__ __forceinline__ void pivot_block(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ diag0,
                                            double *__restrict__ B, int32_t *status, double *panels, double *small_lds)
{
    constexpr int L = kLdp;
    double *P0 = panels, *P1 = P0 + kNB * L, *P2 = P1 + kNB * L, *P3 = P2 + kNB * L;
    double (*rowbuf)[4][kNB] = reinterpret_cast<double (*)[4][kNB]>(small_lds);
    double *a0 = small_lds + 2 * 4 * kNB; // 128 original diagonal entries
    const int tid = threadIdx.x, k0 = 2 * K * kNB, bi = tid >> 4, bj = tid & 15;
    const int wave = tid >> 6, lane = tid & 63, r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 4 * bi + a, c = 4 * bj + b;
            s[a][b] = r >= c ? D[(int64_t)(k0 + r) * ld + k0 + c] : D[(int64_t)(k0 + c) * ld + k0 + r];
            P1[r * L + c] = D[(int64_t)(k0 + kNB + r) * ld + k0 + c]; // C
        }
    // -D of the second diagonal tile in the matrix-core layout (the accumulator of S = D - W C^T starts from it)
    v4d acc[2][2];
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
        acc[ti][tj][g] = -(r >= c ? D[(int64_t)(k0 + kNB + r) * ld + k0 + kNB + c] : D[(int64_t)(k0 + kNB + c) * ld + k0 + kNB + r]);
    });
    if (tid < 2 * kNB) a0[tid] = diag0[k0 + tid];
    __syncthreads();
    int n_dead = 0, n_failed = 0;
    sweep64(s, a0, rowbuf, bi, bj, n_dead, n_failed); // s = -B_A
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) P0[(4 * bi + a) * L + 4 * bj + b] = -s[a][b];
    __syncthreads();
    {
        v4d w[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) w[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        quadrant_xyt<kNB, L>(P1, P0, r0, c0, lane, w); // W = C B_A = C (B_A^T)^T
        for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
            P2[r * L + c] = w[ti][tj][g];
            P3[c * L + r] = w[ti][tj][g];
        });
    }
    __syncthreads();
    quadrant_xyt<kNB, L>(P2, P1, r0, c0, lane, acc); // W C^T - D
    __syncthreads();                                  // every wave is through with C
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) { P1[r * L + c] = -acc[ti][tj][g]; }); // S
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) s[a][b] = P1[(4 * bi + a) * L + 4 * bj + b];
    __syncthreads();
    sweep64(s, a0 + kNB, rowbuf, bi, bj, n_dead, n_failed); // s = -B_S
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 4 * bi + a, c = 4 * bj + b;
            P1[r * L + c] = -s[a][b];
            B[(kNB + r) * kSW + kNB + c] = -s[a][b]; // X22 = B_S
        }
    __syncthreads();
    {
        v4d x[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) x[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        quadrant_xyt<kNB, L>(P1, P3, r0, c0, lane, x); // B_S W = B_S (W^T)^T
        __syncthreads();                                // every wave is through with W^T ... and nobody reads W any more
        for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
            const double v = -x[ti][tj][g]; // X21 = -B_S W
            B[(kNB + r) * kSW + c] = v;
            B[c * kSW + kNB + r] = v;       // X12 = X21^T
            P2[c * L + r] = v;              // X21^T for the last product
        });
    }
    __syncthreads();
    {
        v4d x[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) x[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        quadrant_xyt<kNB, L>(P3, P2, r0, c0, lane, x); // W^T X21 = W^T (X21^T)^T
        for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) { B[r * kSW + c] = P0[r * L + c] - x[ti][tj][g]; }); // X11
    }
    if (tid == 0 && n_dead) {
        if (n_failed) status[0] = 1;
        atomicAdd(&status[1], n_dead);
    }
}

__global__ __launch_bounds__(256) void k_dense_pivot(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ diag0,
                                                     double *__restrict__ B, int32_t *status)
{
    extern __shared__ double lds_dense[];
    pivot_block(D, ld, K, diag0, B, status, lds_dense, lds_dense + 4 * kNB * kLdp);
}

// ---- the same pivot inverse for a workgroup that has 35 KB of LDS and 128 VGPRs (the look-ahead inside k_dense_update) --------
// sweep64 with the row panel consumed as it is read: s, the 4 x 4 pivot sub-block and one 4 x 4 product live in registers
// (52 doubles instead of 100), the rows of the step come from LDS a row of four at a time
__device__ __forceinline__ void sweep64_lean(double s[4][4], const double *a0, double (*rowbuf)[4][kNB], int bi, int bj, int &n_dead,
                                             int &n_failed)
{
    for (int g = 0; g < kNB / 4; g++) {
        const int cur = g & 1;
        if (bi == g) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) rowbuf[cur][k][4 * bj + b] = s[k][b];
        }
        __syncthreads();
        double m[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 4; l++) m[k][l] = rowbuf[cur][k][4 * g + l];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double d = m[k][k], a = a0[4 * g + k];
            const bool failed = !(a > 0.0) || d < -1e-4 * fabs(a);
            const bool dead = failed || d <= 1e-11 * a;
            n_failed += failed ? 1 : 0;
            n_dead += dead ? 1 : 0;
            double inv = 0.0;
            if (!dead) {
                inv = __builtin_amdgcn_rcp(d);
                inv = inv * (2.0 - d * inv);
                inv = inv * (2.0 - d * inv);
            }
            double col[4];
#pragma unroll
            for (int l = 0; l < 4; l++) col[l] = m[l][k];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i == k || j == k) continue;
                    m[i][j] -= col[i] * col[j] * inv;
                }
#pragma unroll
            for (int l = 0; l < 4; l++) {
                if (l == k) continue;
                const double v = col[l] * inv;
                m[l][k] = v;
                m[k][l] = v;
            }
            m[k][k] = -inv;
        }
        // ti = -R_i^T m, one row of the panel at a time
        double ti[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int l = 0; l < 4; l++) ti[a][l] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double ri[4];
#pragma unroll
            for (int q = 0; q < 4; q++) ri[q] = rowbuf[cur][k][4 * bi + q];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int l = 0; l < 4; l++) ti[a][l] -= ri[a] * m[k][l];
        }
        if (bi == g && bj == g) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) s[a][b] = m[a][b];
        } else if (bj == g) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) s[a][b] = ti[a][b];
        } else {
            const bool row_of_step = bi == g;
            if (row_of_step) {
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) s[a][b] = 0.0;
            }
#pragma unroll
            for (int l = 0; l < 4; l++) {
                double rj[4];
#pragma unroll
                for (int q = 0; q < 4; q++) rj[q] = rowbuf[cur][l][4 * bj + q];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) s[a][b] -= (row_of_step ? m[a][l] : ti[a][l]) * rj[b];
            }
        }
    }
    __syncthreads(); // rowbuf may be reused
}

// acc += X Y^T for two 64 x 64 blocks in HBM (row stride kLdp), staged through LDS in two chunks of 32 columns of K (two
// 64 x 34 panels = the 35 KB of an update workgroup), the wave's 32 x 32 quadrant on the matrix cores
__device__ __forceinline__ void staged_xyt(const double *__restrict__ Xg, const double *__restrict__ Yg, double *lds, int r0, int c0,
                                           int lane, int tid, v4d acc[2][2])
{
    double *Xs = lds, *Ys = lds + kNB * kLdh;
    for (int ch = 0; ch < kNB / kHalf; ch++) {
        __syncthreads(); // the previous readers of the panels are through; the blocks in HBM are complete
        for (int e = tid; e < kNB * kHalf / 2; e += 256) {
            const int r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
            *reinterpret_cast<double2 *>(Xs + r * kLdh + k) = *reinterpret_cast<const double2 *>(Xg + r * kLdp + ch * kHalf + k);
            *reinterpret_cast<double2 *>(Ys + r * kLdh + k) = *reinterpret_cast<const double2 *>(Yg + r * kLdp + ch * kHalf + k);
        }
        __syncthreads();
        quadrant_xyt<kHalf, kLdh>(Xs, Ys, r0, c0, lane, acc);
    }
}

// pivot_block for such a workgroup: P0 .. P3 in HBM scratch (cache resident), lds = 2 x 64 x 34 doubles, used as the row
// panel + diagonal entries during the sweeps and as the two operand panels during the products
__device__ __forceinline__ void pivot_block_lean(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ diag0,
                                                 double *__restrict__ B, int32_t *status, double *panels, double *lds)
{
    constexpr int L = kLdp;
    double *P0 = panels, *P1 = P0 + kNB * L, *P2 = P1 + kNB * L, *P3 = P2 + kNB * L;
    double (*rowbuf)[4][kNB] = reinterpret_cast<double (*)[4][kNB]>(lds);
    double *a0 = lds + 2 * 4 * kNB;
    const int tid = threadIdx.x, k0 = 2 * K * kNB, bi = tid >> 4, bj = tid & 15;
    const int wave = tid >> 6, lane = tid & 63, r0 = 32 * (wave >> 1), c0 = 32 * (wave & 1);
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 4 * bi + a, c = 4 * bj + b;
            s[a][b] = r >= c ? D[(int64_t)(k0 + r) * ld + k0 + c] : D[(int64_t)(k0 + c) * ld + k0 + r];
            P1[r * L + c] = D[(int64_t)(k0 + kNB + r) * ld + k0 + c]; // C
        }
    if (tid < 2 * kNB) a0[tid] = diag0[k0 + tid];
    __syncthreads();
    int n_dead = 0, n_failed = 0;
    sweep64_lean(s, a0, rowbuf, bi, bj, n_dead, n_failed); // s = -B_A
    double a0_second = tid < kNB ? a0[kNB + tid] : 0.0;     // (the products take the LDS: the second block's diagonal rides in registers)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) P0[(4 * bi + a) * L + 4 * bj + b] = -s[a][b];
    v4d w[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) w[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    };
    zero();
    staged_xyt(P1, P0, lds, r0, c0, lane, tid, w); // W = C B_A
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
        P2[r * L + c] = w[ti][tj][g];
        P3[c * L + r] = w[ti][tj][g];
    });
    // S = D22 - W C^T: the accumulator starts from -D22
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
        w[ti][tj][g] = -(r >= c ? D[(int64_t)(k0 + kNB + r) * ld + k0 + kNB + c] : D[(int64_t)(k0 + kNB + c) * ld + k0 + kNB + r]);
    });
    staged_xyt(P2, P1, lds, r0, c0, lane, tid, w); // W C^T - D22
    __syncthreads();                                // every wave is through with C
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) { P1[r * L + c] = -w[ti][tj][g]; }); // S
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) s[a][b] = P1[(4 * bi + a) * L + 4 * bj + b];
    if (tid < kNB) a0[tid] = a0_second;
    __syncthreads();
    sweep64_lean(s, a0, rowbuf, bi, bj, n_dead, n_failed); // s = -B_S
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 4 * bi + a, c = 4 * bj + b;
            P1[r * L + c] = -s[a][b];
            B[(kNB + r) * kSW + kNB + c] = -s[a][b]; // X22 = B_S
        }
    zero();
    staged_xyt(P1, P3, lds, r0, c0, lane, tid, w); // B_S W = B_S (W^T)^T
    __syncthreads();                                // every wave is through with W^T ... and nobody reads W any more
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) {
        const double v = -w[ti][tj][g]; // X21 = -B_S W
        B[(kNB + r) * kSW + c] = v;
        B[c * kSW + kNB + r] = v;       // X12 = X21^T
        P2[c * L + r] = v;              // X21^T for the last product
    });
    zero();
    staged_xyt(P3, P2, lds, r0, c0, lane, tid, w); // W^T X21 = W^T (X21^T)^T
    for_quadrant(r0, c0, lane, [&](int ti, int tj, int g, int r, int c) { B[r * kSW + c] = P0[r * L + c] - w[ti][tj][g]; }); // X11
    if (tid == 0 && n_dead) {
        if (n_failed) status[0] = 1;
        atomicAdd(&status[1], n_dead);
    }
}

// C_i = A(i, block K) (64 x 128), gathered from the lower triangle (tiles (i, 2K), (i, 2K+1) below the pivot block, the
// transposed tiles (2K, i), (2K+1, i) above it), W_i = C_i B on the matrix cores.  One workgroup per 64-row block; the two
// block rows of the pivot block itself are skipped.  K runs in four chunks of 32 through LDS (row stride 34 doubles: the 32
// lanes of a ds_read_b64 group hit 32 different bank pairs); B is symmetric, so W = C B = C (B^T)^T has the X Y^T form and
// both operands are read the same way.  Wave w computes the output columns [32 w, 32 w + 32) of all 64 rows.
// (kSplit: two workgroups per 64-row block, 32 rows each -- 2 x 116 workgroups instead of 116 on 256 CUs; the panel product
//  sits between the pivot inverse and the trailing update of every sweep, on the critical path)
template <bool kSplit>
__global__ __launch_bounds__(256) void k_dense_panels(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                      double *__restrict__ Cp, double *__restrict__ Wp)
{
    constexpr int kRows = kSplit ? kNB / 2 : kNB; // rows of C_i this workgroup takes
    __shared__ double Cs[kRows * kLdh];
    __shared__ double Bs[kSW * kLdh];
    const int i = kSplit ? blockIdx.x >> 1 : blockIdx.x, row0 = kSplit ? (blockIdx.x & 1) * kRows : 0, tid = threadIdx.x;
    if ((i >> 1) == K) return;
    const int wave = tid >> 6, lane = tid & 63;
    v4d acc[kRows / 32][2][2]; // [row half][ti][tj]
#pragma unroll
    for (int h = 0; h < kRows / 32; h++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[h][a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const bool below = i > 2 * K + 1;
    for (int ch = 0; ch < kSW / kHalf; ch++) { // columns [32 ch, 32 ch + 32) of C_i = K range of the product
        if (ch) __syncthreads();
        for (int e = tid; e < kRows * kHalf; e += 256) {
            const int r = e / kHalf, k = e % kHalf, col = kHalf * ch + k; // column of the sweep block
            const int64_t gr = (int64_t)i * kNB + row0 + r, gc = (int64_t)2 * K * kNB + col;
            const double v = below ? D[gr * ld + gc] : D[gc * ld + gr]; // (above: a transposed, column-wise read)
            Cs[r * kLdh + k] = v;
            Cp[gr * kSW + col] = v;
        }
        for (int e = tid; e < kSW * kHalf; e += 256) {
            const int n = e / kHalf, k = e % kHalf;
            Bs[n * kLdh + k] = B[n * kSW + kHalf * ch + k]; // B^T[k][n] = B[n][k]
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < kRows / 32; h++) quadrant_xyt<kHalf, kLdh>(Cs, Bs, 32 * h, 32 * wave, lane, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < kRows / 32; h++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj < 2; tj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r = row0 + 32 * h + 16 * ti + (lane >> 4) + 4 * g, c = 32 * wave + 16 * tj + (lane & 15);
                    Wp[((int64_t)i * kNB + r) * kSW + c] = acc[h][ti][tj][g];
                }
}

// Look-ahead of the block sweep (la.on; FEMSHELL_AMG_DENSE_LOOKAHEAD=0 switches it off): the three tiles of the NEXT pivot
// block are the first workgroups of the grid; they count themselves done in la.flag (release), and workgroup 3 -- which has
// no tile -- waits for them (acquire, bounded) and inverts that block into la.B_next while the other workgroups update the
// rest of the triangle.  The one-workgroup pivot inverse (71 us of dependent steps per sweep, 4.1 of 17.2 ms at 7386 dofs)
// thereby leaves the critical path without a second stream and its event waits (round 3: 29.7 ms that way).  A workgroup of
// this kernel has 35 KB of LDS and 128 VGPRs where k_dense_pivot takes 138 KB and 194, so the look-ahead runs
// pivot_block_lean: the 64 x 66 work blocks in la.scratch (HBM, cache resident), every product staged through the 35 KB
// in two chunks of 32 columns like the update's own operands, and a sweep that consumes its row panel as it reads it.
// MEASURED (round 4, 7386 dofs, alternating on one box): 17.1 ms without -> 14.2 ms -> 13.2 ms with the panel kernel's rows
// split over two workgroups and the symmetrisation pass dropped = 32.2 TFLOP/s issued = 41 % of the FP64 matrix peak.  (The
// first version ran the unchanged pivot_block with its blocks in HBM: its products behind global loads and 68 spilled
// doubles took 290 us per sweep, longer than the 161 us update -- 19.2 ms.)
struct DenseLookAhead {
    int on = 0;
    const double *diag0 = nullptr;
    double *B_next = nullptr, *scratch = nullptr;
    int32_t *status = nullptr;
    unsigned int *flag = nullptr; // one counter per sweep, zero-initialised
    int spin_limit = 1 << 24;     // polls of the pivot workgroup before it gives up (FEMSHELL_AMG_DENSE_LOOKAHEAD_SPINS: tests)
};

// one workgroup per lower 64 x 64 tile (i >= j): the sweep of the 128-wide block K
// (kLookAhead = false: the kernel without the pivot workgroup's code -- the last sweep, and FEMSHELL_AMG_DENSE_LOOKAHEAD=0)
template <bool kLookAhead>
__global__ __launch_bounds__(256, 4) void k_dense_update(double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                         const double *__restrict__ Cp, const double *__restrict__ Wp, int tiles,
                                                         DenseLookAhead la)
{
    __shared__ double lds_upd[2 * kNB * kLdh > kNB * kLdp ? 2 * kNB * kLdh : kNB * kLdp];
    double *Ws = lds_upd, *Cs = lds_upd + kNB * kLdh;
    const int tid = threadIdx.x;
    // workgroup -> tile.  With the look-ahead: workgroups 0, 1, 2 take the tiles (2K+2, 2K+2), (2K+3, 2K+2), (2K+3, 2K+3) of the
    // next pivot block, workgroup 3 is the pivot workgroup, and the tiles those three took are handed to the workgroups that
    // would have had the indices 0 .. 2 (a swap: every tile still has exactly one workgroup)
    int t = blockIdx.x;
    bool ahead_tile = false;
    if (kLookAhead && la.on) {
        const int i2 = 2 * K + 2;
        const int t_a = i2 * (i2 + 1) / 2 + i2, t_b = (i2 + 1) * (i2 + 2) / 2 + i2, t_c = t_b + 1; // linear indices of the three tiles
        if (blockIdx.x == 3) {
            // ---- the pivot workgroup
            __shared__ int ok;
            if (tid == 0) {
                int spins = 0;
                while (spins < la.spin_limit && __hip_atomic_load(la.flag + K, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 3u) {
                    __builtin_amdgcn_s_sleep(8);
                    spins++;
                }
                ok = spins < la.spin_limit ? 1 : 0;
                // (not seen on a card of its own: the three tile workgroups precede this one in dispatch order.  The host runs the
                //  inverse again without the look-ahead: amg_dense_inverse_device)
                if (!ok) la.status[0] = 2;
            }
            __syncthreads();
            if (!ok) return;
            __threadfence(); // (acquire for every thread's loads of the three tiles)
            pivot_block_lean(D, ld, K + 1, la.diag0, la.B_next, la.status, la.scratch, lds_upd);
            return;
        }
        // blockIdx 0..2 -> t_a, t_b, t_c; blockIdx 4.. -> its own index shifted by one, and whoever lands on t_a / t_b / t_c
        // takes the tile of the workgroup that left (0, 1, 2)
        if (blockIdx.x < 3) {
            t = blockIdx.x == 0 ? t_a : (blockIdx.x == 1 ? t_b : t_c);
            ahead_tile = true;
        } else {
            t = blockIdx.x - 1;                 // 3 .. tiles-1 (the pivot workgroup took index 3)
            if (t == t_a) t = 0;
            else if (t == t_b) t = 1;
            else if (t == t_c) t = 2;
            // (t = 3 .. tiles-1 covers every tile but 0, 1, 2 and, through the swap, t_a, t_b, t_c are replaced by them)
        }
        if (t >= tiles) return;
    }
    auto done = [&]() { // a tile of the next pivot block is in place
        if (ahead_tile) {
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(la.flag + K, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // linear tile index -> (i, j), i >= j
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    while (i * (i + 1) / 2 > t) i--;
    const int j = t - i * (i + 1) / 2;
    double *tile = D + (int64_t)(i * kNB) * ld + j * kNB;
    const bool ki = (i >> 1) == K, kj = (j >> 1) == K;
    if (ki && kj) { // inside the pivot block: -B
        for (int e = tid; e < kNB * kNB; e += 256)
            tile[(int64_t)(e / kNB) * ld + e % kNB] = -B[((i & 1) * kNB + e / kNB) * kSW + (j & 1) * kNB + e % kNB];
        return;
    }
    if (kj) { // below the pivot block: A_i,K <- W_i
        for (int e = tid; e < kNB * kNB; e += 256)
            tile[(int64_t)(e / kNB) * ld + e % kNB] = Wp[((int64_t)i * kNB + e / kNB) * kSW + (j & 1) * kNB + e % kNB];
        return;
    }
    if (ki) { // left of the pivot block: A_K,j <- W_j^T
        for (int e = tid; e < kNB * kNB; e += 256)
            Ws[(e / kNB) * kLdp + e % kNB] = Wp[((int64_t)j * kNB + e / kNB) * kSW + (i & 1) * kNB + e % kNB];
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
    // K = 128 in four chunks of 32 (row stride 34 doubles: conflict-free): 35 KB of LDS per workgroup, four per CU
    const double *wsrc = Wp + (int64_t)(i * kNB) * kSW, *csrc = Cp + (int64_t)(j * kNB) * kSW;
    // (MEASURED, round 5, and dropped: the next chunk's operands fetched into registers while the matrix cores work on this chunk's
    //  -- eight 16-byte words per thread.  With the look-ahead's pivot code in the same kernel the 128 registers of four workgroups
    //  per CU do not hold them: 480 bytes of scratch per lane, 25.0 ms instead of 13.7 ms.)
    for (int h = 0; h < kSW / kHalf; h++) {
        if (h) __syncthreads(); // the previous chunk's readers are through
        for (int e = tid; e < kNB * kHalf / 2; e += 256) { // 16-byte words: 16 per row and chunk
            const int r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
            const double2 wv = *reinterpret_cast<const double2 *>(wsrc + r * kSW + h * kHalf + k);
            const double2 cv = *reinterpret_cast<const double2 *>(csrc + r * kSW + h * kHalf + k);
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
    done();
}

// ---- the trailing update on 128 x 128 tiles (round 5) ---------------------------------------------------------------------------
// One workgroup per lower 128 x 128 tile of the triangle (= 2 x 2 of the 64-tiles above; the sweep's pivot block is exactly one
// such tile), every wave a 64 x 64 quadrant = 4 x 4 matrix-core tiles: per step of four columns of K a wave reads eight operands
// from LDS and issues sixteen v_mfma_f64_16x16x4_f64 where the 64 x 64 kernel reads four for four, and the workgroup reads its
// two 128 x 128 operand panels and its tile once for 4.2 MFLOP -- 8.2 flop per byte where the 64-tiles have 5.5.  70 KB of LDS
// per workgroup (operands in chunks of 32 columns, row stride 34), two workgroups per CU, 128 accumulator registers.
// Diagonal tiles compute the three quadrants of the lower triangle (the wave of the upper right one stages and waits).
// Look-ahead: workgroup 0 takes the next pivot block's tile -- ONE tile now -- and workgroup 1 inverts it once it is in place.
template <int kDepth, int kStride>
__device__ __forceinline__ void quadrant64_xyt(const double *Xs, const double *Ys, int r0, int c0, int lane, v4d acc[4][4])
{
    const int rr = lane & 15, kq = lane >> 4;
#pragma unroll 2
    for (int k = 0; k < kDepth; k += 4) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            a[t] = Xs[(r0 + 16 * t + rr) * kStride + k + kq];
            b[t] = Ys[(c0 + 16 * t + rr) * kStride + k + kq];
        }
#pragma unroll
        for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int tj = 0; tj < 4; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
}

template <bool kLookAhead>
__global__ __launch_bounds__(256, 2) void k_dense_update128(double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                            const double *__restrict__ Cp, const double *__restrict__ Wp, int stiles,
                                                            DenseLookAhead la)
{
    extern __shared__ double lds_big[]; // 2 x 128 x 34 doubles
    double *Ws = lds_big, *Cs = lds_big + kSW * kLdh;
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    bool ahead_tile = false;
    if (kLookAhead && la.on) {
        const int K1 = K + 1, t_a = K1 * (K1 + 1) / 2 + K1; // the next pivot block's tile
        if (blockIdx.x == 1) {
            // ---- the pivot workgroup (see k_dense_update: bounded wait for the workgroup that precedes it in dispatch order)
            __shared__ int ok;
            if (tid == 0) {
                int spins = 0;
                while (spins < la.spin_limit && __hip_atomic_load(la.flag + K, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 1u) {
                    __builtin_amdgcn_s_sleep(8);
                    spins++;
                }
                ok = spins < la.spin_limit ? 1 : 0;
                if (!ok) la.status[0] = 2;
            }
            __syncthreads();
            if (!ok) return;
            __threadfence();
            pivot_block_lean(D, ld, K + 1, la.diag0, la.B_next, la.status, la.scratch, lds_big);
            return;
        }
        if (blockIdx.x == 0) {
            t = t_a;
            ahead_tile = true;
        } else {
            t = blockIdx.x - 1; // 1 .. stiles-1; whoever lands on t_a takes the tile of the workgroup that left (0)
            if (t == t_a) t = 0;
        }
        if (t >= stiles) return;
    }
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) I++;
    while (I * (I + 1) / 2 > t) I--;
    const int J = t - I * (I + 1) / 2;
    double *tile = D + (int64_t)(I * kSW) * ld + J * kSW;
    const bool ki = I == K, kj = J == K;
    if (ki && kj) { // the pivot block: -B
        for (int e = tid; e < kSW * kSW; e += 256) tile[(int64_t)(e / kSW) * ld + e % kSW] = -B[e];
        return;
    }
    if (kj) { // below the pivot block: A_I,K <- W_I
        for (int e = tid; e < kSW * kSW; e += 256) tile[(int64_t)(e / kSW) * ld + e % kSW] = Wp[((int64_t)I * kSW + e / kSW) * kSW + e % kSW];
        return;
    }
    if (ki) { // left of the pivot block: A_K,J <- W_J^T, 64 rows of W_J at a time through LDS (row stride 65)
        constexpr int L = kNB + 1;
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();
            for (int e = tid; e < kNB * kSW; e += 256) // rows [64 half, 64 half + 64) of W_J, all 128 columns
                lds_big[(e % kSW) * L + e / kSW] = Wp[((int64_t)J * kSW + kNB * half + e / kSW) * kSW + e % kSW];
            __syncthreads();
            for (int e = tid; e < kSW * kNB; e += 256) // tile row r (= column of W), columns [64 half, 64 half + 64)
                tile[(int64_t)(e / kNB) * ld + kNB * half + e % kNB] = lds_big[(e / kNB) * L + e % kNB];
        }
        return;
    }
    const int wave = tid >> 6, lane = tid & 63, r0 = kNB * (wave >> 1), c0 = kNB * (wave & 1);
    const bool idle = I == J && wave == 1; // the quadrant above the diagonal of a diagonal tile
    v4d acc[4][4];
    if (!idle) {
#pragma unroll
        for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int tj = 0; tj < 4; tj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r = r0 + 16 * ti + (lane >> 4) + 4 * g, c = c0 + 16 * tj + (lane & 15);
                    acc[ti][tj][g] = -tile[(int64_t)r * ld + c]; // accumulate W C^T - A, store its negative
                }
    }
    const double *wsrc = Wp + (int64_t)(I * kSW) * kSW, *csrc = Cp + (int64_t)(J * kSW) * kSW;
    for (int h = 0; h < kSW / kHalf; h++) {
        if (h) __syncthreads(); // the previous chunk's readers are through
        for (int e = tid; e < kSW * kHalf / 2; e += 256) { // 16-byte words: 16 per row and chunk, 128 rows
            const int r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
            const double2 wv = *reinterpret_cast<const double2 *>(wsrc + r * kSW + h * kHalf + k);
            const double2 cv = *reinterpret_cast<const double2 *>(csrc + r * kSW + h * kHalf + k);
            *reinterpret_cast<double2 *>(Ws + r * kLdh + k) = wv;
            *reinterpret_cast<double2 *>(Cs + r * kLdh + k) = cv;
        }
        __syncthreads();
        if (!idle) quadrant64_xyt<kHalf, kLdh>(Ws, Cs, r0, c0, lane, acc);
    }
    if (!idle) {
#pragma unroll
        for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int tj = 0; tj < 4; tj++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r = r0 + 16 * ti + (lane >> 4) + 4 * g, c = c0 + 16 * tj + (lane & 15);
                    tile[(int64_t)r * ld + c] = -acc[ti][tj][g];
                }
    }
    if (ahead_tile) {
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(la.flag + K, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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

namespace {

// (*timed_out: the look-ahead's bounded wait expired -- the caller runs the inverse again without it)
int dense_inverse_once(femshell_ctx *c, const Bsr &A, bool single_precision, DevBuf<double> *inv64, DevBuf<float> *inv32,
                       int64_t *lda_out, AmgDenseStats *stats, bool lookahead, bool *timed_out)
{
    hipStream_t st = c->stream;
    *timed_out = false;
    const int n = 6 * A.nr, n_pad = (n + kSW - 1) / kSW * kSW, nt = n_pad / kNB, ns = n_pad / kSW;
    const int64_t ld = n_pad;
    DevBuf<double> D, diag0, B, Cp, Wp, pivot_scratch;
    DevBuf<unsigned int> la_flags;
    DevBuf<int64_t> dptr;
    DevBuf<int32_t> dcol, dstatus;
    DevBuf<double> dval;
    FS_HIP(D.alloc((size_t)n_pad * n_pad));
    FS_HIP(D.zero(st));
    FS_HIP(diag0.alloc(n_pad));
    FS_HIP(B.alloc(2 * kSW * kSW)); // (two: the look-ahead writes the next sweep's while this sweep's is read)
    FS_HIP(pivot_scratch.alloc(4 * (size_t)kNB * kLdp));
    FS_HIP(la_flags.alloc((size_t)ns + 1));
    FS_HIP(la_flags.zero(st));
    FS_HIP(Cp.alloc((size_t)n_pad * kSW));
    FS_HIP(Wp.alloc((size_t)n_pad * kSW));
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
    // FEMSHELL_AMG_DENSE_SYMMETRIZE=1: the mean of the two triangles as before round 4 (A/B runs)
    if (getenv("FEMSHELL_AMG_DENSE_SYMMETRIZE") && atoi(getenv("FEMSHELL_AMG_DENSE_SYMMETRIZE")) == 1) {
        hipLaunchKernelGGL(k_dense_symmetrize, dim3((unsigned)(((int64_t)n_pad * n_pad + 255) / 256)), dim3(256), 0, st, D.p, ld, n, n_pad, diag0.p);
    } else {
        const int64_t work = std::max<int64_t>(n_pad, (int64_t)(n_pad - n) * n_pad);
        hipLaunchKernelGGL(k_dense_prepare, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, D.p, ld, n, n_pad, diag0.p);
    }
    const int tiles = nt * (nt + 1) / 2;
    // LDS of the pivot kernel: four 64 x 66 blocks, the row panel of the sweeps, 128 diagonal entries
    const size_t lds_pivot = (4 * (size_t)kNB * kLdp + 2 * 4 * kNB + 2 * kNB) * sizeof(double);
    FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_pivot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pivot));
    // (the pivot kernel is one workgroup and 70 us of dependent steps; running it on a second stream beside the previous
    //  sweep's update, after bringing its three tiles up to date first, was measured: the two event waits per sweep cost more
    //  than the overlap gains, 17.3 -> 29.7 ms)
    // Look-ahead: workgroup `tiles` of the update's grid waits (bounded spin) for workgroups 0-2 of the SAME grid.  That makes
    // progress because workgroups are dispatched in index order and the three it waits for never wait themselves -- an
    // observed property of the dispatcher, not a contract: on a card shared with other processes, or under a CU mask, the
    // bound can expire.  The kernel then marks status 2 and amg_dense_inverse_device runs the whole inverse again with the
    // pivot as a launch of its own (slower, same numbers).
    const bool panel_split = !(getenv("FEMSHELL_AMG_DENSE_PANEL_SPLIT") && atoi(getenv("FEMSHELL_AMG_DENSE_PANEL_SPLIT")) == 0);
    // FEMSHELL_AMG_DENSE_TILE=128: the trailing update on 128 x 128 tiles (k_dense_update128; default: 64 x 64, four workgroups
    // per CU).  MEASURED, round 5, 7386 dofs, alternating on one box: 14.1 ms against 13.7 ms, 192 against 177 us per sweep --
    // half the operand traffic and four times the matrix instructions per LDS read buy nothing: both kernels leave the matrix
    // pipe idle half of the time (SQ_VALU_MFMA_BUSY_CYCLES 0.48, profiles/r05_pmc_mfma.json) behind their synchronous operand
    // staging, and the larger tile has 1711 workgroups for 512 slots (a fourth, nearly empty round).  The measured alternative stays.
    const bool big_tiles = getenv("FEMSHELL_AMG_DENSE_TILE") && atoi(getenv("FEMSHELL_AMG_DENSE_TILE")) == 128;
    const int stiles = ns * (ns + 1) / 2;
    const size_t lds_big_bytes = 2 * (size_t)kSW * kLdh * sizeof(double);
    if (big_tiles) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_update128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big_bytes));
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_update128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big_bytes));
    }
    for (int K = 0; K < ns; K++) {
        double *Bk = B.p + (size_t)(K & 1) * kSW * kSW;
        if (K == 0 || !lookahead)
            hipLaunchKernelGGL(k_dense_pivot, dim3(1), dim3(256), lds_pivot, st, D.p, ld, K, diag0.p, Bk, dstatus.p);
        if (panel_split) hipLaunchKernelGGL(k_dense_panels<true>, dim3(2 * nt), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p);
        else hipLaunchKernelGGL(k_dense_panels<false>, dim3(nt), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p);
        DenseLookAhead la;
        if (lookahead && K + 1 < ns) {
            la.on = 1;
            la.diag0 = diag0.p;
            la.B_next = B.p + (size_t)((K + 1) & 1) * kSW * kSW;
            la.scratch = pivot_scratch.p;
            la.status = dstatus.p;
            la.flag = la_flags.p;
            if (const char *e = getenv("FEMSHELL_AMG_DENSE_LOOKAHEAD_SPINS")) la.spin_limit = atoi(e);
        }
        if (big_tiles) {
            if (la.on) hipLaunchKernelGGL(k_dense_update128<true>, dim3(stiles + 1), dim3(256), lds_big_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, stiles, la);
            else hipLaunchKernelGGL(k_dense_update128<false>, dim3(stiles), dim3(256), lds_big_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, stiles, la);
        } else if (la.on) {
            hipLaunchKernelGGL(k_dense_update<true>, dim3(tiles + 1), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p, tiles, la);
        } else {
            hipLaunchKernelGGL(k_dense_update<false>, dim3(tiles), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p, tiles, la);
        }
    }
    const int64_t ldo = (n + 1) / 2 * 2;
    const unsigned gfin = (unsigned)(((int64_t)n * ldo + 255) / 256);
    (void)gfin;
    if (single_precision) {
        FS_HIP(inv32->alloc((size_t)n * ldo));
        hipLaunchKernelGGL(k_dense_finish_tiled<float>, dim3(tiles), dim3(256), 0, st, D.p, ld, n, inv32->p, ldo);
    } else {
        FS_HIP(inv64->alloc((size_t)n * ldo));
        hipLaunchKernelGGL(k_dense_finish_tiled<double>, dim3(tiles), dim3(256), 0, st, D.p, ld, n, inv64->p, ldo);
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
        // issued on the matrix cores: per sweep every lower tile 2 x 64 x 64 x 128 and every block row's panel 2 x 64 x 128 x 128
        stats->mfma_flops = (double)ns * ((double)tiles * 2.0 * kNB * kNB * kSW + (double)nt * 2.0 * kNB * kSW * kSW);
        stats->useful_flops = (double)n * n * n;                           // n^3 of a symmetric inversion
        stats->dropped = hstatus[1];
        stats->bytes = (double)ns * (double)tiles * 2.0 * kNB * kNB * 8.0; // lower triangle read + written per sweep
    }
    if (hstatus[0] == 2) {
        *timed_out = true;
        return set_err(FEMSHELL_ERR_HIP, "multigrid setup: the look-ahead of the dense inverse timed out");
    }
    if (hstatus[0] != 0) return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: coarsest operator is not positive definite");
    return FEMSHELL_OK;
}

} // namespace

int amg_dense_inverse_device(femshell_ctx *c, const Bsr &A, bool single_precision, DevBuf<double> *inv64, DevBuf<float> *inv32,
                             int64_t *lda_out, AmgDenseStats *stats)
{
    // FEMSHELL_AMG_DENSE_LOOKAHEAD=0: the pivot inverse as a launch of its own in front of every sweep (A/B runs)
    const bool lookahead = !(getenv("FEMSHELL_AMG_DENSE_LOOKAHEAD") && atoi(getenv("FEMSHELL_AMG_DENSE_LOOKAHEAD")) == 0);
    bool timed_out = false;
    int rc = dense_inverse_once(c, A, single_precision, inv64, inv32, lda_out, stats, lookahead, &timed_out);
    if (rc && timed_out && lookahead) {
        // the look-ahead's workgroup was not scheduled beside the three it waits for (shared card, CU mask): once more, without it
        if (getenv("FEMSHELL_AMG_VERBOSE") && atoi(getenv("FEMSHELL_AMG_VERBOSE")) != 0)
            fprintf(stderr, "[femshell amg setup] the look-ahead of the dense inverse timed out: running it again without\n");
        inv64->release();
        inv32->release();
        rc = dense_inverse_once(c, A, single_precision, inv64, inv32, lda_out, stats, false, &timed_out);
    }
    return rc;
}

// (MEASURED, round 5, and dropped: the product from the lower triangle of the symmetric inverse alone -- one workgroup per 64 x 64
//  tile, the tile used for its block row and, through lane shuffles and LDS, for its block column, two-stage sums in a fixed
//  order: 109 MB + 14 MB of partial sums instead of 218 MB, and 5 % MORE time per solve of the 4M panel (0.632 against 0.600 s): the
//  sixty-four shuffles a thread spends on the transposed sums and 6786 workgroups of 16 KB cost more than the bytes they save.)
void launch_dense_gemv_big(const double *A64, const float *A32, int64_t lda, const double *b, double *y, int32_t n, int32_t n_pad6,
                           const CgScalars *gate, hipStream_t st)
{
    const dim3 g((unsigned)((n_pad6 + 3) / 4)), blk(256);
    if (A32 != nullptr) hipLaunchKernelGGL(k_dense_gemv_big<float>, g, blk, 0, st, A32, lda, b, y, n, n_pad6, gate);
    else hipLaunchKernelGGL(k_dense_gemv_big<double>, g, blk, 0, st, A64, lda, b, y, n, n_pad6, gate);
}

} // namespace femshell
