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
#include <chrono>
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
// (panels: the four 64 x 66 work blocks P0 .. P3; small: 2 x 4 x 64 doubles of row panel + 128 diagonal entries -- LDS, both.)
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
        // (only over a clean status: a bound that expired elsewhere -- status 2, the inverse runs again without the look-ahead -- has
        //  left this sweep with blocks that were never finished, and their "negative pivots" must not turn the timeout into a
        //  breakdown.  Round 6: two ranks sharing one card in tests/test_multirank_gpu.py, one run in four)
        if (n_failed) {
            int32_t clean = 0;
            (void)__hip_atomic_compare_exchange_strong(&status[0], &clean, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        atomicAdd(&status[1], n_dead);
    }
}

// Bounded wait of one workgroup for a counter another launch raises (the look-ahead of the block sweep, below): thread 0 polls
// with acquire semantics, the others join at the barrier; false when the bound expired (status[0] = 2: the host runs the inverse
// again without the look-ahead).
// kFenceAlways = false: for a consumer whose producer is a launch that (nearly always) ended before this one began -- the panel
// kernel waiting for B.  The full __threadfence (acquire + release: an L2 write-back and an invalidate by each of the 244
// workgroups of the panel kernel at once, 15 us per sweep) is paid only when the counter was NOT yet up at the first look;
// otherwise an acquire-only fence at agent scope, which invalidates and writes nothing back.  (Round 5 paid nothing at all on
// that path and relied on the cache invalidate at the start of a launch: runtime behaviour, not a guarantee -- ADVICE r5.)
template <bool kFenceAlways = true>
__device__ __forceinline__ bool wait_for_counter(const unsigned int *counter, unsigned int need, int spin_limit, int32_t *status)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int spins = 0;
        while (spins < spin_limit && __hip_atomic_load(counter, kFenceAlways ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            // (somebody else's bound has expired already: the inverse runs again anyway, nobody waits out a bound of his own)
            if ((spins & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) spins = spin_limit - 1;
            __builtin_amdgcn_s_sleep(8);
            spins++;
        }
        ok = spins >= spin_limit ? 0 : (spins == 0 ? 1 : 2);
        if (!ok) __hip_atomic_store(status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (kFenceAlways || ok == 2) __threadfence(); // (acquire for every thread's loads of what the counter announces)
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // counter already up at the first look: an acquire-ONLY fence -- it
                                                            // invalidates, it does not write L2 back (the 15 us were __threadfence's release half)
    return ok != 0;
}

// Do the context's two streams run side by side?  HIP maps streams onto a handful of hardware queues (four by default): in a
// process with many streams two of them can share one, and a launch that waits for a launch BEHIND it in the same queue waits
// for its bound.  Asked once per context, with the question itself: a launch on the second stream that waits (briefly) for a
// counter, a launch on the first that raises it.
__global__ void k_streams_probe_wait(unsigned int *flag, int spins, int32_t *seen)
{
    int n = 0;
    while (n < spins && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(8);
        n++;
    }
    *seen = n < spins ? 1 : 0;
}
__global__ void k_streams_probe_raise(unsigned int *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait_tiles != nullptr: the look-ahead's launch on the second stream -- the inverse of pivot block K starts when the update of
// sweep K - 1 (first stream, same time) has counted its `need` tiles in place, and `done` announces B to the panel kernel.
__global__ __launch_bounds__(256) void k_dense_pivot(const double *__restrict__ D, int64_t ld, int K, const double *__restrict__ diag0,
                                                     double *__restrict__ B, int32_t *status, const unsigned int *wait_tiles,
                                                     unsigned int need, int spin_limit, unsigned int *done)
{
    extern __shared__ double lds_dense[];
    const bool go = wait_tiles == nullptr || wait_for_counter(wait_tiles, need, spin_limit, status);
    if (go) pivot_block(D, ld, K, diag0, B, status, lds_dense, lds_dense + 4 * kNB * kLdp);

    if (done != nullptr) { // (raised on a failed wait as well: nobody behind this launch waits for its full bound too)
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// C_i = A(i, block K) (64 x 128), gathered from the lower triangle (tiles (i, 2K), (i, 2K+1) below the pivot block, the
// transposed tiles (2K, i), (2K+1, i) above it), W_i = C_i B on the matrix cores.  One workgroup per 64-row block; the two
// block rows of the pivot block itself are skipped.  K runs in four chunks of 32 through LDS (row stride 34 doubles: the 32
// lanes of a ds_read_b64 group hit 32 different bank pairs); B is symmetric, so W = C B = C (B^T)^T has the X Y^T form and
// both operands are read the same way.  Wave w computes the output columns [32 w, 32 w + 32) of all 64 rows.
// (kSplit: two workgroups per 64-row block, 32 rows each -- 2 x 116 workgroups instead of 116 on 256 CUs; the panel product
//  sits between the pivot inverse and the trailing update of every sweep, on the critical path)
typedef double word16 __attribute__((ext_vector_type(2)));
// ---- load batches.  Every load of a batch AND the s_waitcnt that completes them are ONE asm statement, its outputs early-clobber:
// for the compiler no output register exists before the wait, so it cannot copy, split or spill one while the load that fills it
// is still in flight (the hardware does not interlock VMEM returns against such a move), and no output shares a register with
// an address operand.  (Round 5 had one asm per load and the wait in a third: correct only as long as register allocation left
// the outputs alone between them.)  Addresses: a uniform 64-bit base in SGPRs, one 32-bit byte offset per thread in a VGPR that
// the statement advances by a uniform stride between the loads -- the q-th word of a thread lies q strides behind its first
// in every batch of this file -- so a batch of sixteen 16-byte loads needs 16 + 4 operands (the limit is 30).
// See stage_chunk for why these are assembly at all.
#define FS_LDW(o, t, b) "global_load_dwordx4 %[" #o "], %[" #t "], %[" #b "]\n\t"
#define FS_LDD(o, t, b) "global_load_dwordx2 %[" #o "], %[" #t "], %[" #b "]\n\t"
#define FS_ADV(t, s) "v_add_u32 %[" #t "], %[" #s "], %[" #t "]\n\t"
#define FS_B8_LOADS                                                                                                              \
    FS_LDW(b0, bt, bb) FS_ADV(bt, bs) FS_LDW(b1, bt, bb) FS_ADV(bt, bs) FS_LDW(b2, bt, bb) FS_ADV(bt, bs) FS_LDW(b3, bt, bb)      \
    FS_ADV(bt, bs) FS_LDW(b4, bt, bb) FS_ADV(bt, bs) FS_LDW(b5, bt, bb) FS_ADV(bt, bs) FS_LDW(b6, bt, bb) FS_ADV(bt, bs)          \
    FS_LDW(b7, bt, bb)
#define FS_B8_OUTS [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3]), [b4] "=&v"(b[4]), [b5] "=&v"(b[5]), [b6] "=&v"(b[6]), [b7] "=&v"(b[7])

// eight words of B^T and kC words of a tile's rows, all in flight together, then waited for
template <int kC>
__device__ __forceinline__ void load_b8_words_wait(word16 (&b)[8], word16 (&c)[kC], const double *bbase, uint32_t boff, uint32_t bstride,
                                                   const double *cbase, uint32_t coff, uint32_t cstride)
{
    static_assert(kC == 2 || kC == 4, "batch sizes of k_dense_panels");
    if constexpr (kC == 2)
        asm volatile(FS_B8_LOADS FS_LDW(c0, ct, cb) FS_ADV(ct, cs) FS_LDW(c1, ct, cb) "s_waitcnt vmcnt(0)"
                     : FS_B8_OUTS, [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [bt] "+v"(boff), [ct] "+v"(coff)
                     : [bb] "s"(bbase), [cb] "s"(cbase), [bs] "s"(bstride), [cs] "s"(cstride)
                     : "memory");
    else
        asm volatile(FS_B8_LOADS FS_LDW(c0, ct, cb) FS_ADV(ct, cs) FS_LDW(c1, ct, cb) FS_ADV(ct, cs) FS_LDW(c2, ct, cb) FS_ADV(ct, cs)
                         FS_LDW(c3, ct, cb) "s_waitcnt vmcnt(0)"
                     : FS_B8_OUTS, [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [bt] "+v"(boff), [ct] "+v"(coff)
                     : [bb] "s"(bbase), [cb] "s"(cbase), [bs] "s"(bstride), [cs] "s"(cstride)
                     : "memory");
}
// the same with kC single doubles of a transposed tile
template <int kC>
__device__ __forceinline__ void load_b8_doubles_wait(word16 (&b)[8], double (&c)[kC], const double *bbase, uint32_t boff, uint32_t bstride,
                                                     const double *cbase, uint32_t coff, uint32_t cstride)
{
    static_assert(kC == 4 || kC == 8, "batch sizes of k_dense_panels");
    if constexpr (kC == 4)
        asm volatile(FS_B8_LOADS FS_LDD(c0, ct, cb) FS_ADV(ct, cs) FS_LDD(c1, ct, cb) FS_ADV(ct, cs) FS_LDD(c2, ct, cb) FS_ADV(ct, cs)
                         FS_LDD(c3, ct, cb) "s_waitcnt vmcnt(0)"
                     : FS_B8_OUTS, [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [bt] "+v"(boff), [ct] "+v"(coff)
                     : [bb] "s"(bbase), [cb] "s"(cbase), [bs] "s"(bstride), [cs] "s"(cstride)
                     : "memory");
    else
        asm volatile(FS_B8_LOADS FS_LDD(c0, ct, cb) FS_ADV(ct, cs) FS_LDD(c1, ct, cb) FS_ADV(ct, cs) FS_LDD(c2, ct, cb) FS_ADV(ct, cs)
                         FS_LDD(c3, ct, cb) FS_ADV(ct, cs) FS_LDD(c4, ct, cb) FS_ADV(ct, cs) FS_LDD(c5, ct, cb) FS_ADV(ct, cs)
                             FS_LDD(c6, ct, cb) FS_ADV(ct, cs) FS_LDD(c7, ct, cb) "s_waitcnt vmcnt(0)"
                     : FS_B8_OUTS, [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]),
                       [c6] "=&v"(c[6]), [c7] "=&v"(c[7]), [bt] "+v"(boff), [ct] "+v"(coff)
                     : [bb] "s"(bbase), [cb] "s"(cbase), [bs] "s"(bstride), [cs] "s"(cstride)
                     : "memory");
}
// kW words of each of two operands that share their offsets (the W and C panels of the trailing update)
template <int kW>
__device__ __forceinline__ void load_pairs_wait(word16 (&w)[kW], word16 (&c)[kW], const double *wbase, const double *cbase, uint32_t off,
                                                uint32_t stride)
{
    static_assert(kW == 4 || kW == 8, "batch sizes of stage_chunk");
#define FS_PAIR(q) FS_LDW(w##q, t, wb) FS_LDW(c##q, t, cb)
    if constexpr (kW == 4)
        asm volatile(FS_PAIR(0) FS_ADV(t, st) FS_PAIR(1) FS_ADV(t, st) FS_PAIR(2) FS_ADV(t, st) FS_PAIR(3) "s_waitcnt vmcnt(0)"
                     : [w0] "=&v"(w[0]), [w1] "=&v"(w[1]), [w2] "=&v"(w[2]), [w3] "=&v"(w[3]), [c0] "=&v"(c[0]), [c1] "=&v"(c[1]),
                       [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [t] "+v"(off)
                     : [wb] "s"(wbase), [cb] "s"(cbase), [st] "s"(stride)
                     : "memory");
    else
        asm volatile(FS_PAIR(0) FS_ADV(t, st) FS_PAIR(1) FS_ADV(t, st) FS_PAIR(2) FS_ADV(t, st) FS_PAIR(3) FS_ADV(t, st) FS_PAIR(4)
                         FS_ADV(t, st) FS_PAIR(5) FS_ADV(t, st) FS_PAIR(6) FS_ADV(t, st) FS_PAIR(7) "s_waitcnt vmcnt(0)"
                     : [w0] "=&v"(w[0]), [w1] "=&v"(w[1]), [w2] "=&v"(w[2]), [w3] "=&v"(w[3]), [w4] "=&v"(w[4]), [w5] "=&v"(w[5]),
                       [w6] "=&v"(w[6]), [w7] "=&v"(w[7]), [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]),
                       [c4] "=&v"(c[4]), [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7]), [t] "+v"(off)
                     : [wb] "s"(wbase), [cb] "s"(cbase), [st] "s"(stride)
                     : "memory");
#undef FS_PAIR
}

// (ready != nullptr: B comes from the look-ahead's launch on the second stream, which raises *ready when it is complete)
template <bool kSplit>
__global__ __launch_bounds__(256) void k_dense_panels(const double *D, int64_t ld, int K, const double *B, double *__restrict__ Cp,
                                                      double *__restrict__ Wp, const unsigned int *ready, int spin_limit, int32_t *status)
{
    constexpr int kRows = kSplit ? kNB / 2 : kNB; // rows of C_i this workgroup takes
    __shared__ double Cs[kRows * kLdh];
    __shared__ double Bs[kSW * kLdh];
    const int i = kSplit ? blockIdx.x >> 1 : blockIdx.x, row0 = kSplit ? (blockIdx.x & 1) * kRows : 0, tid = threadIdx.x;
    if ((i >> 1) == K) return;
    if (ready != nullptr && !wait_for_counter<false>(ready, 1u, spin_limit, status)) return;
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
        // B^T[k][n] = B[n][k]: 128 rows x 16 words of 16 bytes, eight per thread, and the rows of C_i beside them -- ONE batch, all
        // in flight at once (as scalar loads in a loop they were sixteen round trips to memory per chunk: the whole 31 us of
        // this kernel).  Word q of a thread: e = tid + 256 q, row e / 16 = tid / 16 + 16 q, column 2 (tid % 16).
        constexpr int kBWords = kSW * (kHalf / 2) / 256;
        static_assert(kBWords == 8, "load_b8_*");
        word16 bw[kBWords];
        const double *bbase = B + kHalf * ch;
        const uint32_t boff = (uint32_t)(((tid >> 4) * kSW + 2 * (tid & 15)) * 8), bstride = 16u * kSW * 8u;
        if (below) { // rows of the tile: 16 words per row
            constexpr int kCWords = kRows * (kHalf / 2) / 256;
            word16 cw[kCWords];
            const double *cbase = D + ((int64_t)i * kNB + row0) * ld + (int64_t)2 * K * kNB + kHalf * ch;
            load_b8_words_wait<kCWords>(bw, cw, bbase, boff, bstride, cbase, (uint32_t)(((int64_t)(tid >> 4) * ld + 2 * (tid & 15)) * 8),
                                        (uint32_t)(16 * ld * 8));
#pragma unroll
            for (int q = 0; q < kCWords; q++) {
                const int e = tid + 256 * q, r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
                *reinterpret_cast<word16 *>(Cs + r * kLdh + k) = cw[q];
                *reinterpret_cast<word16 *>(Cp + ((int64_t)i * kNB + row0 + r) * kSW + kHalf * ch + k) = cw[q];
            }
        } else { // above the pivot block: the transposed tiles (2K, i), (2K+1, i), read along their rows (consecutive threads:
                 // consecutive addresses), eight bytes at a time: e = tid + 256 q, r = e % kRows, k = e / kRows = tid / kRows + (256 / kRows) q
            constexpr int kCount = kRows * kHalf / 256;
            double cd[kCount];
            const double *cbase = D + ((int64_t)2 * K * kNB + kHalf * ch) * ld + (int64_t)i * kNB + row0;
            load_b8_doubles_wait<kCount>(bw, cd, bbase, boff, bstride, cbase, (uint32_t)(((int64_t)(tid / kRows) * ld + tid % kRows) * 8),
                                         (uint32_t)((256 / kRows) * ld * 8));
#pragma unroll
            for (int q = 0; q < kCount; q++) {
                const int e = tid + 256 * q, r = e % kRows, k = e / kRows;
                Cs[r * kLdh + k] = cd[q];
                Cp[((int64_t)i * kNB + row0 + r) * kSW + kHalf * ch + k] = cd[q];
            }
        }
#pragma unroll
        for (int q = 0; q < kBWords; q++) {
            const int e = tid + 256 * q, n = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
            *reinterpret_cast<word16 *>(Bs + n * kLdh + k) = bw[q];
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

// Look-ahead of the block sweep (la.on; FEMSHELL_AMG_DENSE_LOOKAHEAD=0 switches it off).  The one-workgroup pivot inverse is 71 us
// of dependent steps per sweep (4.1 of 17.2 ms at 7386 dofs) in front of everything else.  The tiles of the NEXT pivot block are
// therefore the first workgroups of the update's grid; they count themselves done in la.flag (release), and the inverse of that
// block -- k_dense_pivot, launched on a SECOND stream before the update it belongs to -- waits for that count (acquire, bounded),
// runs beside the rest of the update and raises a second flag the next sweep's panel kernel waits for.  No events: the two
// streams meet through the two counters only (round 3's version with two event waits per sweep: 29.7 ms).
// History.  Rounds 4-5 ran the inverse INSIDE the update kernel (a fourth special workgroup, pivot_block_lean: work blocks in HBM
// scratch, products staged through the update's 35 KB): 17.1 -> 13.2 ms.  Measured in round 5 (rocprofv3, 7776 dofs): that code
// costs the update itself -- 194 registers wanted, 128 granted at four workgroups per CU, 348 bytes of scratch per lane: 171 us
// per sweep for the plain instantiation, 194 us for the one that carries the pivot code with the look-ahead switched off, 215 us
// with it on -- whatever the pivot workgroup did (skipping its work altogether: 205 us).  A launch of its own gives the inverse
// 138 KB of LDS and its registers without taking either from the 7503 tile workgroups.
struct DenseLookAhead {
    int on = 0;
    unsigned int *flag = nullptr; // one counter per sweep, zero-initialised: tiles of the next pivot block in place
};

// Workgroup -> tile of the trailing update, by XCD.  A tile (i, j) reads the operand panels W_i and C_j (64 KB each) besides its
// own 32 KB, and the two panel arrays (15 MB at 7.4k dofs) do not fit the 4 MB L2 of an XCD: with tiles handed out in linear
// order -- consecutive workgroups on eight different XCDs -- every XCD streams every panel from the Infinity Cache over and over,
// 870 MB per sweep beside the 436 MB of the tiles themselves, and the update runs at the rate of that traffic (8 TB/s), not of
// the matrix cores.  Here the triangle is cut into super-blocks of 8 x 8 tiles; a super-block belongs to ONE XCD (the workgroups
// with the same blockIdx % 8 share one: cdna_hip_programming.md T1), whose 128 resident workgroups are then two super-blocks:
// 16 + 16 panels = 2 MB in its L2, each fetched once per super-block.  The host deals the super-blocks out by weight (diagonal
// and last-row blocks have fewer tiles) and lists them per XCD in row-major order (DenseTileMap::list: (I << 16) | J, -1 = none).
struct DenseTileMap {
    const int32_t *list = nullptr; // [8][per_xcd]
    int per_xcd = 0;               // super-blocks per XCD (the longest list)
    int nt = 0;                    // 64-tiles per side
    int linear = 0;                // FEMSHELL_AMG_DENSE_XCD_MAP=0: tiles in linear order of the triangle (A/B runs)
};
constexpr int kSb = 8; // tiles per side of a super-block

constexpr int kUpdLds = 2 * kNB * kLdh > kNB * kLdp ? 2 * kNB * kLdh : kNB * kLdp; // doubles of one operand buffer pair

// Columns [32 h, 32 h + 32) of the rows of W_i and C_j into their LDS panels (row stride 34 doubles): kRows x 16 words of 16 bytes
// per operand, kRows / 16 per thread and operand -- ALL of them in flight before the first is waited for.  The loads are inline
// assembly because nothing else kept them together: written as plain loads in front of the LDS writes (a loop, unrolled, arrays
// of registers, sched_group_barrier, amdgpu_waves_per_eu) the compiler's scheduler turns them into load, wait, LDS write, eight
// times over -- eight memory round trips per chunk instead of one.  One box, 7776 dofs, alternating: 13.5 ms (loop) / 13.2 ms
// (unrolled, compiler's order) / 12.1 ms (this) for the whole inverse.
// (The compiler does not count these loads in its own s_waitcnt bookkeeping; they are waited for, all of them, inside the one asm
//  statement that issues them -- load_pairs_wait above -- and loads the compiler issued earlier only complete earlier than it assumes.)
// MEASURED besides (round 5, per-phase clocks inside the workgroups): a tile workgroup lives 24 us, of which its four matrix
// phases are 10.6 (2.6 each: 0.85 alone, the rest is the pipe shared with the three other workgroups of the CU), staging 7.2,
// barriers 2.7, the tile's own values 3.0 before and 0.75 behind.  Built, timed and dropped: global_load_lds_dword straight into the
// padded rows (13.6 ms against 11.6 on that box: 4-byte transfers), the next chunk travelling behind the matrix instructions
// into a second LDS buffer at two workgroups per CU (14.0 ms), the same on 128 x 128 tiles (13.9 ms), and a fixed order of
// precedence (s_setprio) among the four workgroups of a CU against their running in step (12.4 against 12.0 ms).  Memory latency
// under load (1.5 - 2.5 us per 32 KB chunk) against 0.85 us of matrix work per chunk and wave is the shape of the problem:
// hiding it takes three or four chunks in flight per workgroup, a ring this kernel does not have.
template <int kRows = kNB>
__device__ __forceinline__ void stage_chunk(double *Ws, double *Cs, const double *__restrict__ wsrc, const double *__restrict__ csrc,
                                            int h, int tid)
{
    constexpr int kWords = kRows * (kHalf / 2) / 256; // per thread and operand
    word16 wv[kWords], cv[kWords];
    // word q of a thread: e = tid + 256 q, row e / 16 = tid / 16 + 16 q, column 2 (tid % 16) -- 16 rows of the panel further per q
    load_pairs_wait<kWords>(wv, cv, wsrc + h * kHalf, csrc + h * kHalf, (uint32_t)(((tid >> 4) * kSW + 2 * (tid & 15)) * 8), 16u * kSW * 8u);
#pragma unroll
    for (int q = 0; q < kWords; q++) {
        const int e = tid + 256 * q, r = e / (kHalf / 2), k = 2 * (e % (kHalf / 2));
        *reinterpret_cast<word16 *>(Ws + r * kLdh + k) = wv[q];
        *reinterpret_cast<word16 *>(Cs + r * kLdh + k) = cv[q];
    }
}

// one workgroup per lower 64 x 64 tile (i >= j): the sweep of the 128-wide block K
// (kLookAhead: the first three workgroups take the tiles of the next pivot block and count themselves done)
template <bool kLookAhead>
__global__ __launch_bounds__(256, 4) void k_dense_update(double *__restrict__ D, int64_t ld, int K, const double *__restrict__ B,
                                                                     const double *__restrict__ Cp, const double *__restrict__ Wp,
                                                                     DenseTileMap map, DenseLookAhead la)
{
    extern __shared__ double lds_upd[]; // kUpdLds doubles
    double *Ws = lds_upd, *Cs = lds_upd + kNB * kLdh;
    const int tid = threadIdx.x;
    // workgroup -> tile.  With the look-ahead: workgroups 0, 1, 2 take the tiles (2K+2, 2K+2), (2K+3, 2K+2), (2K+3, 2K+3) of the
    // next pivot block, the map starts at workgroup 3 and whoever it sends to one of the three tiles leaves (every tile still
    // has exactly one workgroup)
    int m = blockIdx.x, i = 0, j = 0;
    bool ahead_tile = false;
    if (kLookAhead && la.on) {
        const int i2 = 2 * K + 2;
        if (blockIdx.x < 3) {
            i = blockIdx.x == 0 ? i2 : i2 + 1;
            j = blockIdx.x == 2 ? i2 + 1 : i2;
            ahead_tile = true;
        }
        m = (int)blockIdx.x - 3;
    }
    if (!ahead_tile) {
        // (m & 7 labels the workgroups that share an XCD as blockIdx & 7 does: the offset of 3 permutes the labels)
        if (map.linear) {
            if (m >= map.nt * (map.nt + 1) / 2) return;
            i = (int)((sqrt(8.0 * m + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= m) i++;
            while (i * (i + 1) / 2 > m) i--;
            j = m - i * (i + 1) / 2;
        } else {
            const int idx = m >> 3, sb = map.list[(m & 7) * map.per_xcd + idx / (kSb * kSb)], tin = idx % (kSb * kSb);
            if (sb < 0) return;
            i = kSb * (sb >> 16) + tin / kSb;
            j = kSb * (sb & 0xffff) + tin % kSb;
            if (i >= map.nt || j > i) return;
        }
        if (kLookAhead && la.on && (i >> 1) == K + 1 && (j >> 1) == K + 1) return; // (the first three workgroups took these)
    }
    auto done = [&]() { // a tile of the next pivot block is in place
        if (ahead_tile) {
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(la.flag + K, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
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
                // (MEASURED, round 5: kept in registers of their own until the end -- A - W C^T -- they are not waited for before the
                //  first operand loads go out; from the start that costs 32 registers beside the 32 of the words in flight and spills
                //  at four workgroups per CU (202 against 180 us per sweep); fetched in front of the last chunk's matrix instructions
                //  into the registers the staging has given back: no spill, and no gain either -- 11.07 against 11.07 - 11.10 ms)
                acc[ti][tj][g] = -tile[(int64_t)r * ld + c]; // accumulate W C^T - A, store its negative
            }
    // K = 128 in four chunks of 32 (row stride 34 doubles: conflict-free): 35 KB of LDS per workgroup, four per CU
    const double *wsrc = Wp + (int64_t)(i * kNB) * kSW, *csrc = Cp + (int64_t)(j * kNB) * kSW;
    {
        for (int h = 0; h < kSW / kHalf; h++) {
            if (h) __syncthreads(); // the previous chunk's readers are through
            stage_chunk(Ws, Cs, wsrc, csrc, h, tid);
            __syncthreads();
            quadrant_xyt<kHalf, kLdh>(Ws, Cs, r0, c0, lane, acc);
        }
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
// Look-ahead: workgroup 0 takes the next pivot block's tile -- ONE tile now.
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
        const int K1 = K + 1, t_a = K1 * (K1 + 1) / 2 + K1; // the next pivot block's tile: workgroup 0 takes it, and whoever
        if (blockIdx.x == 0) {                              // lands on it takes the tile of the workgroup that left (0)
            t = t_a;
            ahead_tile = true;
        } else if (t == t_a) {
            t = 0;
        }
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
        stage_chunk<kSW>(Ws, Cs, wsrc, csrc, h, tid);
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

int amg_dense_probe_streams(femshell_ctx *c)
{
    hipStream_t st = c->stream;
    const bool verbose = getenv("FEMSHELL_AMG_VERBOSE") && atoi(getenv("FEMSHELL_AMG_VERBOSE")) != 0;
    DevBuf<unsigned int> flag;
    DevBuf<int32_t> seen;
    FS_HIP(flag.alloc(1));
    FS_HIP(seen.alloc(1));
    // (a second stream that turns out to share the first one's hardware queue: three more are made, all alive at once so that they
    //  sit on different queues, and the first that runs beside the main stream is kept)
    hipStream_t candidates[4] = {c->aux_stream, nullptr, nullptr, nullptr};
    int kept = -1;
    for (int k = 0; k < 4 && kept < 0; k++) {
        if (k == 1)
            for (int q = 1; q < 4; q++) FS_HIP(hipStreamCreateWithFlags(&candidates[q], hipStreamNonBlocking));
        if (!candidates[k]) FS_HIP(hipStreamCreateWithFlags(&candidates[k], hipStreamNonBlocking));
        const double t_probe = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        FS_HIP(flag.zero(st));
        FS_HIP(seen.zero(st));
        FS_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_streams_probe_wait, dim3(1), dim3(1), 0, candidates[k], flag.p, 8000, seen.p); // (a few milliseconds at most)
        hipLaunchKernelGGL(k_streams_probe_raise, dim3(1), dim3(1), 0, st, flag.p);
        FS_HIP(hipStreamSynchronize(candidates[k]));
        FS_HIP(hipStreamSynchronize(st));
        int32_t h = 0;
        FS_HIP(hipMemcpy(&h, seen.p, sizeof h, hipMemcpyDeviceToHost));
        if (verbose)
            fprintf(stderr, "[femshell] do the context's two streams run side by side: %s (asked in %.3f ms)\n", h ? "yes" : "no",
                    1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_probe));
        if (h) kept = k;
    }
    c->aux_streams_side_by_side = kept >= 0 ? 1 : -1;
    c->aux_stream = candidates[kept >= 0 ? kept : 0];
    for (int k = 0; k < 4; k++)
        if (candidates[k] && candidates[k] != c->aux_stream) FS_HIP(hipStreamDestroy(candidates[k]));
    if (c->aux_streams_side_by_side < 0) {
        if (verbose) fprintf(stderr, "[femshell] no second stream on a hardware queue of its own: dense inverse without its look-ahead\n");
    }
    return FEMSHELL_OK;
}

namespace {

// (*timed_out: the look-ahead's bounded wait expired -- the caller runs the inverse again without it)
int dense_inverse_once(femshell_ctx *c, const Bsr &A, bool single_precision, DevBuf<double> *inv64, DevBuf<float> *inv32,
                       int64_t *lda_out, AmgDenseStats *stats, bool lookahead, bool *timed_out)
{
    hipStream_t st = c->stream;
    *timed_out = false;
    const int n = 6 * A.nr, n_pad = (n + kSW - 1) / kSW * kSW, nt = n_pad / kNB, ns = n_pad / kSW;
    const int64_t ld = n_pad;
    if (lookahead && ns > 1 && c->aux_streams_side_by_side == 0) { // (femshell_create asked; a context made some other way asks here)
        const int rc = amg_dense_probe_streams(c);
        if (rc) return rc;
    }
    if (lookahead && c->aux_streams_side_by_side < 0) lookahead = false;
    DevBuf<double> D, diag0, B, Cp, Wp;
    DevBuf<unsigned int> la_flags;
    DevBuf<int64_t> dptr;
    DevBuf<int32_t> dcol, dstatus;
    DevBuf<double> dval;
    FS_HIP(D.alloc((size_t)n_pad * n_pad));
    FS_HIP(D.zero(st));
    FS_HIP(diag0.alloc(n_pad));
    FS_HIP(B.alloc(2 * kSW * kSW)); // (two: the look-ahead writes the next sweep's while this sweep's is read)
    // counters of the look-ahead: [0, ns] tiles of the next pivot block in place (per sweep), [ns + 1, 2 ns + 1] B of a sweep complete
    FS_HIP(la_flags.alloc(2 * (size_t)ns + 2));
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
    // super-blocks of the trailing update, dealt out to the eight XCDs by weight (DenseTileMap)
    DevBuf<int32_t> d_sb_list;
    DenseTileMap tmap;
    {
        const int nsbr = (nt + kSb - 1) / kSb;
        struct Sb { int I, J, weight; };
        std::vector<Sb> sbs;
        for (int I = 0; I < nsbr; I++)
            for (int J = 0; J <= I; J++) {
                int w = 0;
                for (int a = 0; a < kSb; a++)
                    for (int b = 0; b < kSb; b++) w += (kSb * I + a < nt && kSb * J + b <= kSb * I + a) ? 1 : 0;
                sbs.push_back({I, J, w});
            }
        std::stable_sort(sbs.begin(), sbs.end(), [](const Sb &x, const Sb &y) { return x.weight > y.weight; });
        std::vector<std::vector<Sb>> lists(8);
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const Sb &b : sbs) {
            const int x = (int)(std::min_element(load, load + 8) - load);
            lists[(size_t)x].push_back(b);
            load[x] += b.weight;
        }
        size_t longest = 1;
        for (auto &l : lists) {
            std::sort(l.begin(), l.end(), [](const Sb &x, const Sb &y) { return x.I != y.I ? x.I < y.I : x.J < y.J; });
            longest = std::max(longest, l.size());
        }
        std::vector<int32_t> flat(8 * longest, -1);
        for (int x = 0; x < 8; x++)
            for (size_t k = 0; k < lists[(size_t)x].size(); k++) flat[(size_t)x * longest + k] = (lists[(size_t)x][k].I << 16) | lists[(size_t)x][k].J;
        FS_HIP(d_sb_list.upload(flat, st));
        tmap.list = d_sb_list.p;
        tmap.per_xcd = (int)longest;
        tmap.nt = nt;
        tmap.linear = (getenv("FEMSHELL_AMG_DENSE_XCD_MAP") && atoi(getenv("FEMSHELL_AMG_DENSE_XCD_MAP")) == 0) ? 1 : 0;
    }
    const unsigned update_grid = 8u * (unsigned)tmap.per_xcd * kSb * kSb;
    const size_t lds_upd_bytes = (size_t)kUpdLds * sizeof(double);
    // LDS of the pivot kernel: four 64 x 66 blocks, the row panel of the sweeps, 128 diagonal entries
    const size_t lds_pivot = (4 * (size_t)kNB * kLdp + 2 * 4 * kNB + 2 * kNB) * sizeof(double);
    FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_pivot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pivot));
    // Look-ahead: the pivot launch on the second stream waits (bounded) for three workgroups of the update on the first, and the
    // next panel kernel for the pivot launch.  That makes progress because the pivot launch is enqueued in front of the update it
    // waits for and the two streams run side by side -- how HIP maps streams to hardware queues, not a contract: if the streams
    // were serialised, or on a card shared with other processes, a bound expires.  The waiting kernel then marks status 2 and
    // amg_dense_inverse_device runs the whole inverse again with the pivot as a launch in front of every sweep (slower, same
    // numbers).
    const bool panel_split = !(getenv("FEMSHELL_AMG_DENSE_PANEL_SPLIT") && atoi(getenv("FEMSHELL_AMG_DENSE_PANEL_SPLIT")) == 0);
    // FEMSHELL_AMG_DENSE_TILE=128: the trailing update on 128 x 128 tiles (k_dense_update128; default: 64 x 64, four workgroups
    // per CU).  MEASURED, round 5, 7776 dofs, alternating on one box, both with the look-ahead on the second stream: 15.0 ms
    // against 13.5 ms (before the loads of a chunk were kept together: stage_chunk) -- half the operand traffic and four times
    // the matrix instructions per LDS read buy nothing where the workgroups wait for memory latency, and the larger tile has 1891
    // workgroups for 512 slots (a fourth, nearly empty round).  The measured alternative stays, and stays under test.
    const bool big_tiles = getenv("FEMSHELL_AMG_DENSE_TILE") && atoi(getenv("FEMSHELL_AMG_DENSE_TILE")) == 128;
    const int stiles = ns * (ns + 1) / 2;
    const size_t lds_big_bytes = 2 * (size_t)kSW * kLdh * sizeof(double);
    if (big_tiles) {
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_update128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big_bytes));
        FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_update128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big_bytes));
    }
    // the look-ahead's second stream: made once per context; it meets the first through the counters (and one event, here: the
    // counters are zero before anything polls them)
    int spin_limit = 1 << 22; // polls before a waiting workgroup gives up (seconds) (FEMSHELL_AMG_DENSE_LOOKAHEAD_SPINS: tests)
    if (const char *e = getenv("FEMSHELL_AMG_DENSE_LOOKAHEAD_SPINS")) spin_limit = atoi(e);
    unsigned int *tiles_in_place = la_flags.p, *b_complete = la_flags.p + ns + 1;
    if (lookahead && c->aux_streams_side_by_side < 0) lookahead = false;
    if (lookahead && ns > 1) {
        hipEvent_t zeroed;
        FS_HIP(hipEventCreateWithFlags(&zeroed, hipEventDisableTiming));
        FS_HIP(hipEventRecord(zeroed, st));
        FS_HIP(hipStreamWaitEvent(c->aux_stream, zeroed, 0));
        (void)hipEventDestroy(zeroed);
    }
    const unsigned int need_tiles = big_tiles ? 1u : 3u;
    for (int K = 0; K < ns; K++) {
        double *Bk = B.p + (size_t)(K & 1) * kSW * kSW;
        const bool b_from_aux = lookahead && K > 0; // (this sweep's B: by the launch on the second stream during the last sweep)
        if (!b_from_aux)
            hipLaunchKernelGGL(k_dense_pivot, dim3(1), dim3(256), lds_pivot, st, D.p, ld, K, diag0.p, Bk, dstatus.p, nullptr, 0u, 0, nullptr);
        DenseLookAhead la;
        if (lookahead && K + 1 < ns) {
            la.on = 1;
            la.flag = tiles_in_place;
            // in front of the update it waits for, so that it is resident when the three tiles arrive
            hipLaunchKernelGGL(k_dense_pivot, dim3(1), dim3(256), lds_pivot, c->aux_stream, D.p, ld, K + 1, diag0.p,
                               B.p + (size_t)((K + 1) & 1) * kSW * kSW, dstatus.p, tiles_in_place + K, need_tiles, spin_limit, b_complete + K + 1);
        }
        const unsigned int *ready = b_from_aux ? b_complete + K : nullptr;
        if (panel_split) hipLaunchKernelGGL(k_dense_panels<true>, dim3(2 * nt), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p, ready, spin_limit, dstatus.p);
        else hipLaunchKernelGGL(k_dense_panels<false>, dim3(nt), dim3(256), 0, st, D.p, ld, K, Bk, Cp.p, Wp.p, ready, spin_limit, dstatus.p);
        if (big_tiles) {
            if (la.on) hipLaunchKernelGGL(k_dense_update128<true>, dim3(stiles), dim3(256), lds_big_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, stiles, la);
            else hipLaunchKernelGGL(k_dense_update128<false>, dim3(stiles), dim3(256), lds_big_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, stiles, la);
        } else if (la.on) {
            hipLaunchKernelGGL(k_dense_update<true>, dim3(update_grid + 3), dim3(256), lds_upd_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, tmap, la);
        } else {
            hipLaunchKernelGGL(k_dense_update<false>, dim3(update_grid), dim3(256), lds_upd_bytes, st, D.p, ld, K, Bk, Cp.p, Wp.p, tmap, la);
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
    if (lookahead && c->aux_stream) FS_HIP(hipStreamSynchronize(c->aux_stream)); // (its last launch raised its flag before it ended)
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
