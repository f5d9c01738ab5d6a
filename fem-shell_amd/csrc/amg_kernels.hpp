// amg_kernels.hpp -- launchers of the multigrid cycle's vector kernels (amg_kernels.hip).  The matrix products of
// the cycle (level operators, restriction, prolongation) all run through k_spmv of kernels.hip: every operator of
// the hierarchy is a sliced block ELL matrix of 6x6 blocks.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amg_patch.hpp"
#include "kernels.hpp"

namespace femshell {

// Chebyshev smoothing in D^-1 A (D = 6x6 diagonal blocks, m.minv):
//   start: d = inv_theta * D^-1 rin;  x = d (zero initial guess) or x += d (accumulate)
//   step:  r = rin - q (q = A d);  d = a d + c D^-1 r;  x += d        (rin may be r itself)
// every kernel is a no-op once gate->done != 0 (gate may be null)
// (vec32: what the smoothing products of the level keep in single precision, DeviceMatrix::vec32 -- 1: q and the transposed
//  products, 2: the direction d as well; d and q are then arrays of floats in the same buffers)
void launch_cheb_start(const DeviceMatrix &m, const double *rin, double *d, double *x, double inv_theta, bool accumulate,
                       const CgScalars *gate, hipStream_t st, int vec32 = 0);
// (gather: symmetric storage, q is the direct part of A d from launch_spmv_direct; the kernel collects the transposed products)
void launch_cheb_step(const DeviceMatrix &m, const double *rin, const double *q, double *rout, double *d, double *x,
                      double a, double c, const CgScalars *gate, hipStream_t st, bool gather = false, int vec32 = 0);

// ---- fused passes (round 5): the start of a Chebyshev smoothing in the epilogue of the kernel that produces its residual
// second phase of a symmetric-storage product + start of the post-smoothing: out = base_vec + sign (y + transposed products),
// d = inv_theta D^-1 out, x += d  (m: the level as the smoother sees it; q32 / d32: y and the transposed products / d are floats)
void launch_sym_gather_start(const DeviceMatrix &m, const double *y, double *out, const double *base_vec, double sign, double *d, double *x,
                             double inv_theta, bool q32, bool d32, const CgScalars *s, hipStream_t st);
// update of the flexible PCG + (gather) the second phase of q = K p in front of it, q stored whole + the start of the cycle's
// pre-smoothing on the new residual behind it: d = inv_theta D^-1 r, z = d  (m: level 0 as the smoother sees it)
void launch_pcg_update_start(const DeviceMatrix &m, const CgVectors &v, double *d, double *z, double inv_theta, bool gather, bool d32,
                             hipStream_t st);

// power iteration for lambda_max(D^-1 A): z = D^-1 q with the partial sums of z.z (one per workgroup of slice_grid(m))
void launch_minv_apply_norm(const DeviceMatrix &m, const double *q, double *z, double *partials, hipStream_t st);
// x[i] = deterministic pseudo-random value in (-1,1) for rows of real nodes, 0 for padding rows
void launch_fill_hash(double *x, int64_t n_real, int64_t n_total, hipStream_t st);

// coarsest level: y = Ainv b, Ainv dense n x n (row-major); rows [n, n_pad6) of y are set to zero
void launch_dense_gemv(const double *Ainv, const double *b, double *y, int32_t n, int32_t n_pad6, const CgScalars *gate,
                       hipStream_t st);

// ---- outer flexible PCG with a general preconditioner z = M(r) --------------------------------------------
// (G = slice_grid(m) partial sums per array; the scalar steps are the CG_PHASE_FLEX_* phases of k_cg_scalar)
// x = 0, r = b, partial sums of b.b into partials[0 ...]
void launch_pcg_init(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
// x += alpha p, r -= alpha q, partial sums of r.r into partials[0 ...]
void launch_pcg_update(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
// partial sums of r.r into partials[0 ...] (explicit residual of the residual replacement)
void launch_pcg_norm(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
// partial sums of r.z into partials[0 ...] and of z.q into partials[G ...]
void launch_pcg_dots(const DeviceMatrix &m, const CgVectors &v, hipStream_t st);
void launch_copy(const double *src, double *dst, int64_t n, const CgScalars *gate, hipStream_t st);
void launch_add(const double *src, double *dst, int64_t n, hipStream_t st); // dst += src
void launch_sub(const double *a, const double *b, double *out, int64_t n, hipStream_t st); // out = a - b

// ---- K cycle: two steps of flexible CG on the coarse problem of a level ------------------------------------
// sums[k] = a_k . b_k for up to three pairs (single workgroup finishes; vectors of n6 entries), then the
// coefficient step `phase` on the K-cycle scalars ks:
//   phase 1 (after c1 = M rc, v1 = A c1):        sums = (c1.v1, c1.rc)        -> ks->rho1, ks->a1, ks->t = a1/rho1
//   phase 2 (after c2 = M r2, v2 = A c2):        sums = (c2.v1, c2.v2, c2.r2) -> ks->w1, ks->w2 with x = w1 c1 + w2 c2
struct KcycScalars {
    double rho1, a1, t, w1, w2;
    double pad[3];
};
void launch_kcyc_dots(int phase, const double *a0, const double *b0, const double *a1, const double *b1, const double *a2,
                      const double *b2, int64_t n6, KcycScalars *ks, double *scratch, const CgScalars *gate, hipStream_t st);
// row-partitioned levels: the rank's three sums into sums[0..2] (no coefficient step), and -- once they are all-reduced --
// the coefficient step from them
void launch_kcyc_dots_local(int phase, const double *a0, const double *b0, const double *a1, const double *b1, const double *a2,
                            const double *b2, int64_t n6, double *scratch, double *sums, const CgScalars *gate, hipStream_t st);
void launch_kcyc_coefficients(int phase, const double *sums, KcycScalars *ks, const CgScalars *gate, hipStream_t st);
// the two coefficient steps of a K cycle on a small level (n6 <= kKcycSmall) as one single-workgroup launch each:
//   step 1: rho1 = c1.v1, a1 = c1.rc, t = a1 / rho1, r2 = rc - t v1
//   step 2: w1, w2 from (c2.v1, c2.v2, c2.r2), x = w1 c1 + w2 c2
constexpr int64_t kKcycSmall = 16384;
void launch_kcyc_step1_small(const double *c1, const double *v1, const double *rc, double *r2, int64_t n6, KcycScalars *ks,
                             const CgScalars *gate, hipStream_t st);
void launch_kcyc_step2_small(const double *c1, const double *c2, const double *v1, const double *v2, const double *r2, double *x,
                             int64_t n6, KcycScalars *ks, const CgScalars *gate, hipStream_t st);
// stage 1 of two dot products a0.b0 and a1.b1 over n entries: scratch[g] and scratch[128 + g] for g < the returned group
// count (<= 128) hold the partial sums, in a fixed order (error estimate of the refinement passes: the host adds them)
int launch_two_dots(const double *a0, const double *b0, const double *a1, const double *b1, int64_t n, double *scratch, hipStream_t st);
// out = in - ks->t * v
// (sums: row-partitioned levels -- the all-reduced sums of the step; the kernel forms the coefficients itself and keeps them in *ks)
void launch_kcyc_r2(const double *rc, const double *v1, double *r2, int64_t n6, KcycScalars *ks, const CgScalars *gate,
                    hipStream_t st, const double *sums = nullptr);
void launch_kcyc_combine(const double *c1, const double *c2, double *x, int64_t n6, KcycScalars *ks,
                         const CgScalars *gate, hipStream_t st, const double *sums = nullptr);

} // namespace femshell

namespace femshell {

// ---- multigrid setup on the device (level 0): numeric values of the smoothed prolongator, A P, R = P^T and the
// Galerkin operator P^T A P, on patterns the host computed from the block graph (amg_device_setup.cpp).  All operators are
// sliced block ELL; `count` gives the real entries of a row (the rest of its slots is padding with zero values).
struct EllView {
    int32_t n_rows = 0, n_slices = 0;
    const int32_t *slice_width = nullptr;
    const int64_t *slice_base = nullptr;
    const int32_t *cols = nullptr;
    const uint8_t *count = nullptr; // real entries per row
    double *vals = nullptr;
    int64_t total = 0;              // slots (= slice_base[n_slices])
};
// Near-null space of a level as the tentative prolongator reads it: n x 36 stored rows (B[36 a + 6 d + m]: dof d of node a
// in mode m -- the R factors of the previous coarsening step), or, on the finest level, generated from the mesh in HBM
// (amg_setup.cpp rigid_body_modes: rigid-body modes about `centre`, rotations projected onto the tangent plane of the
// node when normals != nullptr, constrained dofs zero).
struct NearNullSrc {
    const double *B = nullptr;
    const double *xyz = nullptr, *normals = nullptr;
    const uint8_t *dmask = nullptr;
    double cx = 0.0, cy = 0.0, cz = 0.0;
};
// Tentative prolongator: per aggregate the thin QR factorisation of its rows of the near-null space (modified Gram-Schmidt,
// two passes, dependent columns dropped -- the algorithm of tentative_prolongator in amg_setup.cpp).  The nodes of
// aggregate I are order[aptr[I] .. aptr[I+1]) (ascending); Q (n x 36) gets the orthonormal rows, Bc (na x 36) the R
// factors = the near-null space of the coarse level.  One wave per aggregate; `largest` = nodes of the largest aggregate
// (beyond 42 nodes the rows are kept in Q instead of registers; rows_in_memory: all of them, for tests).
void launch_amg_tentative_qr(const NearNullSrc &B, const int32_t *aptr, const int32_t *order, int32_t na, int32_t largest,
                             double *Q, double *Bc, bool rows_in_memory, hipStream_t st);
// P = P0 - omega D^-1 A P0: P0 has the block Q[i] in column agg[i]; pmap_own / pmap_in give, for every slot of A and every
// entry of its in-lists, the slot of P's row the product feeds
void launch_amg_prolongator(const DeviceMatrix &A, const int32_t *agg, const double *Q, double omega, const uint8_t *pmap_own,
                            const uint8_t *pmap_in, const EllView &P, hipStream_t st);
void launch_amg_ap(const DeviceMatrix &A, const EllView &P, const EllView &AP, hipStream_t st);
// R (rows = aggregates) from P: entry q of row I is the transposed block (rrow[q], slot rk[q]) of P
void launch_amg_restriction(const EllView &P, const int64_t *rptr, const int32_t *rrow, const uint8_t *rk, const EllView &R,
                            hipStream_t st);
// Ac = P^T (A P); diagonal slot first; coarse dofs without fine support get a unit diagonal.  mfma: one wave per coarse
// row contracts its tall-skinny panels with v_mfma_f64_16x16x4_f64; else one lane per result block on the vector ALUs
// (diag_key: row I's diagonal block carries the column key I + diag_key; vector-ALU kernel only when it is not zero)
void launch_amg_galerkin(const EllView &P, const EllView &AP, const int64_t *rptr, const int32_t *rrow, const uint8_t *rk,
                         const EllView &Ac, hipStream_t st, bool mfma, int diag_key = 0);

// ---- row-partitioned setup (amg_dist.cpp): rows of an ELL operator travel as W entries of 37 doubles per node, the column
// key (-1: none) in front of the row-major block.  contig: the operator keeps its blocks as 36 consecutive doubles (A P)
void launch_pack_ell_rows(const EllView &M, bool contig, const int32_t *nodes, int32_t count, int W, double *buf, hipStream_t st);
void launch_extract_keys(const double *buf, int64_t entries, int32_t *keys, hipStream_t st);
// the values of `count` received rows into the rows first_row ... of M (whose cols / count the host filled from the keys)
void launch_unpack_ell_rows(const double *buf, int32_t count, int W, const EllView &M, bool contig, int32_t first_row, hipStream_t st);

// ---- patch smoother (amg_patch.hpp)
// rigid edges of the operator in HBM (block-Jacobi inverse valid): counter[0] counts them all, the first cap are stored; counter[1]
// = edges above trigger_sigma (nearly coincident nodes), counter[2] = pairs of owned rows looked at
void launch_patch_sigma(const DeviceMatrix &A, double tau, double trigger_sigma, PatchEdge *edges, unsigned int *counter, unsigned int cap,
                        hipStream_t st);
// dense diagonal blocks of the clusters and the inverse diagonal blocks of their members
void launch_patch_gather(const DeviceMatrix &A, const PatchView &pv, double *Bc, double *dinv_of_member, hipStream_t st);
// d_c += coef M_c r_c, x_c += coef M_c r_c (either may be null; d_float: d is an array of floats)
void launch_patch_correct(const PatchView &pv, const double *r, double coef, double *d, bool d_float, double *x, const CgScalars *gate,
                          hipStream_t st);
// the rows of P of clustered nodes: -= omega M (A P0)  (width: widest row of P)
void launch_patch_prolongator(const DeviceMatrix &A, const int32_t *agg, const double *Q, double omega, const EllView &P, const PatchView &pv,
                              int width, hipStream_t st);
void launch_sqnorm_partials(const double *z, int64_t n, double *partials, int G, hipStream_t st);
// out2[j] = the sum of part[j G .. (j + 1) G) in index order, j = 0, 1 (the host's loop over the partial norms, on the device)
void launch_sums_in_order(const double *part, int G, double *out2, hipStream_t st);

} // namespace femshell
