// shell_element.hpp -- device-side flat-shell element math (gfx950, FP64).
//
// One call computes ONE 6x6 node block K_e(ia, ib) of one element in global axes:
// membrane + plate bending + drilling stiffness, rotated with TSub = diag(T, T).  The
// assembly kernel sums these blocks per block slot of K (row-owner gather), so no element
// matrix is ever written to memory.
//
// What is computed (reference: precice/fem-shell src/fem-shell/fem-shell.cpp, "SA"):
//   frame + coordinate differences   initElement               SA:306-341, 378-411
//   membrane (CST), node block 2x2   calcPlane                 SA:443-468
//   plate (Specht), node block 3x3   calcPlate + evalBTri      SA:555-603, 698-891
//   drilling stiffness               constructStiffnessMatrix  SA:1035-1052
//   rotation to global axes          localToGlobalTrafo        SA:1084-1102
// The arithmetic is reorganised (closed-form CST blocks; Specht plate blocks from element-level Gram tables
// over the tabulated Gauss-point curvatures of specht_tables.h, see tri3_record; rotation as outer products
// of the frame axes); results agree with the reference's formulation to rounding.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "specht_tables.h"

namespace femshell {

constexpr uint32_t kRefY21 = 0x1u;
constexpr uint32_t kRefDrillMax = 0x2u;

struct MatConst {
    double cm;  // E/(1-nu^2)
    double cp;  // E t^3 / (12 (1-nu^2))
    double nu;
    double g;   // (1-nu)/2
    double t;
    uint32_t flags;
    uint32_t pad;
};

__device__ __forceinline__ double sel3(int i, double a, double b, double c)
{
    return i == 0 ? a : (i == 1 ? b : c);
}

struct TriFrame {
    double ex[3], ey[3], ez[3]; // rows of trafo
    double xs[3], ys[3];        // coordinate differences, rows (12), (31), (23)
    double area;
    double inv2a;               // 1 / (2 area)
};

// SA:318-340, 378-411.  Returns false for a degenerate triangle.
__device__ __forceinline__ bool tri3_frame(const double X[9], TriFrame &f)
{
#pragma clang fp reassociate(on) contract(fast) // element math only; parity bar is 1e-12, not bitwise

    double U[3], V[3], W[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        U[d] = X[3 + d] - X[d];
        V[d] = X[6 + d] - X[d];
    }
    W[0] = U[1] * V[2] - U[2] * V[1];
    W[1] = U[2] * V[0] - U[0] * V[2];
    W[2] = U[0] * V[1] - U[1] * V[0];
    const double lw2 = W[0] * W[0] + W[1] * W[1] + W[2] * W[2];
    const double lu2 = U[0] * U[0] + U[1] * U[1] + U[2] * U[2];
    // no early return: a degenerate triangle gets zero scale factors, so everything derived from the frame is a
    // finite zero and the callers' records need no second code path (a branch here left part of the record array
    // in scratch memory, and the reload waited for every outstanding store of the wave)
    const bool ok = (lw2 > 0.0) && (lu2 > 0.0);
    // two reciprocal square roots instead of two square roots and two divisions
    const double iu = ok ? rsqrt(lu2) : 0.0, iw = ok ? rsqrt(lw2) : 0.0;
    f.area = 0.5 * lw2 * iw;
    f.inv2a = iw;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        f.ex[d] = U[d] * iu;
        f.ez[d] = W[d] * iw;
    }
    f.ey[0] = f.ez[1] * f.ex[2] - f.ez[2] * f.ex[1];
    f.ey[1] = f.ez[2] * f.ex[0] - f.ez[0] * f.ex[2];
    f.ey[2] = f.ez[0] * f.ex[1] - f.ez[1] * f.ex[0];
    // local coordinates of B and C (A is the origin)
    const double x2 = f.ex[0] * U[0] + f.ex[1] * U[1] + f.ex[2] * U[2];
    const double y2 = f.ey[0] * U[0] + f.ey[1] * U[1] + f.ey[2] * U[2];
    const double x3 = f.ex[0] * V[0] + f.ex[1] * V[1] + f.ex[2] * V[2];
    const double y3 = f.ey[0] * V[0] + f.ey[1] * V[1] + f.ey[2] * V[2];
    f.xs[0] = -x2;     f.ys[0] = -y2;      // (12)
    f.xs[1] = x3;      f.ys[1] = y3;       // (31)
    f.xs[2] = x2 - x3; f.ys[2] = y2 - y3;  // (23)
    return ok;
}

// ---- per-element record ---------------------------------------------------------------
// Everything about a TRI3 element that does not depend on which node block is wanted.  The
// assembly kernel computes it once per element and slice and keeps it in LDS.
//
// Plate part.  Specht's curvature matrix of node i at Gauss point g factors as
//     B_i(g) = Q_i(g) u_i^T + Q_k(g) v_i^T + C_k (e0 + w_i)^T - C_i e0^T,       k = (i+2)%3,
// with Q_n(g)[r] = QA[n][g][r] + mu_{(n+2)%3} QB[n][g][r] (curvatures of chi_{7+n}), the constant
// curvatures C_n of chi_{4+n}, and u_i = (2, y_ji, -x_ji), v_i = (-2, -y_ki, x_ki), w_i = (0, y_ki, -x_ki),
// e0 = (1,0,0).  Hence the 3x3 plate block of nodes (i,j),
//     p_ij = A/3 sum_g B_i(g)^T Dt B_j(g) = L_i^T S L_j,
// needs only the element-level Gram tables
//     QQ[n][m] = A/3 sum_g Q_n(g)^T Dt Q_m(g),  QC[n][m] = A/3 (sum_g Q_n(g))^T Dt C_m,  CC[n][m] = A C_n^T Dt C_m
// (27 doubles): S is a 4x4 pick from them and L_i has the rows u_i, v_i, e0 + w_i, -e0.
// 38 doubles = 304 B: 16-byte aligned, and 76 = 4*19 dwords, so the 16 lanes of a ds_read_b128 group that read
// the same field of consecutive records hit 16 different bank quads (40 doubles = 4*20 dwords was 50 % slower)
constexpr int kRecDoubles = 38;
constexpr int kRecKind = 16;    // 1.0 = TRI3, 2.0 = QUAD4, 0.0 = degenerate
// [0..8] ex,ey,ez  [9..11] xs  [12..14] ys  [15] membrane scale t*cm/(4A)  [16] kind
// [17..22] QQ (symmetric: 00,01,02,11,12,22)  [23..31] QC (row-major 3x3)  [32..37] CC (symmetric)
constexpr int kRecQQ = 17, kRecQC = 23, kRecCC = 32;
constexpr int kQuadX = 18, kQuadY = 22; // QUAD4 records: local x and y of the four nodes
// Records of meshes with quadrilaterals are 46 doubles (again 2*odd): a QUAD4 record also carries, per Gauss point g,
// the inverse Jacobian (four numbers) and its determinant at [kQuadGp + 5 g ..]
constexpr int kRecDoublesQuad = 66;
constexpr int kQuadSide = 46; // DKQ coefficients (a, b, c, d, e) of the four sides, side s = node s -> node s+1
constexpr int kQuadGp = 26;
// index of entry (n,m) of a symmetric 3x3 table stored as 00,01,02,11,12,22
__device__ __forceinline__ int sym3(int n, int m)
{
    const int lo = n < m ? n : m, hi = n < m ? m : n;
    return ((lo * (5 - lo)) >> 1) + hi;
}

__device__ __forceinline__ bool tri3_record(const double X[9], const MatConst &mc, double rec[kRecDoubles])
{
#pragma clang fp reassociate(on) contract(fast) // element math only; parity bar is 1e-12, not bitwise

    constexpr double QA[3][3][3] = SPECHT_QA_INIT;
    constexpr double QB[3][3][3] = SPECHT_QB_INIT;
    TriFrame f;
    const bool ok = tri3_frame(X, f); // degenerate: frame, side vectors and scales are zero, and so is the record
#pragma unroll
    for (int d = 0; d < 3; d++) {
        rec[d] = f.ex[d];
        rec[3 + d] = f.ey[d];
        rec[6 + d] = f.ez[d];
        rec[9 + d] = f.xs[d];
        rec[12 + d] = f.ys[d];
    }
    double C[3], mu[3];
#pragma unroll
    for (int e = 0; e < 3; e++) C[e] = f.xs[e] * f.xs[e] + f.ys[e] * f.ys[e];
    {
        // SA:702-704: (C0-C1)/C2, (C2-C0)/C1, (C1-C2)/C0 with one division
        const double c01 = C[0] * C[1], rall = ok ? 1.0 / (c01 * C[2]) : 0.0;
        mu[0] = (C[0] - C[1]) * (c01 * rall);
        mu[1] = (C[2] - C[0]) * (C[0] * C[2] * rall);
        mu[2] = (C[1] - C[2]) * (C[1] * C[2] * rall);
    }
    // Y (SA:578-588) and Dt = Y^T Dp Y (symmetric)
    const double A = f.area;
    const double x31 = f.xs[1], y31 = f.ys[1], x23 = f.xs[2], y23 = f.ys[2];
    const double sY = f.inv2a * f.inv2a; // 1 / (4 A^2)
    double Y[3][3];
    Y[0][0] = y23 * y23 * sY; Y[0][1] = y31 * y31 * sY; Y[0][2] = y23 * y31 * sY;
    Y[1][0] = x23 * x23 * sY; Y[1][1] = x31 * x31 * sY; Y[1][2] = x31 * x23 * sY;
    Y[2][0] = -2.0 * x23 * y23 * sY;
    Y[2][1] = ((mc.flags & kRefY21) ? -2.0 * x31 * x31 : -2.0 * x31 * y31) * sY;
    Y[2][2] = (-x23 * y31 - x31 * y23) * sY;
    const double sp = A * (1.0 / 3.0); // 2A * (Gauss weight 1/6), folded into Dt
    double DY[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        DY[0][c] = sp * mc.cp * (Y[0][c] + mc.nu * Y[1][c]);
        DY[1][c] = sp * mc.cp * (mc.nu * Y[0][c] + Y[1][c]);
        DY[2][c] = sp * mc.cp * mc.g * Y[2][c];
    }
    double Dt[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) Dt[r][c] = Dt[c][r] = Y[0][r] * DY[0][c] + Y[1][r] * DY[1][c] + Y[2][r] * DY[2][c];
    // Q_n(g), Dt Q_n(g) and the Gram tables
    double sQ[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double QQ[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int g = 0; g < 3; g++) {
        double Q[3][3], DQ[3][3]; // [n][r]: Q_n(g) and Dt Q_n(g)
#pragma unroll
        for (int n = 0; n < 3; n++) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                Q[n][r] = QA[n][g][r] + mu[(n + 2) % 3] * QB[n][g][r];
                sQ[n][r] += Q[n][r];
            }
#pragma unroll
            for (int r = 0; r < 3; r++) DQ[n][r] = Dt[r][0] * Q[n][0] + Dt[r][1] * Q[n][1] + Dt[r][2] * Q[n][2];
        }
#pragma unroll
        for (int n = 0; n < 3; n++)
#pragma unroll
            for (int m = n; m < 3; m++) QQ[n][m] += Q[n][0] * DQ[m][0] + Q[n][1] * DQ[m][1] + Q[n][2] * DQ[m][2];
    }
    // Dt C_m with C_0 = (0,0,2), C_1 = (0,-2,-2), C_2 = (-2,0,-2)  (curvatures of L1L2, L2L3, L3L1)
    double DC[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        DC[0][r] = 2.0 * Dt[r][2];
        DC[1][r] = -2.0 * (Dt[r][1] + Dt[r][2]);
        DC[2][r] = -2.0 * (Dt[r][0] + Dt[r][2]);
    }
    {
        int q = kRecQQ;
#pragma unroll
        for (int n = 0; n < 3; n++)
#pragma unroll
            for (int m = n; m < 3; m++) rec[q++] = QQ[n][m];
    }
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int m = 0; m < 3; m++)
            rec[kRecQC + 3 * n + m] = sQ[n][0] * DC[m][0] + sQ[n][1] * DC[m][1] + sQ[n][2] * DC[m][2];
    // CC[n][m] = 3 C_n . Dt C_m (symmetric): rows C_0 = (0,0,2), C_1 = (0,-2,-2), C_2 = (-2,0,-2)
    rec[kRecCC + 0] = 6.0 * DC[0][2];
    rec[kRecCC + 1] = 6.0 * DC[1][2];
    rec[kRecCC + 2] = 6.0 * DC[2][2];
    rec[kRecCC + 3] = -6.0 * (DC[1][1] + DC[1][2]);
    rec[kRecCC + 4] = -6.0 * (DC[2][1] + DC[2][2]);
    rec[kRecCC + 5] = -6.0 * (DC[2][0] + DC[2][2]);
    rec[15] = mc.t * mc.cm * (0.5 * f.inv2a); // t*cm/(4A)
    rec[kRecKind] = ok ? 1.0 : 0.0;
    return ok;
}

// Where the fields of a TRI3 record sit.  RecFull is the record tri3_record writes (k_assemble keeps it in LDS as it is);
// RecLean drops what the block functions can do without -- ey = ez x ex and the kind word -- and is 34 doubles = 4*17
// dwords (an odd multiple of a bank quad, like 38): the pipelined assembly kernel keeps two slices of records in LDS.
struct RecFull {
    static constexpr int ex = 0, ey = 3, ez = 6, xs = 9, ys = 12, sm = 15, QQ = kRecQQ, QC = kRecQC, CC = kRecCC;
    static constexpr bool has_ey = true;
};
struct RecLean {
    static constexpr int ex = 0, ey = 0, ez = 3, xs = 6, ys = 9, sm = 12, QQ = 13, QC = 19, CC = 28;
    static constexpr bool has_ey = false;
    static constexpr int doubles = 34;
};
// full record -> lean record, field by field (q = 0..33)
__host__ __device__ constexpr int lean_from_full(int q)
{
    return q < 3 ? q : (q < 6 ? q + 3 : (q < 12 ? q + 3 : (q == 12 ? 15 : q + 4)));
}
// the three frame axes of a record
template <class L> __device__ __forceinline__ void rec_axes(const double *rec, double ex[3], double ey[3], double ez[3])
{
#pragma clang fp reassociate(on) contract(fast)
#pragma unroll
    for (int d = 0; d < 3; d++) {
        ex[d] = rec[L::ex + d];
        ez[d] = rec[L::ez + d];
    }
    if (L::has_ey) {
#pragma unroll
        for (int d = 0; d < 3; d++) ey[d] = rec[L::ey + d];
    } else { // as in tri3_frame
        ey[0] = ez[1] * ex[2] - ez[2] * ex[1];
        ey[1] = ez[2] * ex[0] - ez[0] * ex[2];
        ey[2] = ez[0] * ex[1] - ez[1] * ex[0];
    }
}

// Adds the global-axes 6x6 block K_e(ia, ib) of the element described by rec to acc (row-major).
// rec may live in LDS (assembly kernel) or in registers/global (export kernel).  No selects: everything
// that depends on the (runtime) node indices is fetched by address.
template <class L = RecFull>
__device__ __forceinline__ void tri3_block_add_rec(const double *rec, int ia, int ib, const MatConst &mc, double acc[36])
{
#pragma clang fp reassociate(on) contract(fast) // element math only; parity bar is 1e-12, not bitwise

    const int ka = (ia == 0) ? 2 : ia - 1, kb = (ib == 0) ? 2 : ib - 1; // (i+2)%3
    // rows of (xs,ys) are (12),(31),(23).  Seen from node i: (x_ki,y_ki) = row {1,0,2}[i],
    // (x_ji,y_ji) = -row {0,2,1}[i]; membrane: beta = y of row {2,1,0}[i], gamma = -x of that row
    const int rki_a = (ia == 2) ? 2 : 1 - ia, rji_a = (ia == 0) ? 0 : 3 - ia;
    const int rki_b = (ib == 2) ? 2 : 1 - ib, rji_b = (ib == 0) ? 0 : 3 - ib;
    const double xki_a = rec[L::xs + rki_a], yki_a = rec[L::ys + rki_a], xji_a = -rec[L::xs + rji_a], yji_a = -rec[L::ys + rji_a];
    const double xki_b = rec[L::xs + rki_b], yki_b = rec[L::ys + rki_b], xji_b = -rec[L::xs + rji_b], yji_b = -rec[L::ys + rji_b];
    const double bi = rec[L::ys + 2 - ia], gi = -rec[L::xs + 2 - ia];
    const double bj = rec[L::ys + 2 - ib], gj = -rec[L::xs + 2 - ib];

    // ---- membrane block (2x2), closed form of t*A*B_i^T Dm B_j  (SA:448-467)
    const double sm = rec[L::sm];
    const double m00 = sm * (bi * bj + mc.g * gi * gj);
    const double m01 = sm * (mc.nu * bi * gj + mc.g * gi * bj);
    const double m10 = sm * (mc.nu * gi * bj + mc.g * bi * gj);
    const double m11 = sm * (gi * gj + mc.g * bi * bj);

    // ---- plate block (3x3)  (SA:555-603): p = L_i^T S L_j from the record's Gram tables
    const double *QQ = rec + L::QQ, *QC = rec + L::QC, *CC = rec + L::CC;
    const int s_ab = sym3(ia, ib), s_akb = sym3(ia, kb), s_kab = sym3(ka, ib), s_kakb = sym3(ka, kb);
    double S[4][4];
    S[0][0] = QQ[s_ab];        S[0][1] = QQ[s_akb];       S[0][2] = QC[3 * ia + kb]; S[0][3] = QC[3 * ia + ib];
    S[1][0] = QQ[s_kab];       S[1][1] = QQ[s_kakb];      S[1][2] = QC[3 * ka + kb]; S[1][3] = QC[3 * ka + ib];
    S[2][0] = QC[3 * ib + ka]; S[2][1] = QC[3 * kb + ka]; S[2][2] = CC[s_kakb];      S[2][3] = CC[s_kab];
    S[3][0] = QC[3 * ib + ia]; S[3][1] = QC[3 * kb + ia]; S[3][2] = CC[s_akb];       S[3][3] = CC[s_ab];
    double Xm[4][3];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const double d = S[m][2] - S[m][1];
        Xm[m][0] = 2.0 * (S[m][0] - S[m][1]) + (S[m][2] - S[m][3]);
        Xm[m][1] = yji_b * S[m][0] + yki_b * d;
        Xm[m][2] = -(xji_b * S[m][0] + xki_b * d);
    }
    double p[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double d = Xm[2][c] - Xm[1][c];
        p[0][c] = 2.0 * (Xm[0][c] - Xm[1][c]) + (Xm[2][c] - Xm[3][c]);
        p[1][c] = yji_a * Xm[0][c] + yki_a * d;
        p[2][c] = -(xji_a * Xm[0][c] + xki_a * d);
    }

    // ---- drilling stiffness of this block  (SA:1035-1052)
    double d;
    if (mc.flags & kRefDrillMax) {
        d = fmax(fmax(fmax(m00, m11), fmax(p[0][0], p[1][1])), p[2][2]) * 1.0e-3;
    } else {
        d = (ia == ib) ? fmin(fmin(fmin(m00, m11), fmin(p[0][0], p[1][1])), p[2][2]) * 1.0e-3 : 0.0;
    }

    // ---- rotation: [T^T A11 T, T^T A12 T; T^T A21 T, T^T A22 T] as outer products of the
    //      frame axes (SA:1084-1102 with TSub = diag(T,T))
    double ex[3], ey[3], ez[3];
    rec_axes<L>(rec, ex, ey, ez);
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const double a_x = m00 * ex[s] + m01 * ey[s]; // row ex of A11
        const double a_y = m10 * ex[s] + m11 * ey[s];
        const double a_z = p[0][0] * ez[s];
        const double b_z = p[0][1] * ex[s] + p[0][2] * ey[s]; // A12: row w
        const double c_x = p[1][0] * ez[s];                   // A21: column w
        const double c_y = p[2][0] * ez[s];
        const double d_x = p[1][1] * ex[s] + p[1][2] * ey[s]; // A22
        const double d_y = p[2][1] * ex[s] + p[2][2] * ey[s];
        const double d_z = d * ez[s];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            acc[6 * r + s] += ex[r] * a_x + ey[r] * a_y + ez[r] * a_z;
            acc[6 * r + 3 + s] += ez[r] * b_z;
            acc[6 * (3 + r) + s] += ex[r] * c_x + ey[r] * c_y;
            acc[6 * (3 + r) + 3 + s] += ex[r] * d_x + ey[r] * d_y + ez[r] * d_z;
        }
    }
}

// Index of entry (i,j), i <= j, of a symmetric 6x6 block stored as its upper triangle, row by row (21 doubles).
__host__ __device__ constexpr int sym6(int i, int j)
{
    return 6 * i - (i * (i - 1)) / 2 + (j - i);
}

// Diagonal blocks: K_e(ia, ia) is symmetric (S of tri3_block_add_rec is symmetric for ia == ib, m01 == m10), and so is
// every rotated contribution.  The block slot of a node with itself receives only such contributions -- half of all
// contributions of a mesh with symmetric storage -- so its lanes accumulate the upper triangle alone:
// 21 accumulators instead of 36, 26 record fields instead of 38, about 0.7 of the arithmetic.
// acc is the packed upper triangle (sym6).
template <class L = RecFull>
__device__ __forceinline__ void tri3_diag_add_rec(const double *rec, int ia, const MatConst &mc, double acc[21])
{
#pragma clang fp reassociate(on) contract(fast) // element math only; parity bar is 1e-12, not bitwise

    const int ka = (ia == 0) ? 2 : ia - 1;
    const int rki = (ia == 2) ? 2 : 1 - ia, rji = (ia == 0) ? 0 : 3 - ia;
    const double xki = rec[L::xs + rki], yki = rec[L::ys + rki], xji = -rec[L::xs + rji], yji = -rec[L::ys + rji];
    const double bi = rec[L::ys + 2 - ia], gi = -rec[L::xs + 2 - ia];

    // membrane (SA:448-467) with i == j
    const double sm = rec[L::sm];
    const double m00 = sm * (bi * bi + mc.g * gi * gi);
    const double m01 = sm * ((mc.nu + mc.g) * bi * gi);
    const double m11 = sm * (gi * gi + mc.g * bi * bi);

    // plate (SA:555-603): p = L^T S L, S symmetric
    const double *QQ = rec + L::QQ, *QC = rec + L::QC, *CC = rec + L::CC;
    const int s_aa = sym3(ia, ia), s_ak = sym3(ia, ka), s_kk = sym3(ka, ka);
    const double S00 = QQ[s_aa], S01 = QQ[s_ak], S02 = QC[3 * ia + ka], S03 = QC[3 * ia + ia];
    const double S11 = QQ[s_kk], S12 = QC[3 * ka + ka], S13 = QC[3 * ka + ia];
    const double S22 = CC[s_kk], S23 = CC[s_ak], S33 = CC[s_aa];
    const double S[4][4] = {{S00, S01, S02, S03}, {S01, S11, S12, S13}, {S02, S12, S22, S23}, {S03, S13, S23, S33}};
    double Xm[4][3];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const double d = S[m][2] - S[m][1];
        Xm[m][0] = 2.0 * (S[m][0] - S[m][1]) + (S[m][2] - S[m][3]);
        Xm[m][1] = yji * S[m][0] + yki * d;
        Xm[m][2] = -(xji * S[m][0] + xki * d);
    }
    double p0[3];
#pragma unroll
    for (int c = 0; c < 3; c++) p0[c] = 2.0 * (Xm[0][c] - Xm[1][c]) + (Xm[2][c] - Xm[3][c]);
    const double p11 = yji * Xm[0][1] + yki * (Xm[2][1] - Xm[1][1]);
    const double p12 = yji * Xm[0][2] + yki * (Xm[2][2] - Xm[1][2]);
    const double p22 = -(xji * Xm[0][2] + xki * (Xm[2][2] - Xm[1][2]));

    // drilling stiffness (SA:1035-1052)
    const double d = ((mc.flags & kRefDrillMax) ? fmax(fmax(fmax(m00, m11), fmax(p0[0], p11)), p22)
                                                 : fmin(fmin(fmin(m00, m11), fmin(p0[0], p11)), p22)) * 1.0e-3;

    // rotation (SA:1084-1102), upper triangle only
    double ex[3], ey[3], ez[3];
    rec_axes<L>(rec, ex, ey, ez);
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const double a_x = m00 * ex[s] + m01 * ey[s];
        const double a_y = m01 * ex[s] + m11 * ey[s];
        const double a_z = p0[0] * ez[s];
        const double b_z = p0[1] * ex[s] + p0[2] * ey[s];
        const double d_x = p11 * ex[s] + p12 * ey[s];
        const double d_y = p12 * ex[s] + p22 * ey[s];
        const double d_z = d * ez[s];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (r <= s) acc[sym6(r, s)] += ex[r] * a_x + ey[r] * a_y + ez[r] * a_z;
            acc[sym6(r, 3 + s)] += ez[r] * b_z;
            if (r <= s) acc[sym6(3 + r, 3 + s)] += ex[r] * d_x + ey[r] * d_y + ez[r] * d_z;
        }
    }
}

// =========================================================================================
// QUAD4: bilinear iso-parametric membrane (SA:469-541) + DKQ plate (SA:604-687, 901-990), 2x2 Gauss.
// Record: [0..8] ex,ey,ez  [kRecKind] = 2.0 if valid  [kQuadX..+3] local x of the 4 nodes  [kQuadY..+3] local y.
// =========================================================================================

// DKQ side coefficients (SA:613-621) of one element side with difference vector (x, y): they do not depend on
// the Gauss point, so a block computes them once per side it needs
struct DkqSide {
    double a, b, c, d, e;
};
__device__ __forceinline__ DkqSide dkq_side(double x, double y)
{
    const double l = 1.0 / (x * x + y * y);
    DkqSide s;
    s.a = -x * l;
    s.b = 0.75 * x * y * l;
    s.c = (0.25 * x * x - 0.5 * y * y) * l;
    s.d = -y * l;
    s.e = (0.25 * y * y - 0.5 * x * x) * l;
    return s;
}

// SA:342-375: frame from the mid-side points; local coordinates are T*X without translation.
__device__ __forceinline__ bool quad4_record(const double X[12], const MatConst &mc, double rec[kRecDoublesQuad])
{
    (void)mc;
#pragma unroll
    for (int i = 0; i < kRecDoublesQuad; i++) rec[i] = 0.0;
    double ex[3], ey[3], ez[3], vr[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double mI = X[d] + 0.5 * (X[3 + d] - X[d]);          // mid AB
        const double mJ = X[3 + d] + 0.5 * (X[6 + d] - X[3 + d]);  // mid BC
        const double mK = X[6 + d] + 0.5 * (X[9 + d] - X[6 + d]);  // mid CD
        const double mL = X[9 + d] + 0.5 * (X[d] - X[9 + d]);      // mid DA
        ex[d] = mJ - mL;
        vr[d] = mK - mI;
    }
    // no early returns (they left part of the record array in scratch memory, see tri3_frame): a degenerate quad gets
    // zero scale factors and kind 0, which block_add_rec treats as an all-zero triangle record
    const double lx2 = ex[0] * ex[0] + ex[1] * ex[1] + ex[2] * ex[2];
    bool ok = lx2 > 0.0;
    const double ilx = ok ? 1.0 / sqrt(lx2) : 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) ex[d] *= ilx;
    ez[0] = ex[1] * vr[2] - ex[2] * vr[1];
    ez[1] = ex[2] * vr[0] - ex[0] * vr[2];
    ez[2] = ex[0] * vr[1] - ex[1] * vr[0];
    const double lz2 = ez[0] * ez[0] + ez[1] * ez[1] + ez[2] * ez[2];
    ok = ok && lz2 > 0.0;
    const double ilz = ok ? 1.0 / sqrt(lz2) : 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) ez[d] *= ilz;
    ey[0] = ez[1] * ex[2] - ez[2] * ex[1];
    ey[1] = ez[2] * ex[0] - ez[0] * ex[2];
    ey[2] = ez[0] * ex[1] - ez[1] * ex[0];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        rec[d] = ex[d];
        rec[3 + d] = ey[d];
        rec[6 + d] = ez[d];
    }
#pragma unroll
    for (int n = 0; n < 4; n++) {
        rec[kQuadX + n] = ex[0] * X[3 * n] + ex[1] * X[3 * n + 1] + ex[2] * X[3 * n + 2];
        rec[kQuadY + n] = ey[0] * X[3 * n] + ey[1] * X[3 * n + 1] + ey[2] * X[3 * n + 2];
    }
    // the Jacobian determinant must not vanish at the Gauss points; a zero-area quad is rejected here
    double a2 = 0.0;
#pragma unroll
    for (int n = 0; n < 4; n++) a2 += rec[kQuadX + n] * rec[kQuadY + (n + 1) % 4] - rec[kQuadX + (n + 1) % 4] * rec[kQuadY + n];
    ok = ok && fabs(a2) > 0.0;
    // Jacobian of the bilinear map at the four Gauss points (SA:482-487 order; SA:489-538 and SA:641-684 use the
    // same four numbers), its inverse and determinant
    {
        const double *x = rec + kQuadX, *y = rec + kQuadY;
        const double x12 = x[0] - x[1], y12 = y[0] - y[1], x23 = x[1] - x[2], y23 = y[1] - y[2];
        const double x34 = x[2] - x[3], y34 = y[2] - y[3], x41 = x[3] - x[0], y41 = y[3] - y[0];
        const double root = 0.57735026918962576451; // sqrt(1/3)
#pragma unroll
        for (int gp = 0; gp < 4; gp++) {
            const double r = (gp & 2) ? -root : root, s = (gp & 1) ? -root : root;
            const double J00 = 0.25 * ((x12 + x34) * s - x12 + x34), J01 = 0.25 * ((y12 + y34) * s - y12 + y34);
            const double J10 = 0.25 * ((x12 + x34) * r - x23 + x41), J11 = 0.25 * ((y12 + y34) * r - y23 + y41);
            const double det = J00 * J11 - J01 * J10, idet = (ok && det != 0.0) ? 1.0 / det : 0.0;
            double *g = rec + kQuadGp + 5 * gp;
            g[0] = J11 * idet;
            g[1] = -J01 * idet;
            g[2] = -J10 * idet;
            g[3] = J00 * idet;
            g[4] = det;
        }
    }
    // side coefficients of the plate part (they do not depend on the Gauss point nor on the block): side s from node s
    // to node s+1, differences as in SA:413-424
#pragma unroll
    for (int sd = 0; sd < 4; sd++) {
        const double dx = rec[kQuadX + sd] - rec[kQuadX + (sd + 1) % 4], dy = rec[kQuadY + sd] - rec[kQuadY + (sd + 1) % 4];
        const double l2 = dx * dx + dy * dy;
        const double l = (ok && l2 > 0.0) ? 1.0 / l2 : 0.0;
        double *q = rec + kQuadSide + 5 * sd;
        q[0] = -dx * l;
        q[1] = 0.75 * dx * dy * l;
        q[2] = (0.25 * dx * dx - 0.5 * dy * dy) * l;
        q[3] = -dy * l;
        q[4] = (0.25 * dy * dy - 0.5 * dx * dx) * l;
    }
    rec[kRecKind] = ok ? 2.0 : 0.0;
    if (!ok) {
#pragma unroll
        for (int i = 0; i < kRecDoublesQuad; i++) rec[i] = 0.0; // (selects, no branch: the values above may be Inf / NaN)
    }
    return ok;
}

// DKQ curvature columns of node n at one Gauss point: B[r][c], r = (kxx, kyy, kxy), c = (w, tx, ty).
// sa / sb: the two sides meeting at the node (sa = n: towards the next node, sb = n-1: from the previous node);
// serendipity derivatives of the corner and the two mid-side functions.
__device__ __forceinline__ void dkq_node_block(const DkqSide &sa, const DkqSide &sb, double Nxn, double Nen,
                                               double Nxa, double Nea, double Nxb, double Neb, const double Ji[4],
                                               double B[3][3])
{
#pragma clang fp reassociate(on) contract(fast) // element math only; parity bar is 1e-12, not bitwise
    // SA:931-981, xi then eta derivatives
    const double Hxx[3] = {1.5 * (sa.a * Nxa - sb.a * Nxb), sa.b * Nxa + sb.b * Nxb, Nxn - sa.c * Nxa - sb.c * Nxb};
    const double Hxe[3] = {1.5 * (sa.a * Nea - sb.a * Neb), sa.b * Nea + sb.b * Neb, Nen - sa.c * Nea - sb.c * Neb};
    const double Hyx[3] = {1.5 * (sa.d * Nxa - sb.d * Nxb), -Nxn + sa.e * Nxa + sb.e * Nxb, -Hxx[1]};
    const double Hye[3] = {1.5 * (sa.d * Nea - sb.d * Neb), -Nen + sa.e * Nea + sb.e * Neb, -Hxe[1]};
#pragma unroll
    for (int c = 0; c < 3; c++) { // SA:984-989
        B[0][c] = Ji[0] * Hxx[c] + Ji[1] * Hxe[c];
        B[1][c] = Ji[2] * Hyx[c] + Ji[3] * Hye[c];
        B[2][c] = Ji[0] * Hyx[c] + Ji[1] * Hye[c] + Ji[2] * Hxx[c] + Ji[3] * Hxe[c];
    }
}

// Adds the global-axes 6x6 block K_e(ia, ib) of the QUAD4 element described by rec to acc.
__device__ __forceinline__ void quad4_block_add_rec(const double *rec, int ia, int ib, const MatConst &mc,
                                                    double acc[36])
{
    // node data by dynamic index straight from the record (LDS or registers), no selects
    const int ia_p = (ia + 3) & 3, ib_p = (ib + 3) & 3; // side s runs from node s to node s+1: a node sees sides s and s-1
    // natural coordinates of the corner nodes: (-1,-1), (1,-1), (1,1), (-1,1)
    const double ri = (ia == 1 || ia == 2) ? 1.0 : -1.0, si = (ia >= 2) ? 1.0 : -1.0;
    const double rj = (ib == 1 || ib == 2) ? 1.0 : -1.0, sj = (ib >= 2) ? 1.0 : -1.0;
    auto side = [&](int sd) {
        const double *q = rec + kQuadSide + 5 * sd;
        DkqSide r;
        r.a = q[0]; r.b = q[1]; r.c = q[2]; r.d = q[3]; r.e = q[4];
        return r;
    };
    const DkqSide sa_i = side(ia), sb_i = side(ia_p), sa_j = side(ib), sb_j = side(ib_p);

    double m00 = 0.0, m01 = 0.0, m10 = 0.0, m11 = 0.0;
    double p[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const double root = 0.57735026918962576451; // sqrt(1/3)
#pragma unroll 1 // (unrolled by two the kernel spills 356 bytes per lane and takes 1.87 ms instead of 1.14 per million quads)
    for (int gp = 0; gp < 4; gp++) {
        const double r = (gp & 2) ? -root : root, s = (gp & 1) ? -root : root; // SA:482-487 order
        // inverse Jacobian and determinant of this Gauss point from the record
        const double *gq = rec + kQuadGp + 5 * gp;
        const double Ji[4] = {gq[0], gq[1], gq[2], gq[3]};
        const double det = gq[4];
        // ---- membrane
        {
            const double dri = 0.25 * ri * (1.0 + si * s), dsi = 0.25 * si * (1.0 + ri * r);
            const double drj = 0.25 * rj * (1.0 + sj * s), dsj = 0.25 * sj * (1.0 + rj * r);
            const double bi = Ji[0] * dri + Ji[1] * dsi, gi = Ji[2] * dri + Ji[3] * dsi; // dN/dx, dN/dy
            const double bj = Ji[0] * drj + Ji[1] * dsj, gj = Ji[2] * drj + Ji[3] * dsj;
            const double w = det * mc.t * mc.cm;
            m00 += w * (bi * bj + mc.g * gi * gj);
            m01 += w * (mc.nu * bi * gj + mc.g * gi * bj);
            m10 += w * (mc.nu * gi * bj + mc.g * bi * gj);
            m11 += w * (gi * gj + mc.g * bi * bj);
        }
        // ---- plate
        {
            // serendipity derivatives (SA:906-923): corner n, mid-side of side s (nodes 5..8)
            auto corner_x = [&](double rr, double ss) { return 0.25 * rr * (1.0 + s * ss) * (2.0 * r * rr + s * ss); };
            auto corner_e = [&](double rr, double ss) { return 0.25 * ss * (1.0 + r * rr) * (2.0 * s * ss + r * rr); };
            auto mid_x = [&](int sd) {
                return sd == 0 ? -r * (1.0 - s) : (sd == 1 ? 0.5 * (1.0 - s * s) : (sd == 2 ? -r * (1.0 + s) : -0.5 * (1.0 - s * s)));
            };
            auto mid_e = [&](int sd) {
                return sd == 0 ? -0.5 * (1.0 - r * r) : (sd == 1 ? -s * (1.0 + r) : (sd == 2 ? 0.5 * (1.0 - r * r) : -s * (1.0 - r)));
            };
            double Bi[3][3], Bj[3][3];
            dkq_node_block(sa_i, sb_i, corner_x(ri, si), corner_e(ri, si), mid_x(ia), mid_e(ia), mid_x(ia_p), mid_e(ia_p), Ji, Bi);
            if (ia != ib) { // (the work items of the diagonal slots sit together in the first wave: it skips this as a whole)
                dkq_node_block(sa_j, sb_j, corner_x(rj, sj), corner_e(rj, sj), mid_x(ib), mid_e(ib), mid_x(ib_p), mid_e(ib_p), Ji, Bj);
            } else {
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int c = 0; c < 3; c++) Bj[a][c] = Bi[a][c];
            }
            const double w = det * mc.cp;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double M0 = w * (Bj[0][c] + mc.nu * Bj[1][c]);
                const double M1 = w * (mc.nu * Bj[0][c] + Bj[1][c]);
                const double M2 = w * mc.g * Bj[2][c];
#pragma unroll
                for (int a = 0; a < 3; a++) p[a][c] += Bi[0][a] * M0 + Bi[1][a] * M1 + Bi[2][a] * M2;
            }
        }
    }

    double d;
    if (mc.flags & kRefDrillMax) {
        d = fmax(fmax(fmax(m00, m11), fmax(p[0][0], p[1][1])), p[2][2]) * 1.0e-3;
    } else {
        d = (ia == ib) ? fmin(fmin(fmin(m00, m11), fmin(p[0][0], p[1][1])), p[2][2]) * 1.0e-3 : 0.0;
    }
    const double ex[3] = {rec[0], rec[1], rec[2]}, ey[3] = {rec[3], rec[4], rec[5]}, ez[3] = {rec[6], rec[7], rec[8]};
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const double a_x = m00 * ex[s] + m01 * ey[s];
        const double a_y = m10 * ex[s] + m11 * ey[s];
        const double a_z = p[0][0] * ez[s];
        const double b_z = p[0][1] * ex[s] + p[0][2] * ey[s];
        const double c_x = p[1][0] * ez[s];
        const double c_y = p[2][0] * ez[s];
        const double d_x = p[1][1] * ex[s] + p[1][2] * ey[s];
        const double d_y = p[2][1] * ex[s] + p[2][2] * ey[s];
        const double d_z = d * ez[s];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            acc[6 * r + s] += ex[r] * a_x + ey[r] * a_y + ez[r] * a_z;
            acc[6 * r + 3 + s] += ez[r] * b_z;
            acc[6 * (3 + r) + s] += ex[r] * c_x + ey[r] * c_y;
            acc[6 * (3 + r) + 3 + s] += ex[r] * d_x + ey[r] * d_y + ez[r] * d_z;
        }
    }
}

// dispatch on the record's element kind
template <bool kHasQuads>
__device__ __forceinline__ void block_add_rec(const double *rec, int ia, int ib, const MatConst &mc, double acc[36])
{
    if (kHasQuads && rec[kRecKind] == 2.0) quad4_block_add_rec(rec, ia, ib, mc, acc);
    else tri3_block_add_rec(rec, ia, ib, mc, acc);
}

} // namespace femshell
