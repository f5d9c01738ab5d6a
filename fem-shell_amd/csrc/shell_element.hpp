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
// The arithmetic is reorganised (closed-form CST blocks, Specht curvatures from the
// tabulated Gauss-point values of specht_tables.h, rotation as outer products of the frame
// axes); results agree with the reference's formulation to rounding.
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
};

// SA:318-340, 378-411.  Returns false for a degenerate triangle.
__device__ __forceinline__ bool tri3_frame(const double X[9], TriFrame &f)
{
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
    if (!(lw2 > 0.0) || !(lu2 > 0.0)) return false;
    const double lw = sqrt(lw2), lu = sqrt(lu2);
    f.area = 0.5 * lw;
    const double iu = 1.0 / lu, iw = 1.0 / lw;
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
    return true;
}

// Specht curvature block of node i at the three Gauss points: Bt[g][r][c], r = (d11, d22, 2 d12),
// c = (w, theta_x, theta_y).  Q[i][g][r] are the curvatures of chi7..chi9.
__device__ __forceinline__ void specht_node_block(int i, const TriFrame &f, const double Q[3][3][3],
                                                  double Bt[3][3][3])
{
    constexpr double C456[3][3] = SPECHT_C456_INIT;
    const int k = (i == 0) ? 2 : i - 1; // (i+2)%3
    // coordinate differences seen from node i: rows of (xs,ys) are (12),(31),(23);
    // (x_ki,y_ki) = row {1,0,2}[i], (x_ji,y_ji) = -row {0,2,1}[i]
    const double xki = sel3(i, f.xs[1], f.xs[0], f.xs[2]);
    const double yki = sel3(i, f.ys[1], f.ys[0], f.ys[2]);
    const double xji = -sel3(i, f.xs[0], f.xs[2], f.xs[1]);
    const double yji = -sel3(i, f.ys[0], f.ys[2], f.ys[1]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double ci = sel3(i, C456[0][r], C456[1][r], C456[2][r]);
        const double ck = sel3(k, C456[0][r], C456[1][r], C456[2][r]);
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const double qi = sel3(i, Q[0][g][r], Q[1][g][r], Q[2][g][r]);
            const double qk = sel3(k, Q[0][g][r], Q[1][g][r], Q[2][g][r]);
            const double P = qk - ck;                       // chi_{k+6} - chi_{k+3}
            Bt[g][r][0] = (ck - ci) + 2.0 * (qi - qk);      // N_w
            Bt[g][r][1] = yji * qi - yki * P;               // N_theta_x
            Bt[g][r][2] = xki * P - xji * qi;               // N_theta_y
        }
    }
}

// ---- per-element record ---------------------------------------------------------------
// Everything about a TRI3 element that does not depend on which node block is wanted.  The
// assembly kernel computes it once per element and slice and keeps it in LDS.
constexpr int kRecDoubles = 28; // 224 B: 16-byte aligned rows for ds_read_b128
// [0..8] ex,ey,ez  [9..11] xs  [12..14] ys  [15..17] mu  [18..23] Dt (00,01,02,11,12,22)
// [24] membrane scale t*cm/(4A)  [25] plate scale A/3  [26] 1.0 if valid  [27] unused

__device__ __forceinline__ bool tri3_record(const double X[9], const MatConst &mc, double rec[kRecDoubles])
{
    TriFrame f;
    const bool ok = tri3_frame(X, f);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < kRecDoubles; i++) rec[i] = 0.0;
        return false;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
        rec[d] = f.ex[d];
        rec[3 + d] = f.ey[d];
        rec[6 + d] = f.ez[d];
        rec[9 + d] = f.xs[d];
        rec[12 + d] = f.ys[d];
    }
    double C[3];
#pragma unroll
    for (int e = 0; e < 3; e++) C[e] = f.xs[e] * f.xs[e] + f.ys[e] * f.ys[e];
    rec[15] = (C[0] - C[1]) / C[2]; // SA:702-704
    rec[16] = (C[2] - C[0]) / C[1];
    rec[17] = (C[1] - C[2]) / C[0];
    // Y (SA:578-588) and Dt = Y^T Dp Y (symmetric)
    const double A = f.area;
    const double x31 = f.xs[1], y31 = f.ys[1], x23 = f.xs[2], y23 = f.ys[2];
    const double sY = 1.0 / (4.0 * A * A);
    double Y[3][3];
    Y[0][0] = y23 * y23 * sY; Y[0][1] = y31 * y31 * sY; Y[0][2] = y23 * y31 * sY;
    Y[1][0] = x23 * x23 * sY; Y[1][1] = x31 * x31 * sY; Y[1][2] = x31 * x23 * sY;
    Y[2][0] = -2.0 * x23 * y23 * sY;
    Y[2][1] = ((mc.flags & kRefY21) ? -2.0 * x31 * x31 : -2.0 * x31 * y31) * sY;
    Y[2][2] = (-x23 * y31 - x31 * y23) * sY;
    double DY[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        DY[0][c] = mc.cp * (Y[0][c] + mc.nu * Y[1][c]);
        DY[1][c] = mc.cp * (mc.nu * Y[0][c] + Y[1][c]);
        DY[2][c] = mc.cp * mc.g * Y[2][c];
    }
    int q = 18;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) rec[q++] = Y[0][r] * DY[0][c] + Y[1][r] * DY[1][c] + Y[2][r] * DY[2][c];
    rec[24] = mc.t * mc.cm / (4.0 * A);
    rec[25] = A / 3.0; // 2A * (Gauss weight 1/6)
    rec[26] = 1.0;
    rec[27] = 0.0;
    return true;
}

// Adds the global-axes 6x6 block K_e(ia, ib) of the element described by rec to acc (row-major).
__device__ __forceinline__ void tri3_block_add_rec(const double *rec, int ia, int ib, const MatConst &mc,
                                                   double acc[36])
{
    TriFrame f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        f.ex[d] = rec[d];
        f.ey[d] = rec[3 + d];
        f.ez[d] = rec[6 + d];
        f.xs[d] = rec[9 + d];
        f.ys[d] = rec[12 + d];
    }
    const double mu[3] = {rec[15], rec[16], rec[17]};
    double Dt[3][3];
    Dt[0][0] = rec[18]; Dt[0][1] = rec[19]; Dt[0][2] = rec[20];
    Dt[1][1] = rec[21]; Dt[1][2] = rec[22]; Dt[2][2] = rec[23];
    Dt[1][0] = Dt[0][1]; Dt[2][0] = Dt[0][2]; Dt[2][1] = Dt[1][2];
    const double sm = rec[24], sp = rec[25];

    // ---- membrane block (2x2), closed form of t*A*B_i^T Dm B_j  (SA:448-467)
    // node n has beta = y of row {2,1,0}[n], gamma = -x of that row
    const double bi = sel3(ia, f.ys[2], f.ys[1], f.ys[0]), gi = -sel3(ia, f.xs[2], f.xs[1], f.xs[0]);
    const double bj = sel3(ib, f.ys[2], f.ys[1], f.ys[0]), gj = -sel3(ib, f.xs[2], f.xs[1], f.xs[0]);
    const double m00 = sm * (bi * bj + mc.g * gi * gj);
    const double m01 = sm * (mc.nu * bi * gj + mc.g * gi * bj);
    const double m10 = sm * (mc.nu * gi * bj + mc.g * bi * gj);
    const double m11 = sm * (gi * gj + mc.g * bi * bj);

    // ---- plate block (3x3)  (SA:555-603), one Gauss point at a time to keep the live set small
    constexpr double QA[3][3][3] = SPECHT_QA_INIT;
    constexpr double QB[3][3][3] = SPECHT_QB_INIT;
    constexpr double C456[3][3] = SPECHT_C456_INIT;
    const int ka = (ia == 0) ? 2 : ia - 1, kb = (ib == 0) ? 2 : ib - 1; // (i+2)%3
    // coordinate differences seen from node i: (x_ki,y_ki) = row {1,0,2}[i], (x_ji,y_ji) = -row {0,2,1}[i]
    const double xki_a = sel3(ia, f.xs[1], f.xs[0], f.xs[2]), yki_a = sel3(ia, f.ys[1], f.ys[0], f.ys[2]);
    const double xji_a = -sel3(ia, f.xs[0], f.xs[2], f.xs[1]), yji_a = -sel3(ia, f.ys[0], f.ys[2], f.ys[1]);
    const double xki_b = sel3(ib, f.xs[1], f.xs[0], f.xs[2]), yki_b = sel3(ib, f.ys[1], f.ys[0], f.ys[2]);
    const double xji_b = -sel3(ib, f.xs[0], f.xs[2], f.xs[1]), yji_b = -sel3(ib, f.ys[0], f.ys[2], f.ys[1]);
    double p[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int g = 0; g < 3; g++) {
        double Bi[3][3], Bj[3][3]; // [r][c]: r = (d11, d22, 2 d12), c = (w, theta_x, theta_y)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            // curvatures of chi7..chi9 at this Gauss point: chi_{7+i} pairs with mu_{(i+2)%3}
            const double q0 = QA[0][g][r] + mu[2] * QB[0][g][r];
            const double q1 = QA[1][g][r] + mu[0] * QB[1][g][r];
            const double q2 = QA[2][g][r] + mu[1] * QB[2][g][r];
            {
                const double qi = sel3(ia, q0, q1, q2), qk = sel3(ka, q0, q1, q2);
                const double ci = sel3(ia, C456[0][r], C456[1][r], C456[2][r]);
                const double ck = sel3(ka, C456[0][r], C456[1][r], C456[2][r]);
                const double P = qk - ck;                  // chi_{k+6} - chi_{k+3}
                Bi[r][0] = (ck - ci) + 2.0 * (qi - qk);    // N_w
                Bi[r][1] = yji_a * qi - yki_a * P;         // N_theta_x
                Bi[r][2] = xki_a * P - xji_a * qi;         // N_theta_y
            }
            {
                const double qi = sel3(ib, q0, q1, q2), qk = sel3(kb, q0, q1, q2);
                const double ci = sel3(ib, C456[0][r], C456[1][r], C456[2][r]);
                const double ck = sel3(kb, C456[0][r], C456[1][r], C456[2][r]);
                const double P = qk - ck;
                Bj[r][0] = (ck - ci) + 2.0 * (qi - qk);
                Bj[r][1] = yji_b * qi - yki_b * P;
                Bj[r][2] = xki_b * P - xji_b * qi;
            }
        }
        double M[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) M[r][c] = Dt[r][0] * Bj[0][c] + Dt[r][1] * Bj[1][c] + Dt[r][2] * Bj[2][c];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) p[a][c] += Bi[0][a] * M[0][c] + Bi[1][a] * M[1][c] + Bi[2][a] * M[2][c];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) p[a][c] *= sp;

    // ---- drilling stiffness of this block  (SA:1035-1052)
    double d;
    if (mc.flags & kRefDrillMax) {
        d = fmax(fmax(fmax(m00, m11), fmax(p[0][0], p[1][1])), p[2][2]) / 1000.0;
    } else {
        d = (ia == ib) ? fmin(fmin(fmin(m00, m11), fmin(p[0][0], p[1][1])), p[2][2]) / 1000.0 : 0.0;
    }

    // ---- rotation: [T^T A11 T, T^T A12 T; T^T A21 T, T^T A22 T] as outer products of the
    //      frame axes (SA:1084-1102 with TSub = diag(T,T))
    const double *ex = f.ex, *ey = f.ey, *ez = f.ez;
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

// One-shot form (record kept in registers): used by the element-matrix export.
__device__ __forceinline__ bool tri3_block_add(const double X[9], int ia, int ib, const MatConst &mc,
                                               double acc[36])
{
    double rec[kRecDoubles];
    if (!tri3_record(X, mc, rec)) return false;
    tri3_block_add_rec(rec, ia, ib, mc, acc);
    return true;
}

} // namespace femshell
