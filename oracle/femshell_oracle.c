/*
 * femshell_oracle.c -- CPU oracle for the fem-shell hot path (see femshell_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: never linked into, loaded by or called from the
 * product library.  Plain C99, no dependencies beyond libm.
 *
 * Every function cites the reference lines it restates.  "SA" is
 * /root/reference/src/fem-shell/fem-shell.cpp; "thesis" is /root/reference/doc/.
 * The code is written from the algorithm, with fixed-size stack arrays and no
 * DenseMatrix temporaries; nothing is transcribed from the reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "femshell_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <time.h>

/* ---------------------------------------------------------------- small dense helpers */

/* C(m x n) = A(m x k) * B(k x n), all row-major */
static void mm(int m, int k, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}

/* C(k x n) = A(m x k)^T * B(m x n) */
static void mtm(int m, int k, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < k; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < m; l++) s += A[l * k + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}

static double wall_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------- material (SA:273-294) */

void fso_material_matrices(const fso_material *mat, double Dm[9], double Dp[9])
{
    const double nu = mat->nu, E = mat->E, t = mat->thickness;
    const double D[9] = {1.0, nu, 0.0, nu, 1.0, 0.0, 0.0, 0.0, (1.0 - nu) / 2.0};
    const double cm = E / (1.0 - nu * nu);
    const double cp = E * pow(t, 3.0) / (12.0 * (1.0 - nu * nu));
    for (int i = 0; i < 9; i++) {
        Dm[i] = cm * D[i];
        Dp[i] = cp * D[i];
    }
}

/* ---------------------------------------------------------------- TRI3 frame (SA:306-341, 378-411) */

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

static int tri3_frame(const double xyz[9], double trafo[9], double transUV[6], double dphi[6],
                      double *area)
{
    double U[3], V[3], W[3];
    for (int i = 0; i < 3; i++) {
        U[i] = xyz[3 + i] - xyz[i]; /* B - A */
        V[i] = xyz[6 + i] - xyz[i]; /* C - A */
    }
    cross3(U, V, W);
    const double lw = norm3(W), lu = norm3(U);
    if (!(lw > 0.0) || !(lu > 0.0)) return -1;
    *area = 0.5 * lw;
    double ex[3], ey[3], ez[3];
    for (int i = 0; i < 3; i++) {
        ex[i] = U[i] / lu;
        ez[i] = W[i] / lw;
    }
    cross3(ez, ex, ey);
    for (int i = 0; i < 3; i++) {
        trafo[0 + i] = ex[i];
        trafo[3 + i] = ey[i];
        trafo[6 + i] = ez[i];
    }
    /* transUV = trafo * [U V]  (3x2) */
    for (int r = 0; r < 3; r++) {
        transUV[2 * r + 0] = trafo[3 * r] * U[0] + trafo[3 * r + 1] * U[1] + trafo[3 * r + 2] * U[2];
        transUV[2 * r + 1] = trafo[3 * r] * V[0] + trafo[3 * r + 1] * V[1] + trafo[3 * r + 2] * V[2];
    }
    /* coordinate differences, rows (12), (31), (23); node A is the local origin */
    dphi[0] = -transUV[0];              /* x12 */
    dphi[2] = transUV[1];               /* x31 */
    dphi[4] = transUV[0] - transUV[1];  /* x23 */
    dphi[1] = -transUV[2];              /* y12 */
    dphi[3] = transUV[3];               /* y31 */
    dphi[5] = transUV[2] - transUV[3];  /* y23 */
    return 0;
}

/* ---------------------------------------------------------------- TRI3 membrane, CST (SA:443-468) */

static void tri3_membrane(const double dphi[6], double area, const double Dm[9], double t,
                          double Ke_m[36])
{
    const double x12 = dphi[0], y12 = dphi[1], x31 = dphi[2], y31 = dphi[3], x23 = dphi[4],
                 y23 = dphi[5];
    const double s = 1.0 / (2.0 * area);
    double B[18] = {0};
    B[0 * 6 + 0] = y23 * s;  B[0 * 6 + 2] = y31 * s;  B[0 * 6 + 4] = y12 * s;
    B[1 * 6 + 1] = -x23 * s; B[1 * 6 + 3] = -x31 * s; B[1 * 6 + 5] = -x12 * s;
    B[2 * 6 + 0] = -x23 * s; B[2 * 6 + 1] = y23 * s;
    B[2 * 6 + 2] = -x31 * s; B[2 * 6 + 3] = y31 * s;
    B[2 * 6 + 4] = -x12 * s; B[2 * 6 + 5] = y12 * s;
    double DB[18];
    mm(3, 3, 6, Dm, B, DB);
    mtm(3, 6, 6, B, DB, Ke_m);
    for (int i = 0; i < 36; i++) Ke_m[i] *= t * area;
}

/* ---------------------------------------------------------------- Specht B~ (SA:698-891)
 * Built from the shape functions of thesis shellelements.tex:1023-1039 (chi) and
 * :1107-1111 (N_i): every chi is held as a polynomial in (L1,L2) with
 * L3 = 1-L1-L2, the shape functions are linear combinations of them and the
 * rows of B~ are their second derivatives (d2/dL1^2, d2/dL2^2, 2 d2/dL1dL2,
 * shellelements.tex:1148-1152; the factor 2 is SA:889-890). */

#define PDEG 4 /* chi7..chi9 are quartic: L_j L_i^2 plus (L1 L2 L3) x (linear) */
typedef struct { double c[PDEG + 1][PDEG + 1]; } poly; /* c[i][j] * L1^i * L2^j, i+j <= PDEG */

static poly p_zero(void)
{
    poly p;
    memset(&p, 0, sizeof p);
    return p;
}

static poly p_lin(double c0, double c1, double c2) /* c0 + c1 L1 + c2 L2 */
{
    poly p = p_zero();
    p.c[0][0] = c0;
    p.c[1][0] = c1;
    p.c[0][1] = c2;
    return p;
}

static poly p_axpby(double a, const poly *x, double b, const poly *y)
{
    poly r;
    for (int i = 0; i <= PDEG; i++)
        for (int j = 0; j <= PDEG; j++) r.c[i][j] = a * x->c[i][j] + b * y->c[i][j];
    return r;
}

static poly p_mul(const poly *x, const poly *y)
{
    poly r = p_zero();
    for (int i = 0; i <= PDEG; i++)
        for (int j = 0; i + j <= PDEG; j++)
            for (int k = 0; i + j + k <= PDEG; k++)
                for (int l = 0; i + j + k + l <= PDEG; l++)
                    r.c[i + k][j + l] += x->c[i][j] * y->c[k][l];
    return r;
}

/* (d2/dL1^2, d2/dL2^2, 2*d2/dL1dL2) of p at (L1,L2) */
static void p_curv(const poly *p, double L1, double L2, double out[3])
{
    double d11 = 0.0, d22 = 0.0, d12 = 0.0;
    for (int i = 0; i <= PDEG; i++)
        for (int j = 0; i + j <= PDEG; j++) {
            const double c = p->c[i][j];
            if (c == 0.0) continue;
            if (i >= 2) d11 += c * i * (i - 1) * pow(L1, i - 2) * pow(L2, j);
            if (j >= 2) d22 += c * j * (j - 1) * pow(L1, i) * pow(L2, j - 2);
            if (i >= 1 && j >= 1) d12 += c * i * j * pow(L1, i - 1) * pow(L2, j - 1);
        }
    out[0] = d11;
    out[1] = d22;
    out[2] = 2.0 * d12;
}

/* chi_1..chi_9 for the side ratios mu (SA:702-704) as polynomials */
static void specht_chi(const double mu[3], poly chi[9])
{
    poly L[3];
    L[0] = p_lin(0.0, 1.0, 0.0);
    L[1] = p_lin(0.0, 0.0, 1.0);
    L[2] = p_lin(1.0, -1.0, -1.0);
    for (int i = 0; i < 3; i++) chi[i] = L[i];
    chi[3] = p_mul(&L[0], &L[1]);
    chi[4] = p_mul(&L[1], &L[2]);
    chi[5] = p_mul(&L[2], &L[0]);
    poly L123 = p_mul(&chi[3], &L[2]);
    for (int i = 0; i < 3; i++) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double m = mu[k]; /* chi7 uses mu3, chi8 mu1, chi9 mu2 */
        poly sq = p_mul(&L[i], &L[i]);
        poly lead = p_mul(&L[j], &sq);           /* L_j L_i^2 */
        poly a = p_axpby(3.0 * (1.0 - m), &L[i], -(1.0 + 3.0 * m), &L[j]);
        poly lin = p_axpby(1.0, &a, 1.0 + 3.0 * m, &L[k]);
        poly bub = p_mul(&L123, &lin);
        chi[6 + i] = p_axpby(1.0, &lead, 0.5, &bub);
    }
}

/* rows of B~ from the curvature triples cc[n] of chi_1..chi_9 at one point: the shape functions are linear
 * combinations of the chi (shellelements.tex:1107-1111), and so are their curvatures */
static void specht_compose(const double cc[9][3], const double dphi[6], double B[27])
{
    /* coordinate differences seen from node i: (x_ki, y_ki) and (x_ji, y_ji);
     * dphi rows are (12),(31),(23) */
    static const int row_ki[3] = {1, 0, 2};
    static const int row_ji[3] = {0, 2, 1};
    for (int i = 0; i < 3; i++) {
        const int k = (i + 2) % 3;
        const double xki = dphi[2 * row_ki[i]], yki = dphi[2 * row_ki[i] + 1];
        const double xji = -dphi[2 * row_ji[i]], yji = -dphi[2 * row_ji[i] + 1];
        for (int r = 0; r < 3; r++) {
            const double d = cc[6 + k][r] - cc[3 + k][r];           /* chi_{k+6} - chi_{k+3} */
            const double e = cc[6 + i][r] - cc[6 + k][r];
            const double w = cc[i][r] - cc[3 + i][r] + cc[3 + k][r] + 2.0 * e;
            B[r * 9 + 3 * i + 0] = w;
            B[r * 9 + 3 * i + 1] = -yki * d + yji * cc[6 + i][r];
            B[r * 9 + 3 * i + 2] = xki * d - xji * cc[6 + i][r];
        }
    }
}

void fso_tri3_specht_B(const double C[3], double L1, double L2, const double dphi[6], double B[27])
{
    /* SA:702-704 */
    const double mu[3] = {(C[0] - C[1]) / C[2], (C[2] - C[0]) / C[1], (C[1] - C[2]) / C[0]};
    poly chi[9];
    specht_chi(mu, chi);
    double cc[9][3];
    for (int n = 0; n < 9; n++) p_curv(&chi[n], L1, L2, cc[n]);
    specht_compose(cc, dphi, B);
}

/* Tabulated form used by the assembly (the reference evaluates closed-form entries, SA:706-888; the device kernels
 * use the same idea, csrc/specht_tables.h): at a fixed Gauss point the curvatures of chi_1..chi_6 are constants and
 * those of chi_7..chi_9 are affine in one side ratio mu.  The tables are filled once from the polynomial machinery
 * above (values at mu = 0 and mu = 1), so the derivation stays the single source; fso_tri3_specht_B remains the
 * cross-check (tests/test_oracle_known_answers.py). */
static const double kGauss[3][2] = {{1.0 / 6.0, 1.0 / 6.0}, {2.0 / 3.0, 1.0 / 6.0}, {1.0 / 6.0, 2.0 / 3.0}};
static double g_chi0[3][9][3], g_chi1[3][3][3];
static int g_tables_ready = 0;

void fso_init_tables(void)
{
    if (g_tables_ready) return;
    const double mu0[3] = {0.0, 0.0, 0.0}, mu1[3] = {1.0, 1.0, 1.0};
    poly c0[9], c1[9];
    specht_chi(mu0, c0);
    specht_chi(mu1, c1);
    for (int g = 0; g < 3; g++) {
        for (int n = 0; n < 9; n++) p_curv(&c0[n], kGauss[g][0], kGauss[g][1], g_chi0[g][n]);
        for (int i = 0; i < 3; i++) {
            double t[3];
            p_curv(&c1[6 + i], kGauss[g][0], kGauss[g][1], t);
            for (int r = 0; r < 3; r++) g_chi1[g][i][r] = t[r] - g_chi0[g][6 + i][r];
        }
    }
    g_tables_ready = 1;
}

static int g_specht_polynomial = 0;
void fso_set_specht_polynomial(int on) { g_specht_polynomial = on; }

static void specht_B_gauss(const double C[3], int g, const double dphi[6], double B[27])
{
    if (g_specht_polynomial || !g_tables_ready) {
        fso_tri3_specht_B(C, kGauss[g][0], kGauss[g][1], dphi, B);
        return;
    }
    const double mu[3] = {(C[0] - C[1]) / C[2], (C[2] - C[0]) / C[1], (C[1] - C[2]) / C[0]};
    double cc[9][3];
    for (int n = 0; n < 6; n++)
        for (int r = 0; r < 3; r++) cc[n][r] = g_chi0[g][n][r];
    for (int i = 0; i < 3; i++) {
        const double m = mu[(i + 2) % 3];
        for (int r = 0; r < 3; r++) cc[6 + i][r] = g_chi0[g][6 + i][r] + m * g_chi1[g][i][r];
    }
    specht_compose(cc, dphi, B);
}

/* ---------------------------------------------------------------- TRI3 plate (SA:555-603) */

static void tri3_plate(const double dphi[6], double area, const double Dp[9], uint32_t flags,
                       double Ke_p[81])
{
    const double x31 = dphi[2], y31 = dphi[3], x23 = dphi[4], y23 = dphi[5];
    double C[3];
    for (int i = 0; i < 3; i++) C[i] = dphi[2 * i] * dphi[2 * i] + dphi[2 * i + 1] * dphi[2 * i + 1];

    /* SA:578-588; Y(2,1) as coded unless the flag is cleared */
    double Y[9];
    Y[0] = y23 * y23;  Y[1] = y31 * y31;  Y[2] = y23 * y31;
    Y[3] = x23 * x23;  Y[4] = x31 * x31;  Y[5] = x31 * x23;
    Y[6] = -2.0 * x23 * y23;
    Y[7] = (flags & FSO_REF_Y21) ? -2.0 * x31 * x31 : -2.0 * x31 * y31;
    Y[8] = -x23 * y31 - x31 * y23;
    const double sY = 1.0 / (4.0 * area * area);
    for (int i = 0; i < 9; i++) Y[i] *= sY;

    memset(Ke_p, 0, 81 * sizeof(double));
    for (int g = 0; g < 3; g++) {
        double B[27], YB[27], DYB[27], YtDYB[27], BtK[81];
        specht_B_gauss(C, g, dphi, B);
        mm(3, 3, 9, Y, B, YB);        /* Y B        */
        mm(3, 3, 9, Dp, YB, DYB);     /* Dp Y B     */
        mtm(3, 3, 9, Y, DYB, YtDYB);  /* Y^T Dp Y B */
        mtm(3, 9, 9, B, YtDYB, BtK);  /* B^T ...    */
        for (int i = 0; i < 81; i++) Ke_p[i] += BtK[i] * (1.0 / 6.0);
    }
    for (int i = 0; i < 81; i++) Ke_p[i] *= 2.0 * area;
}

/* ---------------------------------------------------------------- shell superposition (SA:999-1053) */

static void shell_superpose(int nodes, const double *Ke_m, const double *Ke_p, uint32_t flags,
                            double *K /* (6n)^2 node-major */)
{
    const int N = 6 * nodes, nm = 2 * nodes, np = 3 * nodes;
    memset(K, 0, (size_t)N * N * sizeof(double));
    for (int i = 0; i < nodes; i++)
        for (int j = 0; j < nodes; j++) {
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++)
                    K[(6 * i + a) * N + 6 * j + b] = Ke_m[(2 * i + a) * nm + 2 * j + b];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    K[(6 * i + 2 + a) * N + 6 * j + 2 + b] = Ke_p[(3 * i + a) * np + 3 * j + b];
            double d;
            if (flags & FSO_REF_DRILL_MAX) {
                /* as coded: max over the five "diagonal" entries of block (i,j), for every (i,j) */
                d = Ke_m[(2 * i) * nm + 2 * j];
                d = fmax(d, Ke_m[(2 * i + 1) * nm + 2 * j + 1]);
                d = fmax(d, Ke_p[(3 * i) * np + 3 * j]);
                d = fmax(d, Ke_p[(3 * i + 1) * np + 3 * j + 1]);
                d = fmax(d, Ke_p[(3 * i + 2) * np + 3 * j + 2]);
                d /= 1000.0;
            } else {
                /* thesis theory chapter (shellelements.tex:1722): smallest diagonal
                 * entry / 1000, diagonal node blocks only */
                if (i != j) continue;
                d = Ke_m[(2 * i) * nm + 2 * j];
                d = fmin(d, Ke_m[(2 * i + 1) * nm + 2 * j + 1]);
                d = fmin(d, Ke_p[(3 * i) * np + 3 * j]);
                d = fmin(d, Ke_p[(3 * i + 1) * np + 3 * j + 1]);
                d = fmin(d, Ke_p[(3 * i + 2) * np + 3 * j + 2]);
                d /= 1000.0;
            }
            K[(6 * i + 5) * N + 6 * j + 5] = d;
        }
}

/* ---------------------------------------------------------------- local -> global (SA:1061-1110) */

/* Kg(node-major) = blockwise TSub^T K_ij TSub with TSub = diag(trafo, trafo) */
static void rotate_blocks(int nodes, const double trafo[9], const double *K, double *Kg)
{
    const int N = 6 * nodes;
    double TS[36] = {0};
    for (int h = 0; h < 2; h++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) TS[(3 * h + i) * 6 + 3 * h + j] = trafo[3 * i + j];
    for (int i = 0; i < nodes; i++)
        for (int j = 0; j < nodes; j++) {
            double S[36], ST[36], R[36];
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) S[a * 6 + b] = K[(6 * i + a) * N + 6 * j + b];
            mm(6, 6, 6, S, TS, ST);
            mtm(6, 6, 6, TS, ST, R);
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) Kg[(6 * i + a) * N + 6 * j + b] = R[a * 6 + b];
        }
}

/* node-major (6*i+alpha) -> variable-major (nodes*alpha+i), SA:1105-1109 */
static void to_var_major(int nodes, const double *Knm, double *Kvm)
{
    const int N = 6 * nodes;
    for (int al = 0; al < 6; al++)
        for (int be = 0; be < 6; be++)
            for (int i = 0; i < nodes; i++)
                for (int j = 0; j < nodes; j++)
                    Kvm[(nodes * al + i) * N + nodes * be + j] = Knm[(6 * i + al) * N + 6 * j + be];
}

static int element_tri3_nm(const double xyz[9], const fso_material *mat, const double Dm[9],
                           const double Dp[9], double Kg[324], fso_tri3_parts *parts)
{
    double trafo[9], transUV[6], dphi[6], area, Ke_m[36], Ke_p[81], Kl[324];
    if (tri3_frame(xyz, trafo, transUV, dphi, &area)) return -1;
    tri3_membrane(dphi, area, Dm, mat->thickness, Ke_m);
    tri3_plate(dphi, area, Dp, mat->flags, Ke_p);
    shell_superpose(3, Ke_m, Ke_p, mat->flags, Kl);
    rotate_blocks(3, trafo, Kl, Kg);
    if (parts) {
        memcpy(parts->trafo, trafo, sizeof trafo);
        memcpy(parts->transUV, transUV, sizeof transUV);
        memcpy(parts->dphi, dphi, sizeof dphi);
        parts->area = area;
        memcpy(parts->Ke_m, Ke_m, sizeof Ke_m);
        memcpy(parts->Ke_p, Ke_p, sizeof Ke_p);
        memcpy(parts->K_local, Kl, sizeof Kl);
        memcpy(parts->K_global_nm, Kg, 324 * sizeof(double));
    }
    return 0;
}

int fso_element_tri3(const double xyz[9], const fso_material *mat, double Ke[324],
                     fso_tri3_parts *parts)
{
    double Dm[9], Dp[9], Kg[324];
    fso_material_matrices(mat, Dm, Dp);
    fso_init_tables(); /* the same arithmetic as the global assembly */
    if (element_tri3_nm(xyz, mat, Dm, Dp, Kg, parts)) return -1;
    to_var_major(3, Kg, Ke);
    return 0;
}

/* ---------------------------------------------------------------- QUAD4 frame (SA:342-375, 413-432) */

static int quad4_frame(const double xyz[12], double trafo[9], double loc[12] /*3x4*/,
                       double dphi[8] /*4x2*/, double *area)
{
    double mid[4][3]; /* midpoints of AB, BC, CD, DA */
    for (int s = 0; s < 4; s++)
        for (int i = 0; i < 3; i++) {
            const double p = xyz[3 * s + i], q = xyz[3 * ((s + 1) % 4) + i];
            mid[s][i] = p + 0.5 * (q - p);
        }
    double ex[3], ey[3], ez[3], vr[3];
    for (int i = 0; i < 3; i++) {
        ex[i] = mid[1][i] - mid[3][i]; /* nJ - nL */
        vr[i] = mid[2][i] - mid[0][i]; /* nK - nI */
    }
    double l = norm3(ex);
    if (!(l > 0.0)) return -1;
    for (int i = 0; i < 3; i++) ex[i] /= l;
    cross3(ex, vr, ez);
    l = norm3(ez);
    if (!(l > 0.0)) return -1;
    for (int i = 0; i < 3; i++) ez[i] /= l;
    cross3(ez, ex, ey);
    for (int i = 0; i < 3; i++) {
        trafo[0 + i] = ex[i];
        trafo[3 + i] = ey[i];
        trafo[6 + i] = ez[i];
    }
    /* loc = trafo * X (3x4), X columns = global node coordinates (no translation) */
    for (int r = 0; r < 3; r++)
        for (int n = 0; n < 4; n++)
            loc[4 * r + n] = trafo[3 * r] * xyz[3 * n] + trafo[3 * r + 1] * xyz[3 * n + 1] +
                             trafo[3 * r + 2] * xyz[3 * n + 2];
    /* rows (12),(23),(34),(41) */
    for (int s = 0; s < 4; s++) {
        dphi[2 * s + 0] = loc[0 * 4 + s] - loc[0 * 4 + (s + 1) % 4];
        dphi[2 * s + 1] = loc[1 * 4 + s] - loc[1 * 4 + (s + 1) % 4];
    }
    double a = 0.0; /* shoelace formula */
    for (int i = 0; i < 4; i++)
        a += loc[0 * 4 + i] * loc[1 * 4 + (i + 1) % 4] - loc[0 * 4 + (i + 1) % 4] * loc[1 * 4 + i];
    *area = 0.5 * a;
    return 0;
}

/* ---------------------------------------------------------------- QUAD4 membrane (SA:469-541) */

static void quad4_membrane(const double loc[12], const double Dm[9], double t, double Ke_m[64])
{
    static const double rn[4] = {-1.0, 1.0, 1.0, -1.0}, sn[4] = {-1.0, -1.0, 1.0, 1.0};
    const double root = sqrt(1.0 / 3.0);
    memset(Ke_m, 0, 64 * sizeof(double));
    for (int ii = 0; ii < 2; ii++)
        for (int jj = 0; jj < 2; jj++) {
            const double r = (ii ? -root : root), s = (jj ? -root : root);
            double dr[4], ds[4], J[4] = {0, 0, 0, 0};
            for (int n = 0; n < 4; n++) {
                dr[n] = 0.25 * rn[n] * (1.0 + sn[n] * s);
                ds[n] = 0.25 * sn[n] * (1.0 + rn[n] * r);
                J[0] += dr[n] * loc[0 * 4 + n];
                J[1] += dr[n] * loc[1 * 4 + n];
                J[2] += ds[n] * loc[0 * 4 + n];
                J[3] += ds[n] * loc[1 * 4 + n];
            }
            const double det = J[0] * J[3] - J[1] * J[2];
            /* A (3x4) maps (u_r,u_s,v_r,v_s) to strains, G (4x8) maps nodal (u,v) to those */
            double A[12] = {0}, G[32] = {0}, B[24], DB[24], BtDB[64];
            A[0 * 4 + 0] = J[3] / det;  A[0 * 4 + 1] = -J[1] / det;
            A[1 * 4 + 2] = -J[2] / det; A[1 * 4 + 3] = J[0] / det;
            A[2 * 4 + 0] = -J[2] / det; A[2 * 4 + 1] = J[0] / det;
            A[2 * 4 + 2] = J[3] / det;  A[2 * 4 + 3] = -J[1] / det;
            for (int n = 0; n < 4; n++) {
                G[0 * 8 + 2 * n] = dr[n];
                G[1 * 8 + 2 * n] = ds[n];
                G[2 * 8 + 2 * n + 1] = dr[n];
                G[3 * 8 + 2 * n + 1] = ds[n];
            }
            mm(3, 4, 8, A, G, B);
            mm(3, 3, 8, Dm, B, DB);
            mtm(3, 8, 8, B, DB, BtDB);
            for (int i = 0; i < 64; i++) Ke_m[i] += BtDB[i] * det * t;
        }
}

/* ---------------------------------------------------------------- QUAD4 plate, DKQ (SA:604-687, 901-990) */

static void quad4_dkq_B(const double H[5][4], double xi, double eta, const double Jinv[4],
                        double B[36])
{
    /* derivatives of the 8-node serendipity functions; corners (-1,-1),(1,-1),(1,1),(-1,1),
     * mid-sides 5..8 on sides 12,23,34,41  (SA:906-923) */
    static const double xn[4] = {-1.0, 1.0, 1.0, -1.0}, en[4] = {-1.0, -1.0, 1.0, 1.0};
    double Nx[8], Ne[8];
    for (int n = 0; n < 4; n++) {
        Nx[n] = 0.25 * xn[n] * (1.0 + eta * en[n]) * (2.0 * xi * xn[n] + eta * en[n]);
        Ne[n] = 0.25 * en[n] * (1.0 + xi * xn[n]) * (2.0 * eta * en[n] + xi * xn[n]);
    }
    Nx[4] = -xi * (1.0 - eta);          Ne[4] = -0.5 * (1.0 - xi * xi);
    Nx[5] = 0.5 * (1.0 - eta * eta);    Ne[5] = -eta * (1.0 + xi);
    Nx[6] = -xi * (1.0 + eta);          Ne[6] = 0.5 * (1.0 - xi * xi);
    Nx[7] = -0.5 * (1.0 - eta * eta);   Ne[7] = -eta * (1.0 - xi);

    double Hx_x[12], Hy_x[12], Hx_e[12], Hy_e[12];
    for (int n = 0; n < 4; n++) {
        const int sa = n, sb = (n + 3) % 4; /* the two sides meeting at node n (SA:931-981) */
        const double *N[2] = {Nx, Ne};
        double *Hx[2] = {Hx_x, Hx_e}, *Hy[2] = {Hy_x, Hy_e};
        for (int d = 0; d < 2; d++) {
            const double Na = N[d][4 + sa], Nb = N[d][4 + sb], Nn = N[d][n];
            Hx[d][3 * n + 0] = 1.5 * (H[0][sa] * Na - H[0][sb] * Nb);
            Hx[d][3 * n + 1] = H[1][sa] * Na + H[1][sb] * Nb;
            Hx[d][3 * n + 2] = Nn - H[2][sa] * Na - H[2][sb] * Nb;
            Hy[d][3 * n + 0] = 1.5 * (H[3][sa] * Na - H[3][sb] * Nb);
            Hy[d][3 * n + 1] = -Nn + H[4][sa] * Na + H[4][sb] * Nb;
            Hy[d][3 * n + 2] = -Hx[d][3 * n + 1];
        }
    }
    for (int i = 0; i < 12; i++) {
        B[0 * 12 + i] = Jinv[0] * Hx_x[i] + Jinv[1] * Hx_e[i];
        B[1 * 12 + i] = Jinv[2] * Hy_x[i] + Jinv[3] * Hy_e[i];
        B[2 * 12 + i] = Jinv[0] * Hy_x[i] + Jinv[1] * Hy_e[i] + Jinv[2] * Hx_x[i] + Jinv[3] * Hx_e[i];
    }
}

static void quad4_plate(const double dphi[8], const double Dp[9], double Ke_p[144])
{
    double H[5][4];
    for (int s = 0; s < 4; s++) {
        const double x = dphi[2 * s], y = dphi[2 * s + 1], l2 = x * x + y * y;
        H[0][s] = -x / l2;
        H[1][s] = 0.75 * x * y / l2;
        H[2][s] = (0.25 * x * x - 0.5 * y * y) / l2;
        H[3][s] = -y / l2;
        H[4][s] = (0.25 * y * y - 0.5 * x * x) / l2;
    }
    const double root = sqrt(1.0 / 3.0);
    memset(Ke_p, 0, 144 * sizeof(double));
    for (int ii = 0; ii < 2; ii++)
        for (int jj = 0; jj < 2; jj++) {
            const double r = (ii ? -root : root), s = (jj ? -root : root);
            double J[4]; /* SA:641-645 */
            J[0] = 0.25 * ((dphi[0] + dphi[4]) * s - dphi[0] + dphi[4]);
            J[1] = 0.25 * ((dphi[1] + dphi[5]) * s - dphi[1] + dphi[5]);
            J[2] = 0.25 * ((dphi[0] + dphi[4]) * r - dphi[2] + dphi[6]);
            J[3] = 0.25 * ((dphi[1] + dphi[5]) * r - dphi[3] + dphi[7]);
            const double det = J[0] * J[3] - J[1] * J[2];
            const double Jinv[4] = {J[3] / det, -J[1] / det, -J[2] / det, J[0] / det};
            double B[36], DB[36], BtDB[144];
            quad4_dkq_B(H, r, s, Jinv, B);
            mm(3, 3, 12, Dp, B, DB);
            mtm(3, 12, 12, B, DB, BtDB);
            for (int i = 0; i < 144; i++) Ke_p[i] += BtDB[i] * det;
        }
}

static int element_quad4_nm(const double xyz[12], const fso_material *mat, const double Dm[9],
                            const double Dp[9], double Kg[576], double *Ke_m_out, double *Ke_p_out)
{
    double trafo[9], loc[12], dphi[8], area, Ke_m[64], Ke_p[144], Kl[576];
    if (quad4_frame(xyz, trafo, loc, dphi, &area)) return -1;
    quad4_membrane(loc, Dm, mat->thickness, Ke_m);
    quad4_plate(dphi, Dp, Ke_p);
    shell_superpose(4, Ke_m, Ke_p, mat->flags, Kl);
    rotate_blocks(4, trafo, Kl, Kg);
    if (Ke_m_out) memcpy(Ke_m_out, Ke_m, sizeof Ke_m);
    if (Ke_p_out) memcpy(Ke_p_out, Ke_p, sizeof Ke_p);
    return 0;
}

int fso_element_quad4(const double xyz[12], const fso_material *mat, double Ke[576], double *Ke_m,
                      double *Ke_p, double *K_global_nm)
{
    double Dm[9], Dp[9], Kg[576];
    fso_material_matrices(mat, Dm, Dp);
    if (element_quad4_nm(xyz, mat, Dm, Dp, Kg, Ke_m, Ke_p)) return -1;
    if (K_global_nm) memcpy(K_global_nm, Kg, sizeof Kg);
    to_var_major(4, Kg, Ke);
    return 0;
}

/* ---------------------------------------------------------------- sparsity pattern */

static int cmp_i32(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

int64_t fso_bsr_pattern(int32_t n_nodes, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                        const int32_t *quad, int32_t *rowptr, int32_t *colidx)
{
    /* candidate neighbours per node (with duplicates), then sort + unique */
    int64_t *cnt = (int64_t *)calloc((size_t)n_nodes + 1, sizeof(int64_t));
    for (int32_t e = 0; e < n_tri; e++)
        for (int i = 0; i < 3; i++) cnt[tri[3 * e + i] + 1] += 3;
    for (int32_t e = 0; e < n_quad; e++)
        for (int i = 0; i < 4; i++) cnt[quad[4 * e + i] + 1] += 4;
    for (int32_t n = 0; n < n_nodes; n++) cnt[n + 1] += cnt[n];
    int32_t *cand = (int32_t *)malloc((size_t)(cnt[n_nodes] ? cnt[n_nodes] : 1) * sizeof(int32_t));
    int64_t *fill = (int64_t *)malloc((size_t)n_nodes * sizeof(int64_t));
    memcpy(fill, cnt, (size_t)n_nodes * sizeof(int64_t));
    for (int32_t e = 0; e < n_tri; e++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cand[fill[tri[3 * e + i]]++] = tri[3 * e + j];
    for (int32_t e = 0; e < n_quad; e++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) cand[fill[quad[4 * e + i]]++] = quad[4 * e + j];
    int64_t nnzb = 0;
    rowptr[0] = 0;
    for (int32_t n = 0; n < n_nodes; n++) {
        int32_t *c = cand + cnt[n];
        const int64_t m = cnt[n + 1] - cnt[n];
        qsort(c, (size_t)m, sizeof(int32_t), cmp_i32);
        int32_t prev = -1;
        for (int64_t q = 0; q < m; q++)
            if (c[q] != prev) {
                if (colidx) colidx[nnzb] = c[q];
                nnzb++;
                prev = c[q];
            }
        rowptr[n + 1] = (int32_t)nnzb;
    }
    free(cnt);
    free(cand);
    free(fill);
    return nnzb;
}

/* ---------------------------------------------------------------- assembly (SA:1160-1233) */

static int64_t find_block(const int32_t *rowptr, const int32_t *colidx, int32_t a, int32_t b)
{
    int32_t lo = rowptr[a], hi = rowptr[a + 1] - 1;
    while (lo <= hi) {
        const int32_t mid = (lo + hi) / 2;
        if (colidx[mid] == b) return mid;
        if (colidx[mid] < b) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* libMesh constrain_element_matrix_and_vector for homogeneous Dirichlet dofs
 * (SA:1227): row and column of every fixed dof zeroed, its diagonal set to 1 */
static void constrain_element(int nodes, const int32_t *conn, const uint8_t *dirichlet, double *Kg)
{
    const int N = 6 * nodes;
    if (!dirichlet) return;
    for (int i = 0; i < nodes; i++)
        for (int v = 0; v < 6; v++)
            if (dirichlet[conn[i]] & (1u << v)) {
                const int d = 6 * i + v;
                for (int q = 0; q < N; q++) {
                    Kg[d * N + q] = 0.0;
                    Kg[q * N + d] = 0.0;
                }
                Kg[d * N + d] = 1.0;
            }
}

static void scatter_element(int nodes, const int32_t *conn, const double *Kg, const int32_t *rowptr,
                            const int32_t *colidx, double *vals, int32_t n0, int32_t n1)
{
    const int N = 6 * nodes;
    for (int i = 0; i < nodes; i++) {
        if (conn[i] < n0 || conn[i] >= n1) continue; /* rows of another thread's range */
        for (int j = 0; j < nodes; j++) {
            double *blk = vals + 36 * find_block(rowptr, colidx, conn[i], conn[j]);
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) blk[6 * a + b] += Kg[(6 * i + a) * N + 6 * j + b];
        }
    }
}

/* rows [n0,n1) of K: every element touching such a node is computed, only its rows in the range are added
 * (elements on a range boundary are computed by both neighbours: no write is shared between threads) */
static int assemble_rows(int32_t n0, int32_t n1, const double *xyz, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                         const int32_t *quad, const fso_material *mat, const double *Dm, const double *Dp,
                         const uint8_t *dirichlet, const int32_t *rowptr, const int32_t *colidx, double *vals,
                         const int32_t *tri_list, int64_t n_tri_list, const int32_t *quad_list, int64_t n_quad_list)
{
    /* tri_list / quad_list: the elements touching a node of [n0,n1), ascending (the element partition of a thread,
     * built once per mesh like libMesh's local element ranges); NULL: scan all elements */
    const int64_t nt_loop = tri_list ? n_tri_list : n_tri;
    for (int64_t le = 0; le < nt_loop; le++) {
        const int32_t e = tri_list ? tri_list[le] : (int32_t)le;
        const int32_t *c = tri + 3 * (int64_t)e;
        int mine = 0;
        for (int i = 0; i < 3; i++) mine |= (c[i] >= n0 && c[i] < n1);
        if (!mine) continue;
        double X[9], Kg[324];
        for (int i = 0; i < 3; i++)
            for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
        if (element_tri3_nm(X, mat, Dm, Dp, Kg, NULL)) return -(e + 1);
        constrain_element(3, c, dirichlet, Kg);
        scatter_element(3, c, Kg, rowptr, colidx, vals, n0, n1);
    }
    const int64_t nq_loop = quad_list ? n_quad_list : n_quad;
    for (int64_t le = 0; le < nq_loop; le++) {
        const int32_t e = quad_list ? quad_list[le] : (int32_t)le;
        const int32_t *c = quad + 4 * (int64_t)e;
        int mine = 0;
        for (int i = 0; i < 4; i++) mine |= (c[i] >= n0 && c[i] < n1);
        if (!mine) continue;
        double X[12], Kg[576];
        for (int i = 0; i < 4; i++)
            for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
        if (element_quad4_nm(X, mat, Dm, Dp, Kg, NULL, NULL)) return -(n_tri + e + 1);
        constrain_element(4, c, dirichlet, Kg);
        scatter_element(4, c, Kg, rowptr, colidx, vals, n0, n1);
    }
    return 0;
}

static int g_threads = 1;
void fso_set_threads(int n)
{
    g_threads = n > 0 ? n : 1;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#else
    g_threads = 1;
#endif
}
int fso_threads(void) { return g_threads; }

/* Element lists of the threads' node ranges [n_nodes t/nt, n_nodes (t+1)/nt): an element belongs to every range that owns
 * one of its nodes.  Built once per (mesh, thread count) and kept: the threaded baseline's counterpart of the local element
 * ranges libMesh hands each MPI rank after partitioning (assemble_elasticity iterates active_local_elements, SA:1180). */
static struct {
    const int32_t *tri, *quad;
    int32_t n_tri, n_quad, n_nodes;
    int nt;
    uint64_t hash;
    int64_t *tptr, *qptr; /* nt + 1 */
    int32_t *tlist, *qlist;
} g_own = {0};

static int owner_of(int32_t node, int32_t n_nodes, int nt)
{
    int t = (int)(((int64_t)node * nt) / n_nodes);
    /* (n0 = n_nodes t / nt rounds down: step to the range that holds the node) */
    while (t + 1 < nt && node >= (int32_t)((int64_t)n_nodes * (t + 1) / nt)) t++;
    while (t > 0 && node < (int32_t)((int64_t)n_nodes * t / nt)) t--;
    return t;
}

static void build_lists(int nodes_per, int32_t n_el, const int32_t *conn, int32_t n_nodes, int nt, int64_t **ptr_out,
                        int32_t **list_out)
{
    int64_t *ptr = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
    int16_t *own = (int16_t *)malloc((size_t)(n_el ? n_el : 1) * 4 * sizeof(int16_t));
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < n_el; e++) {
        int16_t o[4] = {-1, -1, -1, -1};
        int k = 0;
        for (int i = 0; i < nodes_per; i++) {
            const int16_t t = (int16_t)owner_of(conn[(int64_t)nodes_per * e + i], n_nodes, nt);
            int seen = 0;
            for (int j = 0; j < k; j++) seen |= (o[j] == t);
            if (!seen) o[k++] = t;
        }
        for (int j = 0; j < 4; j++) own[4 * (int64_t)e + j] = o[j];
    }
    for (int32_t e = 0; e < n_el; e++)
        for (int j = 0; j < 4 && own[4 * (int64_t)e + j] >= 0; j++) ptr[own[4 * (int64_t)e + j] + 1]++;
    for (int t = 0; t < nt; t++) ptr[t + 1] += ptr[t];
    int32_t *list = (int32_t *)malloc((size_t)(ptr[nt] ? ptr[nt] : 1) * sizeof(int32_t));
    int64_t *fill = (int64_t *)malloc((size_t)nt * sizeof(int64_t));
    memcpy(fill, ptr, (size_t)nt * sizeof(int64_t));
    for (int32_t e = 0; e < n_el; e++)
        for (int j = 0; j < 4 && own[4 * (int64_t)e + j] >= 0; j++) list[fill[own[4 * (int64_t)e + j]]++] = e;
    free(fill);
    free(own);
    *ptr_out = ptr;
    *list_out = list;
}

static uint64_t conn_hash(const int32_t *v, int64_t n)
{
    uint64_t h = 0;
#pragma omp parallel for reduction(+ : h) schedule(static)
    for (int64_t i = 0; i < n; i++) h += ((uint64_t)(uint32_t)v[i] + 0x9E3779B97F4A7C15ull) * (2 * (uint64_t)i + 1);
    return h;
}

static void ownership_lists(int32_t n_nodes, int32_t n_tri, const int32_t *tri, int32_t n_quad, const int32_t *quad, int nt)
{
    /* (the hash guards against another mesh of the same size at the same address) */
    const uint64_t h = conn_hash(tri, 3 * (int64_t)n_tri) ^ (conn_hash(quad, 4 * (int64_t)n_quad) << 1);
    if (g_own.tptr && g_own.tri == tri && g_own.quad == quad && g_own.n_tri == n_tri && g_own.n_quad == n_quad &&
        g_own.n_nodes == n_nodes && g_own.nt == nt && g_own.hash == h)
        return;
    g_own.hash = h;
    free(g_own.tptr); free(g_own.qptr); free(g_own.tlist); free(g_own.qlist);
    build_lists(3, n_tri, tri, n_nodes, nt, &g_own.tptr, &g_own.tlist);
    build_lists(4, n_quad, quad, n_nodes, nt, &g_own.qptr, &g_own.qlist);
    g_own.tri = tri; g_own.quad = quad; g_own.n_tri = n_tri; g_own.n_quad = n_quad; g_own.n_nodes = n_nodes; g_own.nt = nt;
}

/* forget the lists (the caller is about to free or rewrite the connectivity arrays they were built from) */
void fso_drop_thread_lists(void)
{
    free(g_own.tptr); free(g_own.qptr); free(g_own.tlist); free(g_own.qlist);
    memset(&g_own, 0, sizeof g_own);
}

int fso_assemble_bsr(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                     int32_t n_quad, const int32_t *quad, const fso_material *mat,
                     const uint8_t *dirichlet, const double *loads, const int32_t *rowptr,
                     const int32_t *colidx, double *vals, double *F)
{
    double Dm[9], Dp[9];
    fso_material_matrices(mat, Dm, Dp);
    fso_init_tables();
    const int nt = g_threads;
    int rc = 0;
    if (nt <= 1) {
        memset(vals, 0, (size_t)rowptr[n_nodes] * 36 * sizeof(double));
        rc = assemble_rows(0, n_nodes, xyz, n_tri, tri, n_quad, quad, mat, Dm, Dp, dirichlet, rowptr, colidx, vals, NULL, 0, NULL, 0);
    } else {
        /* one contiguous node range per thread (the MPI ranks of the reference own contiguous dof ranges too) and the
         * elements that touch it; a thread zeroes (first call: first-touches) and fills the rows of its own range */
        ownership_lists(n_nodes, n_tri, tri, n_quad, quad, nt);
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nt; t++) {
            const int32_t n0 = (int32_t)((int64_t)n_nodes * t / nt), n1 = (int32_t)((int64_t)n_nodes * (t + 1) / nt);
            memset(vals + 36 * (int64_t)rowptr[n0], 0, (size_t)(rowptr[n1] - rowptr[n0]) * 36 * sizeof(double));
            const int r = assemble_rows(n0, n1, xyz, n_tri, tri, n_quad, quad, mat, Dm, Dp, dirichlet, rowptr, colidx, vals,
                                        g_own.tlist + g_own.tptr[t], g_own.tptr[t + 1] - g_own.tptr[t],
                                        g_own.qlist + g_own.qptr[t], g_own.qptr[t + 1] - g_own.qptr[t]);
            if (r) {
#pragma omp critical
                rc = r;
            }
        }
    }
    if (rc) return rc;
    /* SA:1118-1153: each node's load enters once; fixed dofs get rhs 0 (SA:1227) */
    if (F) {
#pragma omp parallel for schedule(static)
        for (int32_t n = 0; n < n_nodes; n++)
            for (int v = 0; v < 6; v++) {
                const int fixed = dirichlet && (dirichlet[n] & (1u << v));
                F[6 * (int64_t)n + v] = (fixed || !loads) ? 0.0 : loads[6 * (int64_t)n + v];
            }
    }
    return 0;
}

/* ---------------------------------------------------------------- SpMV and PCG */

void fso_bsr_spmv(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                  const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < n_nodes; a++) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int32_t q = rowptr[a]; q < rowptr[a + 1]; q++) {
            const double *blk = vals + 36 * (int64_t)q, *xb = x + 6 * (int64_t)colidx[q];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) acc[i] += blk[6 * i + j] * xb[j];
        }
        for (int i = 0; i < 6; i++) y[6 * (int64_t)a + i] = acc[i];
    }
}

/* in-place inverse of a 6x6 matrix by Gauss-Jordan with partial pivoting; 0 on success */
static int inv6(double *A)
{
    double M[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            M[i][j] = A[6 * i + j];
            M[i][6 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < 12; j++) {
                const double t = M[c][j];
                M[c][j] = M[p][j];
                M[p][j] = t;
            }
        const double d = 1.0 / M[c][c];
        for (int j = 0; j < 12; j++) M[c][j] *= d;
        for (int r = 0; r < 6; r++)
            if (r != c) {
                const double f = M[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j];
            }
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) A[6 * i + j] = M[i][6 + j];
    return 0;
}

static double dot(int64_t n, const double *a, const double *b)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

int fso_pcg_block_jacobi(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx,
                         const double *vals, const double *b, double rtol, int32_t max_it, double *x,
                         double *resid_hist, fso_pcg_info *info)
{
    const int64_t n = 6 * (int64_t)n_nodes;
    double *Minv = (double *)malloc((size_t)n_nodes * 36 * sizeof(double));
    double *r = (double *)malloc((size_t)n * sizeof(double));
    double *z = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *q = (double *)malloc((size_t)n * sizeof(double));
    int rc = 0;
    /* every vector is first touched by the thread that streams its rows later (same static schedule over the node rows
     * as fso_bsr_spmv and the vector loops below): on a multi-socket host the pages land on the right memory controller */
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < n_nodes; a++) {
        for (int i = 0; i < 6; i++) {
            r[6 * (int64_t)a + i] = 0.0;
            z[6 * (int64_t)a + i] = 0.0;
            p[6 * (int64_t)a + i] = 0.0;
            q[6 * (int64_t)a + i] = 0.0;
        }
        const int64_t d = find_block(rowptr, colidx, a, a);
        if (d < 0) {
#pragma omp critical
            if (!rc) rc = -2;
            continue;
        }
        memcpy(Minv + 36 * (int64_t)a, vals + 36 * d, 36 * sizeof(double));
        if (inv6(Minv + 36 * (int64_t)a)) {
#pragma omp critical
            if (!rc) rc = -3;
        }
    }
    fso_pcg_info res = {0, 0, 0.0, 0.0};
    if (!rc) {
        const double t0 = wall_seconds();
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            x[i] = 0.0;
            r[i] = b[i];
        }
        const double bnorm = sqrt(dot(n, b, b));
        if (bnorm == 0.0) {
            res.converged = 1;
        } else {
#pragma omp parallel for schedule(static)
            for (int32_t a = 0; a < n_nodes; a++)
                for (int i = 0; i < 6; i++) {
                    double s = 0.0;
                    for (int j = 0; j < 6; j++) s += Minv[36 * (int64_t)a + 6 * i + j] * r[6 * (int64_t)a + j];
                    z[6 * (int64_t)a + i] = s;
                    p[6 * (int64_t)a + i] = s;
                }
            double rz = dot(n, r, z);
            res.rel_residual = 1.0;
            for (int32_t it = 1; it <= max_it; it++) {
                fso_bsr_spmv(n_nodes, rowptr, colidx, vals, p, q);
                const double pq = dot(n, p, q);
                if (!(pq > 0.0)) { res.converged = -1; break; }
                const double alpha = rz / pq;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; i++) {
                    x[i] += alpha * p[i];
                    r[i] -= alpha * q[i];
                }
                const double rel = sqrt(dot(n, r, r)) / bnorm;
                res.iterations = it;
                res.rel_residual = rel;
                if (resid_hist) resid_hist[it - 1] = rel;
                if (rel <= rtol) { res.converged = 1; break; }
#pragma omp parallel for schedule(static)
                for (int32_t a = 0; a < n_nodes; a++)
                    for (int i = 0; i < 6; i++) {
                        double s = 0.0;
                        for (int j = 0; j < 6; j++)
                            s += Minv[36 * (int64_t)a + 6 * i + j] * r[6 * (int64_t)a + j];
                        z[6 * (int64_t)a + i] = s;
                    }
                const double rz_new = dot(n, r, z);
                const double beta = rz_new / rz;
                rz = rz_new;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
            }
        }
        res.seconds = wall_seconds() - t0;
    }
    if (info) *info = res;
    free(Minv); free(r); free(z); free(p); free(q);
    return rc;
}

double fso_time_assembly(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                         const fso_material *mat, const uint8_t *dirichlet, const double *loads,
                         const int32_t *rowptr, const int32_t *colidx, double *vals, double *F,
                         int32_t repeat)
{
    const double t0 = wall_seconds();
    for (int32_t k = 0; k < repeat; k++)
        if (fso_assemble_bsr(n_nodes, xyz, n_tri, tri, 0, NULL, mat, dirichlet, loads, rowptr, colidx,
                             vals, F))
            return -1.0;
    const double dt = wall_seconds() - t0;
    return dt > 0.0 ? (double)n_tri * repeat / dt : 0.0;
}

/* STREAM triad a = b + s c on the host threads (arrays first touched by the threads that stream them), best of `reps`
 * sweeps: the memory bandwidth the threaded baseline could reach, printed beside it by bench.py.  Returns GB/s counting
 * 24 bytes per element (two reads, one write; write-allocate traffic not counted, as in STREAM). */
double fso_stream_triad(int64_t n, int32_t reps)
{
    double *a = (double *)malloc((size_t)n * sizeof(double)), *b = (double *)malloc((size_t)n * sizeof(double)),
           *c = (double *)malloc((size_t)n * sizeof(double));
    if (!a || !b || !c) {
        free(a); free(b); free(c);
        return -1.0;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        a[i] = 0.0;
        b[i] = 1.0;
        c[i] = 2.0;
    }
    double best = 0.0;
    for (int32_t k = 0; k < reps; k++) {
        const double t0 = wall_seconds();
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) a[i] = b[i] + 3.0 * c[i];
        const double dt = wall_seconds() - t0;
        if (dt > 0.0 && 24.0 * (double)n / dt > best) best = 24.0 * (double)n / dt;
    }
    const double check = a[n / 2];
    free(a); free(b); free(c);
    return check == 7.0 ? best * 1e-9 : -1.0;
}
