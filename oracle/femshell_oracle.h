/*
 * femshell_oracle.h -- CPU oracle for the fem-shell hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (precice/fem-shell, src/fem-shell/fem-shell.cpp, "SA" below) used as
 * the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library
 * (fem-shell_amd/csrc) never links, loads or calls anything in this directory.
 *
 * Parity pinning: the reference needs libMesh + PETSc (absent, unbuildable
 * here), so the oracle is pinned by the reference's own known answers: the
 * thesis validation tables A, B, C, D, F, G (doc/validation.tex:62-65, 133-136,
 * 200, 289, 474, 518) reproduced on the reference's shipped example meshes
 * (tests/golden/meshes, tests/test_oracle_known_answers.py).
 */
#ifndef FEMSHELL_ORACLE_H
#define FEMSHELL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* behaviour switches; the default (both set) is the reference as coded */
#define FSO_REF_Y21       1u /* SA:586: Y(2,1) = -2*x31*x31 (thesis: -2*x31*y31) */
#define FSO_REF_DRILL_MAX 2u /* SA:1035-1052: drilling = max(...)/1000 on every node block */
#define FSO_REF_DEFAULT   (FSO_REF_Y21 | FSO_REF_DRILL_MAX)

typedef struct fso_material {
    double nu;        /* Poisson's ratio   (-nu) */
    double E;         /* Young's modulus   (-e)  */
    double thickness; /* shell thickness   (-t)  */
    uint32_t flags;   /* FSO_REF_* */
} fso_material;

/* SA:273-294 */
void fso_material_matrices(const fso_material *mat, double Dm[9], double Dp[9]);

/* intermediate results of one TRI3 element, for golden comparisons */
typedef struct fso_tri3_parts {
    double trafo[9];   /* rows = local x,y,z axes              (SA:378-384) */
    double transUV[6]; /* 3x2, local coords of B and C          (SA:324-329, 391) */
    double dphi[6];    /* 3x2: (x12,y12),(x31,y31),(x23,y23)    (SA:405-411) */
    double area;
    double Ke_m[36];   /* 6x6 membrane                          (SA:443-468) */
    double Ke_p[81];   /* 9x9 plate (Specht)                    (SA:555-603) */
    double K_local[324];     /* 18x18 node-major, local axes    (SA:999-1053) */
    double K_global_nm[324]; /* 18x18 node-major, global axes   (SA:1084-1102) */
} fso_tri3_parts;

/* Specht B~ (3x9, row-major) at area coordinates (L1,L2); C = squared side
 * lengths (|12|^2,|31|^2,|23|^2), dphi as above.  SA:698-891 */
void fso_tri3_specht_B(const double C[3], double L1, double L2, const double dphi[6], double B[27]);

/* Full TRI3 element: xyz = 3 nodes x 3 coords.  Ke = 18x18 row-major in the
 * reference's variable-major element ordering Ke(3*alpha+i, 3*beta+j)
 * (SA:1105-1109).  parts may be NULL.  Returns 0, or -1 for a degenerate element. */
int fso_element_tri3(const double xyz[9], const fso_material *mat, double Ke[324], fso_tri3_parts *parts);

/* Full QUAD4 element (bilinear membrane + DKQ plate): xyz = 4 nodes x 3.
 * Ke = 24x24 variable-major.  Ke_m (8x8), Ke_p (12x12), K_global_nm (24x24
 * node-major, global axes) may each be NULL. */
int fso_element_quad4(const double xyz[12], const fso_material *mat, double Ke[576],
                      double *Ke_m, double *Ke_p, double *K_global_nm);

/* ---- global assembly (SA:1160-1233) into 6x6-block CSR, node-major dofs -------- */

/* Pattern: block (a,b) exists iff nodes a,b share an element.  Columns sorted
 * ascending.  Call with colidx == NULL to get rowptr (n_nodes+1) and the
 * block count (return value); call again with colidx to fill it. */
int64_t fso_bsr_pattern(int32_t n_nodes, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                        const int32_t *quad, int32_t *rowptr, int32_t *colidx);

/* Assemble K (vals: nnzb x 36, row-major inside each block) and F (6*n_nodes).
 * dirichlet: one byte per node, bit v set = dof v (u,v,w,tx,ty,tz) fixed to 0
 * (SA:90-120 + libMesh constrain_element_matrix_and_vector: element row/col
 * zeroed, diagonal 1 per touching element, rhs 0).  loads: n_nodes x 6 nodal
 * forces/moments (SA:1118-1153: every node contributes exactly once).
 * Returns 0, or -(e+1) if element e is degenerate. */
int fso_assemble_bsr(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                     int32_t n_quad, const int32_t *quad, const fso_material *mat,
                     const uint8_t *dirichlet, const double *loads, const int32_t *rowptr,
                     const int32_t *colidx, double *vals, double *F);

/* y = K x */
void fso_bsr_spmv(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx,
                  const double *vals, const double *x, double *y);

typedef struct fso_pcg_info {
    int32_t iterations;
    int32_t converged; /* 1 = ||r|| <= rtol*||b||, 0 = max_it reached, -1 = breakdown (p.Ap <= 0) */
    double rel_residual; /* recurrence ||r||/||b|| at exit */
    double seconds;      /* wall time of the iteration loop */
} fso_pcg_info;

/* 6x6-block-Jacobi preconditioned CG, x0 = 0, stop when ||r||_2 <= rtol*||b||_2.
 * resid_hist (may be NULL) receives ||r||/||b|| after each iteration (max_it entries). */
int fso_pcg_block_jacobi(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx,
                         const double *vals, const double *b, double rtol, int32_t max_it,
                         double *x, double *resid_hist, fso_pcg_info *info);

/* CPU-baseline controls (bench.py's cpu_baseline leg).  The library built by the default `make` target is serial
 * (-O2 -ffp-contract=off: the checker of the tests); `make fast` builds libfemshell_oracle_fast.so from the same
 * source with -O3 -march=native -fopenmp on the machine that runs it, where fso_set_threads(n) makes the assembly
 * (contiguous node ranges per thread, like the reference's MPI ranks) and the PCG kernels run on n threads. */
void fso_set_threads(int n);
int fso_threads(void);
/* Specht curvatures: tabulated per Gauss point (default; filled from the polynomial derivation) or rebuilt from
 * the polynomials for every element and Gauss point (1: the cross-check of the tables, about 20x slower) */
void fso_set_specht_polynomial(int on);
void fso_init_tables(void);

/* timing helper for bench.py's cpu_baseline leg: assemble element matrices of
 * the first n_sample triangles `repeat` times, return elements per second */
double fso_time_assembly(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                         const fso_material *mat, const uint8_t *dirichlet, const double *loads,
                         const int32_t *rowptr, const int32_t *colidx, double *vals, double *F,
                         int32_t repeat);

/* the threaded assembly keeps per-thread element lists keyed on the connectivity pointers: drop them before those arrays
 * are freed or rewritten */
void fso_drop_thread_lists(void);
/* STREAM triad on the host threads, GB/s (24 bytes per element), best of reps: the bandwidth beside the CPU baseline */
double fso_stream_triad(int64_t n, int32_t reps);

#ifdef __cplusplus
}
#endif
#endif
