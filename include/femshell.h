/*
 * femshell.h -- C ABI of libfemshell, the MI355X (gfx950) implementation of
 * fem-shell's hot path: per-element flat-shell stiffness assembly and the
 * sparse solve for nodal displacements.
 *
 * This is the drop-in boundary.  Every entry point names the reference
 * interface it replaces ("SA" = src/fem-shell/fem-shell.cpp, "PC" =
 * src/fem-shell/preCICE/fem-shell_precice.cpp of precice/fem-shell).  The
 * reference-side bindings (libMesh assemble callback, LinearSolver subclass,
 * preCICE adapter loop) are shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, host pointers, caller-owned buffers; the library copies in/out
 *  - every call returns FEMSHELL_OK (0) or a negative femshell_status; the text
 *    of the last error of the calling thread is femshell_last_error()
 *  - no exceptions cross the boundary
 *  - one context per host thread / MPI-style rank; a context is not thread-safe,
 *    distinct contexts are independent
 *  - all floating point is IEEE double (libMesh Real/Number), node ids are
 *    int32, dof of (node, var) is 6*node + var with var = u,v,w,tx,ty,tz
 *    (the layout of build_solution_vector, SA:141, 163-169)
 *  - there is NO CPU fallback: without a HIP device every compute call fails
 *    with FEMSHELL_ERR_NO_DEVICE
 */
#ifndef FEMSHELL_H
#define FEMSHELL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEMSHELL_VERSION 2

typedef enum femshell_status {
    FEMSHELL_OK = 0,
    FEMSHELL_ERR_INVALID = -1,     /* bad argument / call order */
    FEMSHELL_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    FEMSHELL_ERR_HIP = -3,         /* HIP runtime error */
    FEMSHELL_ERR_MESH = -4,        /* index out of range, degenerate element */
    FEMSHELL_ERR_BREAKDOWN = -5,   /* CG breakdown: p.Ap <= 0 (matrix not SPD) */
    FEMSHELL_ERR_COMM = -6,        /* RCCL error */
    FEMSHELL_ERR_UNSUPPORTED = -7
} femshell_status;

/* behaviour flags; FEMSHELL_REF_DEFAULT reproduces the reference as coded */
#define FEMSHELL_REF_Y21        0x1u /* SA:586: Y(2,1) = -2*x31*x31 instead of the thesis' -2*x31*y31 */
#define FEMSHELL_REF_DRILL_MAX  0x2u /* SA:1035-1052: drilling stiffness max(..)/1000 on every node block */
#define FEMSHELL_REASSEMBLE_EACH_SOLVE 0x4u /* PC:271: rebuild K on every solve (the reference does; K is constant) */
#define FEMSHELL_REF_DEFAULT    (FEMSHELL_REF_Y21 | FEMSHELL_REF_DRILL_MAX)
/* node renumbering inside the library (any number of ranks): rows are stored along a Morton curve through the mesh /
 * in reverse Cuthill-McKee order, so that the x entries a 32-node slice gathers are close together whatever the caller's
 * numbering is.  libMesh renumbers for locality by default; the reference switches it off only because its force file is
 * indexed by the original ids (SA:36).  Every node-indexed argument of this ABI keeps the caller's numbering.
 * FEMSHELL_REORDER=morton|rcm in the environment sets the flag for contexts created without one. */
#define FEMSHELL_REORDER_MORTON 0x10u
#define FEMSHELL_REORDER_RCM    0x20u

typedef struct femshell_config {
    double nu;          /* Poisson's ratio  -nu (SA:217) */
    double E;           /* Young's modulus  -e  (SA:225) */
    double thickness;   /* shell thickness  -t  (SA:233) */
    uint32_t flags;     /* FEMSHELL_REF_* | FEMSHELL_REASSEMBLE_EACH_SOLVE */
    int32_t device;     /* HIP device ordinal; -1 = the calling thread's current device */
    int32_t rank;       /* this process' index in the row partition (0 for one GPU) */
    int32_t world_size; /* number of processes = GPUs sharing the mesh (1 for one GPU) */
} femshell_config;

typedef struct femshell_ctx femshell_ctx;

/* replaces: global state nu/em/thickness/Dp/Dm + initMaterialMatrices (SA:273-294, fem-shell.h:46-52) */
int femshell_create(const femshell_config *cfg, femshell_ctx **out);
int femshell_destroy(femshell_ctx *ctx);
const char *femshell_last_error(void);

/* replaces: what assemble_elasticity reads through es.get_mesh()/DofMap (SA:1166-1205):
 * the replicated mesh (SA:35-37).  xyz[n_nodes][3]; tri[n_tri][3]; quad[n_quad][4]
 * (either count may be 0).  Every rank passes the same global mesh; the library keeps
 * the node rows [femshell_row_begin, femshell_row_end) of its rank plus the ghost
 * nodes/elements they touch, and builds the block sparsity (libMesh does this in
 * EquationSystems::init, SA:125). */
int femshell_set_mesh(femshell_ctx *ctx, int32_t n_nodes, const double *xyz, int32_t n_tri,
                      const int32_t *tri, int32_t n_quad, const int32_t *quad);

/* replaces: DirichletBoundary {0,20}->u,v,w and {1,21}->all six (SA:90-120).  mask6 bit v
 * fixes dof v of the node to 0.  node_ids == NULL: mask6 has one byte per node (n == n_nodes).
 * Calling it again replaces the previous set. */
int femshell_set_dirichlet(femshell_ctx *ctx, int32_t n, const int32_t *node_ids, const uint8_t *mask6);

/* replaces: the global `forces` vector read by contribRHS (SA:44-67, 1118-1153; PC:1377-1438).
 * f6[n][6] nodal forces and moments; node_ids == NULL: one row per node (n == n_nodes).
 * Nodes not listed get zero load.  Only the right-hand side is rebuilt. */
int femshell_set_loads(femshell_ctx *ctx, int32_t n, const int32_t *node_ids, const double *f6);

/* replaces: assemble_elasticity (SA:1160-1233): element stiffness (initElement, calcPlane,
 * calcPlate, constructStiffnessMatrix, localToGlobalTrafo), constraint handling, add_matrix /
 * add_vector.  K and F stay in HBM. */
int femshell_assemble(femshell_ctx *ctx);
/* The same assembly without the round trip that collects its status: the kernels are enqueued and the call returns.  A
 * degenerate element (FEMSHELL_ERR_MESH) -- on any rank of the row partition -- is reported by the next call on this context
 * that synchronises, reads K or changes its inputs: femshell_sync, femshell_solve, femshell_assemble, femshell_export_bsr,
 * femshell_spmv, femshell_residual, femshell_set_dirichlet, femshell_set_loads.  For callers that assemble in a loop
 * (re-meshing, time stepping with changing geometry): on 8 ranks the agreement on the status costs as much as the 0.08 ms
 * assembly step itself.  Multi-rank contexts: every rank makes the same sequence of calls, as for femshell_assemble. */
int femshell_assemble_async(femshell_ctx *ctx);

typedef struct femshell_solve_info {
    int32_t iterations;     /* CG iterations performed */
    int32_t converged;      /* 1: ||r|| <= rtol*||b|| (multigrid with refinement: <= 100 rtol*||b|| after the first phase,
                               then the refinement pass -- see femshell_pc_options::refine_passes), 0: max_it reached */
    double rel_residual;    /* recurrence ||r||_2 / ||b||_2 at exit (what the stopping rule tests; multigrid with
                               refinement: at the end of the first phase) */
    double true_rel_residual; /* ||b - K u||_2 / ||b||_2 recomputed explicitly after a converged solve, -1 if not
                               computed (rtol <= 0, iteration limit, breakdown).  On ill-conditioned shells the
                               recurrence drifts and this floors near kappa*eps; it is reported, not enforced:
                               restarting CG from the explicit residual was tried and made the error worse */
    double assemble_seconds;/* device time of the assembly done inside this call (0 if reused) */
    double setup_seconds;   /* block-Jacobi factorisation */
    double solve_seconds;   /* CG loop, device time */
    double bytes_per_iteration; /* algorithmic HBM bytes of one CG iteration on this rank */
    int32_t pc_type;        /* femshell_pc_type the solve ran with */
    int32_t amg_levels;     /* levels of the multigrid hierarchy (0 with block-Jacobi) */
    double pc_setup_seconds;/* host + device time of the multigrid setup done inside this call (0 if reused) */
    double operator_complexity; /* sum of the level matrices' blocks / blocks of K (0 with block-Jacobi) */
    /* error estimate of the multigrid-preconditioned solve (version 2 of this struct; -1 / 0 when no refinement pass ran):
     * a refinement pass solves K e = b - K x (residual in double-double) and adds e, so ||e||/||x|| of the last pass
     * measures the relative displacement error of the iterate BEFORE that pass, and the pass leaves behind about
     * that times the factor by which it reduced the residual of its correction equation.  Checked against manufactured
     * solutions at the 4M-triangle sizes (tests/test_gpu_fullsize.py, bench.py time_to_solution.manufactured). */
    int32_t refine_passes_done;       /* refinement passes that ran (<= femshell_pc_options::refine_passes + 1) */
    int32_t pc_fp64_fallback;         /* 1: the flexible CG broke down (p.Ap <= 0) under a multigrid hierarchy that keeps
                                         single-precision copies, and the solve ran again from the start with an all-FP64
                                         hierarchy (very thin shells); the context keeps that choice until a new mesh or
                                         preconditioner is set.  (the reserved word of version 2 of this struct) */
    double refine_correction_rel;     /* ||e||_2 / ||x||_2 of the last pass */
    double refine_residual_reduction; /* ||rhs - K e|| / ||rhs|| (recurrence) the last pass stopped at */
    double error_estimate;            /* refine_correction_rel * refine_residual_reduction: estimated relative error of u */
} femshell_solve_info;

/* ---- preconditioner ------------------------------------------------------------------
 * replaces: the -ksp_type / -pc_type options the reference passes through to PETSc
 * (doc/implementation.tex:68-72; equation_systems.parameters stay untouched, SA:130-133).
 * The Krylov method is always CG (K is SPD).  FEMSHELL_PC_BLOCK_JACOBI (default) is the 6x6
 * point-block Jacobi whose iterates the CPU oracle reproduces step by step; its iteration
 * count grows with the element count.  FEMSHELL_PC_AMG is a smoothed-aggregation multigrid
 * (rigid-body modes, Chebyshev/block-Jacobi smoothing, V or K cycle) around which the solve
 * runs a flexible CG: the answer a user of `-pc_type gamg` expects, with iteration counts
 * that stay near 100 up to the 4M-triangle meshes.  On row-partitioned contexts the hierarchy is
 * row-partitioned like K (aggregates never span ranks; levels above 60,000 nodes split over the
 * ranks, the rest replicated): iteration counts within a few of the single-rank ones.  What the
 * cycle only smooths or transfers with is kept in single precision (FP64 arithmetic); should the
 * flexible CG break down under it, the solve runs again with an all-FP64 hierarchy
 * (femshell_solve_info::pc_fp64_fallback).
 * The environment variable FEMSHELL_PC=amg|jacobi sets the default of new contexts. */
typedef enum femshell_pc_type { FEMSHELL_PC_BLOCK_JACOBI = 0, FEMSHELL_PC_AMG = 1 } femshell_pc_type;
typedef enum femshell_cycle { FEMSHELL_CYCLE_V = 0, FEMSHELL_CYCLE_K = 1 } femshell_cycle;
typedef struct femshell_pc_options {
    int32_t type;            /* femshell_pc_type */
    int32_t cycle;           /* femshell_cycle (default K: two flexible-CG steps per coarse level) */
    int32_t smoother_degree; /* Chebyshev degree on the finest level (default 3) */
    int32_t coarse_degree;   /* Chebyshev degree on the coarser levels (default 4) */
    int32_t coarsest_nodes;  /* coarsening stops at the first level of at most this many nodes; dense inverse there (default
                                and maximum 1400: the K cycle visits its last levels 8-16 times per iteration and each
                                visit is a chain of launches, so an exact solve of 8400 dofs -- one dense matrix-vector
                                product -- is cheaper than two more levels; inverses beyond 64 nodes are computed on the
                                matrix cores) */
    int32_t max_levels;      /* default 12 */
    int32_t refine_passes;   /* iterative refinement after convergence, at most this many passes (default 1; 0 = off):
                                the residual of the iterate is evaluated in double-double and the correction equation
                                solved by the same method -- rtol bounds the residual, and on these systems the
                                displacement error sits one to two decades above it (4M triangles, manufactured
                                solution: 4e-9 at a residual of 9e-11 ||b||) -- until femshell_solve_info::error_estimate
                                of the pass so far, ||e|| / ||x|| x the drop of its residual, is a fifth of rtol (a drop
                                between 1e-2 and 1e-6; 1e-4 flat before round 5 and with FEMSHELL_REFINE_ADAPTIVE=0).  The
                                first pass always runs, further ones (one more than this number at most) while
                                femshell_solve_info::error_estimate exceeds rtol.  Plain FP64 CG stalls at a displacement
                                error of kappa*eps -- 2e-10 on the 250k-triangle roof -- one pass brings it to 1e-13.
                                With refinement on, the recurrence of the first phase runs to 100 rtol only: what rtol
                                is for is the displacement error, the pass reduces the error of whatever iterate it
                                starts from by its drop, and the digits between 100 rtol and rtol are the ones the
                                recurrence's rounding noise spoils (4M triangles: 132 instead of 148 iterations, error
                                estimate 4e-12; manufactured solutions 1e-12 / 2e-11) */
    int32_t reserved;
    double eig_ratio;        /* the smoother targets [lambda_max/eig_ratio, lambda_max] of D^-1 A (default 30) */
} femshell_pc_options;
/* fills *out with the defaults of `type` */
int femshell_pc_defaults(int32_t type, femshell_pc_options *out);
/* takes effect at the next femshell_solve; the hierarchy is rebuilt whenever K changes */
int femshell_set_preconditioner(femshell_ctx *ctx, const femshell_pc_options *opt);

/* inspection of the multigrid hierarchy of the last solve (tests; host copies are kept for
 * matrices of up to 2M blocks): level 0 is K itself */
typedef struct femshell_amg_level_info {
    int32_t n_nodes;        /* block rows of the level */
    int32_t n_coarse;       /* aggregates = block rows of the next level (0 on the coarsest) */
    int64_t nnz_blocks;     /* blocks of the level matrix */
    int64_t p_blocks;       /* blocks of the prolongator to this level from the next (0 on the coarsest) */
    double lambda_max;      /* upper end of the smoother's interval for D^-1 A */
} femshell_amg_level_info;
int32_t femshell_amg_levels(femshell_ctx *ctx);
int femshell_amg_level(femshell_ctx *ctx, int32_t level, femshell_amg_level_info *info);
enum { FEMSHELL_AMG_AGGREGATES = 0, /* int32 [n_nodes] */
       FEMSHELL_AMG_A_ROWPTR, FEMSHELL_AMG_A_COLS, FEMSHELL_AMG_A_VALS,   /* int64 [n+1], int32 [nnzb], double [nnzb*36] */
       FEMSHELL_AMG_P_ROWPTR, FEMSHELL_AMG_P_COLS, FEMSHELL_AMG_P_VALS,
       /* the coarsest level only: the dense inverse the cycle multiplies with, double [n][n] row-major (n = 6 x its nodes;
        * the operator it inverts is that level's FEMSHELL_AMG_A_*, kept at every problem size) */
       FEMSHELL_AMG_COARSE_INVERSE,
       /* level 0: the cluster of rigidly coupled nodes every node belongs to, or -1 (the patch smoother of shells of poor element
        * quality, csrc/amg_patch.hpp); -1 as a count: the level has no clusters.  int32 [n_nodes], internal numbering */
       FEMSHELL_AMG_PATCH_LABELS };
/* returns the element count of the array (-1: not available); copies it to out when out != NULL.  The host copies behind
 * the A_* / P_* / AGGREGATES arrays are kept for problems of up to 300,000 blocks of K (up to 2,000,000 with
 * FEMSHELL_AMG_KEEP_HOST=1 in the environment at setup): an inspection interface, not part of the solve */
int64_t femshell_amg_export(femshell_ctx *ctx, int32_t level, int32_t which, void *out);
/* the patch smoother of level 0 of the last setup (csrc/amg_patch.hpp; FEMSHELL_AMG_PATCH_TAU / _MAX in the environment): out[0] =
 * rigid edges found (sigma_max of the scaled coupling above tau), out[1] = clusters, out[2] = nodes in clusters, out[3] = clusters
 * whose diagonal block was not positive definite (they keep their point blocks), out[4] = tau, out[5] = nodes per cluster at most.
 * All zero on a mesh without such edges: nothing of the method runs then. */
int femshell_amg_patch_info(femshell_ctx *ctx, double out[6]);
/* device timings of the first coarsening step of the last setup: out[0..3] = milliseconds of the prolongator, A P,
 * restriction and Galerkin kernels, out[4] = useful flops of the Galerkin product, out[5] = flops issued on the
 * matrix cores (v_mfma_f64_16x16x4_f64 tiles; 0 when the vector-ALU kernel ran), out[6] = 1 if the matrix cores ran */
int femshell_amg_setup_stats(femshell_ctx *ctx, double out[7]);
/* Where the integer work of the last multigrid setup's coarsening steps ran (the setup is part of what replaces
 * equation_systems.solve(), fem-shell.cpp:138 -- PETSc's KSPSetUp): out[0] = steps whose patterns were built in HBM
 * (csrc/amg_symbolic.hip), out[1] = steps that took the host's lists although the device was asked (a row beyond the lane sets),
 * out[2] = steps on the host's lists by rule (clusters of rigidly coupled nodes, FEMSHELL_AMG_SYMBOLIC=host). */
int femshell_amg_symbolic_info(femshell_ctx *ctx, int32_t out[3]);
/* the dense inverse of the coarsest operator when it was computed on the matrix cores (csrc/amg_dense.hip; symmetric block
 * sweeps on v_mfma_f64_16x16x4_f64): out[0] = dofs n (0: the host inverted a small operator), out[1] = milliseconds,
 * out[2] = flops issued on the matrix cores, out[3] = n^3 (the flops of a symmetric inversion), out[4] = dropped
 * (semi-definite) directions, out[5] = bytes of the lower triangle read and written over all steps */
int femshell_amg_dense_stats(femshell_ctx *ctx, double out[6]);
/* Multigrid hierarchy of a row-partitioned context (the reference's PETSc preconditioner lives on the distributed matrix,
 * doc/implementation.tex:463-472): out[0] = levels whose rows are split over the ranks (0: single-rank context), out[1] =
 * bytes of this rank's operators on those levels (they shrink with the rank count), out[2] = bytes of the levels below,
 * which every rank holds in full (dense inverse included), out[3] / out[4] = this rank's rows / ghost rows on the last
 * row-partitioned level, out[5] = nodes of the first replicated level. */
int femshell_amg_partition_info(femshell_ctx *ctx, double out[6]);
/* Algorithmic HBM bytes of ONE multigrid cycle (one application of the preconditioner inside femshell_solve, which replaces the
 * PETSc preconditioner behind equation_systems.solve(), fem-shell.cpp:138), level by level: per_level[l] = what the cycle streams on
 * level l over all its visits per outer iteration -- smoothing products on the single-precision copies where the level keeps them,
 * residual increments, transfers, the K cycle's FP64 products, the dense solve of the coarsest level (csrc/amg_solve.cpp
 * amg_cycle_bytes spells the model out).  Returns the number of levels (< 0: error); at most `cap` entries are written.
 * femshell_solve_info::bytes_per_iteration = their sum + the Krylov method's own passes on level 0. */
int32_t femshell_amg_cycle_bytes(femshell_ctx *ctx, double *per_level, int32_t cap);
/* which assembly kernel femshell_assemble launches for the mesh of this context (after femshell_set_mesh): 1 = the
 * pipelined one (k_assemble_pipe: meshes whose slices touch at most 150 elements -- 77 with quadrilaterals -- and whose
 * work items fit one round of its waves: structured meshes, unstructured ones numbered with locality), 0 = the two-phase
 * one (k_assemble: scattered numberings, full storage, FEMSHELL_ASM_PIPE=0); < 0: error.  Same results either way. */
int femshell_assembly_kernel(femshell_ctx *ctx);

/* replaces: equation_systems.solve() -> PETSc KSPSolve (SA:138, PC:271) followed by
 * build_solution_vector (SA:141; PC:274-280 broadcast): 6x6-block-Jacobi preconditioned CG,
 * x0 = 0 (or what femshell_set_initial_guess handed over for this solve), stop at ||r||_2 <= rtol*||b||_2 or max_it.
 * rtol <= 0 runs exactly max_it iterations.
 * u_out[n_nodes][6] receives the full solution on every rank; NULL leaves it in HBM
 * (fetch with femshell_get_solution).  Assembles first if needed.
 * Contexts with a communicator (femshell_comm_init) run the single-reduction form of the same method
 * (Chronopoulos & Gear: one all-reduce of r.z, r.r, z.Az per iteration); the iterates agree with the classic
 * recurrence up to rounding.  FEMSHELL_CG_SINGLE_REDUCTION=0/1 in the environment overrides the choice. */
int femshell_solve(femshell_ctx *ctx, double rtol, int32_t max_it, double *u_out,
                   femshell_solve_info *info);
int femshell_get_solution(femshell_ctx *ctx, double *u_out);
/* The NEXT femshell_solve of this context starts from u0 (n_nodes x 6, the caller's numbering; every rank passes the whole vector
 * and keeps its own rows) instead of from zero; u0 == NULL: from the solution of the context's previous solve, where it lies in
 * HBM (no transfer).  Consumed by that solve; femshell_set_mesh forgets it.
 * replaces: libMesh hands system.solution to KSPSolve as the initial guess (PetscLinearSolver: KSPSetInitialGuessNonzero), so
 * every equation_systems.solve() of the coupled adapter's loop (fem-shell_precice.cpp:271) starts from the displacements of the
 * last coupling iteration; the stand-alone program's one solve (fem-shell.cpp:138) starts from zero either way.
 * The stopping rule is unchanged (||r|| <= rtol ||b||, the refinement passes of the multigrid-preconditioned solve and their
 * error estimate): the first phase solves the correction equation K e = b - K u0, its right-hand side evaluated in
 * double-double, down to the threshold a solve from zero runs to.  Block-Jacobi: classic recurrence from r = b - K u0. */
int femshell_set_initial_guess(femshell_ctx *ctx, const double *u0);
/* ||r||/||b|| after each iteration of the last solve; returns the count written (<= cap) */
int32_t femshell_residual_history(femshell_ctx *ctx, double *hist, int32_t cap);

/* ---- parity / debug exports (single-rank contexts) --------------------------------- */

/* element matrices in the reference's variable-major element ordering Ke(n*alpha+i, n*beta+j)
 * (SA:1105-1109), unconstrained, as localToGlobalTrafo leaves them.  Triangles are elements
 * [0,n_tri), quads [n_tri, n_tri+n_quad); the range must not mix the two kinds.
 * Ke_out: count x 324 (TRI3) or count x 576 (QUAD4) doubles. */
int femshell_element_matrices(femshell_ctx *ctx, int32_t first, int32_t count, double *Ke_out);

int64_t femshell_nnz_blocks(femshell_ctx *ctx); /* number of 6x6 blocks of K on this rank */
/* K as block CSR with sorted columns (vals: nnzb x 36 row-major) and F, after femshell_assemble -- what
 * system.matrix / system.rhs hold after the callback (SA:1230-1231).  A rank exports the node rows
 * [femshell_row_begin, femshell_row_end) it owns: rowptr has row_end - row_begin + 1 entries, colidx holds global
 * node ids, F has 6 entries per owned row (one rank: the whole matrix). */
int femshell_export_bsr(femshell_ctx *ctx, int32_t *rowptr, int32_t *colidx, double *vals, double *F);
/* y = K x on the device */
int femshell_spmv(femshell_ctx *ctx, const double *x, double *y);
/* r = F - K x with products and row sums in double-double (the residual the iterative refinement of the
 * multigrid-preconditioned solve restarts from); x[n_nodes][6] */
int femshell_residual(femshell_ctx *ctx, const double *x, double *r);

/* ---- row partition over several GPUs (one process per GPU, RCCL over xGMI) ---------- */

int32_t femshell_row_begin(femshell_ctx *ctx); /* first owned node row */
int32_t femshell_row_end(femshell_ctx *ctx);   /* one past the last owned node row */
/* The caller's ids of the node rows this rank owns, in the order femshell_export_bsr gives them; returns their number
 * (ids_out may be NULL).  Without a renumbering flag: row_begin .. row_end - 1.  With FEMSHELL_REORDER_MORTON / _RCM on a
 * row-partitioned context the library partitions the RENUMBERED rows (every rank computes the same permutation of the whole
 * mesh, then takes its stretch: a compact patch of the mesh whatever the caller's numbering is -- libMesh partitions its
 * elements for locality the same way, doc/implementation.tex:103-124), so the owned nodes are not a range of the caller's ids:
 * this list names them.  Node-indexed arguments (femshell_set_dirichlet, femshell_set_loads, the solution) keep the caller's
 * numbering on every rank. */
int32_t femshell_owned_nodes(femshell_ctx *ctx, int32_t *ids_out);
/* rank 0 creates the id, the host program distributes it (e.g. a torch.distributed or MPI
 * broadcast), every rank then calls femshell_comm_init before femshell_set_mesh.
 * replaces: LibMeshInit / init.comm() (SA:28, 35) */
int femshell_comm_unique_id(uint8_t id_out[128]);
int femshell_comm_init(femshell_ctx *ctx, const uint8_t id[128]);
/* ranks RCCL reports for this context's communicator (ncclCommCount); 0 when the context has none */
int32_t femshell_comm_ranks(femshell_ctx *ctx);
/* What femshell_comm_init checked on first contact, beyond the communicator itself: the communication patterns of a solve, once
 * each with a known answer and under the watchdog -- out_us[0]: a grouped ncclSend / ncclRecv ring on the halo stream beside an
 * ncclAllReduce of three words on the main stream (the reference's counterparts: PETSc's VecScatter halo and the MPI_Allreduce
 * inside KSPSolve, fem-shell.cpp:138); out_us[1]: grouped ncclBroadcast, one per rank (build_solution_vector + broadcast,
 * fem-shell.cpp:141, fem-shell_precice.cpp:277-280); out_us[2]: a lone all-reduce of three words.  Wall microseconds, enqueue
 * to completion.  Returns 1 when the self-test ran (a communicator exists), 0 when not (single-rank context), < 0 on error. */
int femshell_comm_selftest(femshell_ctx *ctx, double out_us[3]);
/* Communication this context has enqueued since the counters were last cleared: out[0] = grouped send/recv exchanges on the halo
 * stream (they run beside the interior slices of the product that needs them), out[1] = such exchanges on the main stream (in
 * the dependency chain: restrictions, prolongations, setup), out[2] = all-reduces, out[3] = grouped broadcasts (row gathers).
 * clear != 0 resets them.  What the reference leaves to PETSc (VecScatter, MPI_Allreduce inside KSPSolve, fem-shell.cpp:138) is
 * countable here: tests hold the per-iteration budget of the row-partitioned multigrid to these numbers.  Returns 1 with a
 * communicator, 0 without, < 0 on error. */
int femshell_comm_counters(femshell_ctx *ctx, int64_t out[4], int32_t clear);
/* ... and its volume: out[0] = bytes this rank handed to the sends of grouped exchanges (vector halos: 48 bytes per node another
 * rank reads; the multigrid setup: the rows of Q, P and A P of those nodes), out[1] = bytes it contributed to all-reduces and
 * broadcasts.  Same clearing and return values. */
int femshell_comm_bytes(femshell_ctx *ctx, int64_t out[2], int32_t clear);

/* ---- measurement ----------------------------------------------------------------- */

typedef enum femshell_kernel {
    FEMSHELL_KERNEL_ASSEMBLE = 0, /* element stiffness + block-row gather into K */
    FEMSHELL_KERNEL_SPMV = 1,     /* q = K p with fused p.q */
    FEMSHELL_KERNEL_CG_UPDATE = 2,/* x,r update + block-Jacobi apply + dots */
    FEMSHELL_KERNEL_CG_DIRECTION = 3 /* p = z + beta p */
} femshell_kernel;

/* mean duration of one launch of the kernel from HIP events on the library's stream, over `reps` launches:
 * the assembly kernel back to back; a CG kernel inside `reps` iterations of the recurrence on scratch vectors
 * (this rank's rows, no communication), an event pair around that kernel of every iteration -- where it runs,
 * not back to back with itself.  bytes_out = algorithmic HBM bytes of one launch on this rank (DESIGN.md
 * section "algorithmic bytes").  The state of a solve is not disturbed. */
int femshell_time_kernel(femshell_ctx *ctx, femshell_kernel which, int32_t reps, double *mean_ms_out,
                         double *bytes_out);

/* hipStreamSynchronize on the library's stream */
int femshell_sync(femshell_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
