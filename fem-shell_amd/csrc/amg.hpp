// amg.hpp -- smoothed-aggregation multigrid preconditioner of the CG solve (opt-in; the default stays the 6x6
// block-Jacobi preconditioner whose iterates are the oracle's).
//
// Why it exists: the reference hands K u = F to PETSc's KSP with whatever -ksp_type/-pc_type the user passes
// (fem-shell.cpp:130-138, doc/implementation.tex:68-72).  With point-block Jacobi the iteration count of the shell
// systems grows with the element count (x3.7 per mesh doubling): about 1e6 iterations on the
// 4M-triangle meshes of BASELINE.json.  The hierarchy below keeps it near 100.
//
// Method (Vanek, Mandel, Brezina 1996): every level is again a matrix of 6x6 node blocks, so the sliced block ELL
// layout, k_spmv and the block-Jacobi kernels serve all levels.
//   * near-null space B: the six rigid-body modes of the shell (rows of fixed dofs zeroed)
//   * greedy distance-1 aggregation of the block graph, tentative prolongator by a QR factorisation of B per
//     aggregate (coarse B = the R factors), one damped block-Jacobi smoothing step P = (I - 4/(3 lam) D^-1 A) P0
//   * Galerkin operators A_c = P^T A P
//   * smoother: Chebyshev polynomial in D^-1 A (D = 6x6 diagonal blocks) on [lam/ratio, lam], lam from a power
//     iteration on the device; coarsest level: explicit dense inverse
//   * cycles: V, or K (two flexible-CG steps per coarse level, Notay & Vassilevski 2008) -- the 4th-order bending
//     part loses a factor of about 2.3 in iterations per level with V cycles, the K cycle keeps the two-grid rate
// Setup: the first coarsening step computes its numbers on the device from K where it lies (amg_device_setup.cpp: the
// host does the integer work on the block graph, kernels fill P, A P, R and the Galerkin operator -- the latter on the
// matrix cores); the remaining, nine times smaller levels use the host algebra of this file (amg_setup.cpp), which also
// serves FEMSHELL_AMG_SETUP=host.  Level operators are stored like K: diagonal + upper blocks.  The cycle runs on the
// device (amg_kernels.hip, amg_solve.cpp).
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "amg_patch.hpp"
#include "plan.hpp"

namespace femshell {

// arrays that stay uninitialised on resize (RawVec, plan.hpp): the gigabyte-sized value arrays and the large index arrays
// of the setup are written in full by parallel loops; a serial zero fill first would cost as much as the loop itself
using ValueArray = RawVec<double>;

// block CSR with 6x6 blocks (row-major inside a block), columns ascending within a row
struct Bsr {
    int32_t nr = 0, nc = 0; // block rows / block columns
    std::vector<int64_t> ptr;
    RawVec<int32_t> col;
    ValueArray val;
    int64_t nnzb() const { return (int64_t)col.size(); }
};

// runs f(begin, end) over [0,n) split into contiguous chunks on the host's hardware threads
void parallel_chunks(int64_t n, const std::function<void(int64_t, int64_t)> &f, int64_t min_chunk = 256);
int host_threads();

// rigid-body modes: B[n][6 dofs][6 modes] about the centroid of xyz; rows of fixed dofs (dmask bit v) are zero
// normals (optional, n x 3 unit vectors or zeros): the rotational parts of the three rotation modes are projected onto
// the tangent plane of each node (see amg_setup.cpp)
void rigid_body_modes(int32_t n, const double *xyz, const uint8_t *dmask, std::vector<double> *B, const double *normals = nullptr);
// the point the rotation modes turn about: the mean of the node coordinates (the device generates the modes of the finest
// level from it, amg_kernels.hip near_null_row)
void mesh_centre(int32_t n, const double *xyz, double c[3]);
// area-weighted unit normals of the nodes [0, n) from the elements of a mesh in the same numbering (tri: 3 ids, quad: 4 ids
// per element; nodes >= n are neighbours whose coordinates are in xyz too)
void node_normals(int32_t n, const double *xyz, int64_t n_tri, const int32_t *tri, int64_t n_quad, const int32_t *quad,
                  std::vector<double> *normals);

// ... the same array (bit for bit) for the owned nodes of a plan, from the gather lists of its diagonal slots
void node_normals_plan(const Plan &p, RawVec<double> *normals);

// greedy distance-1 aggregation of the block graph of A; returns the number of aggregates
// visit: order in which the greedy passes visit the nodes (visit[v] = node visited v-th); nullptr: index order, or
// breadth-first order when the numbering is scattered (aggregation_order)
int32_t aggregate_nodes(const Bsr &A, std::vector<int32_t> *agg, const std::vector<int32_t> *visit = nullptr);

// true when aggregate_nodes would run its greedy passes on the unfiltered graph, in one piece, in index order (or the caller's
// visiting order): widest_row = longest row (the node itself included), edges / neighbour_distance_sum as aggregation_order takes them
bool aggregation_is_plain(int32_t n, int64_t widest_row, int64_t edges, double neighbour_distance_sum, bool have_visit);

// ---- clusters of rigidly coupled nodes (amg_patch.hpp): the patch smoother of shells of poor element quality
// Clusters from the rigid edges (sigma2 > tau^2; each pair once, a < c): strongest first -- ties: lower a, then lower c --, two
// clusters are united while the union stays within max_nodes.  label[i] = cluster of node i or -1; clusters are numbered by their
// smallest node; members of cluster k: nodes[ptr[k] .. ptr[k+1]) ascending.  Returns the number of clusters.
int32_t patch_clusters(int32_t n, std::vector<PatchEdge> edges, int max_nodes, std::vector<int32_t> *label, std::vector<int32_t> *ptr,
                       std::vector<int32_t> *nodes);
// rigid edges of a host matrix (tests, levels coarsened on the host): patch_sigma2 over the blocks above the diagonal
void patch_edges_host(const Bsr &A, const std::vector<double> &Dinv, double tau, std::vector<PatchEdge> *edges);
// M_c = (A_cc)^-1 - blockdiag(D_i^-1) of every cluster from its dense diagonal block Bc (row-major (6 m)^2 at moff[c], overwritten);
// a block that is not positive definite gives M_c = 0 (the point blocks alone act on its nodes).  Returns the clusters that fell back.
int32_t patch_matrices(const std::vector<int32_t> &ptr, const std::vector<int64_t> &moff, const double *dinv_of_member /* 36 per member */,
                       double *Bc);
// aggregation with every cluster glued into one node first: the greedy passes run on the quotient graph (clusters and single nodes
// numbered in the order the visiting order meets them), a cluster's nodes share their quotient node's aggregate
int32_t aggregate_nodes_glued(const Bsr &A, const std::vector<int32_t> &label, std::vector<int32_t> *agg,
                              const std::vector<int32_t> *visit = nullptr);

// tentative prolongator: per aggregate B_agg = Q R (modified Gram-Schmidt, two passes; dependent columns give a
// zero column of Q and a zero diagonal of R).  Q[n][6][6] (row block of node n, column block agg[n]),
// Bc[na][6][6] = R
void tentative_prolongator(const std::vector<int32_t> &agg, int32_t na, const std::vector<double> &B,
                           std::vector<double> *Q, std::vector<double> *Bc);

// inverse of the 6x6 diagonal blocks (Cholesky; identity for blocks that are not positive definite)
void block_diagonal_inverse(const Bsr &A, std::vector<double> *Dinv);

void bsr_multiply(const Bsr &A, const Bsr &B, Bsr *C);  // C = A B
void bsr_transpose(const Bsr &A, Bsr *T);

// P = P0 - omega * Dinv * (A * P0), P0 given by (agg, Q)
void smoothed_prolongator(const Bsr &A, const std::vector<double> &Dinv, const std::vector<int32_t> &agg, int32_t na,
                          const std::vector<double> &Q, double omega, Bsr *P);

// Ac = P^T A P (R = P^T is returned too); coarse dofs without any fine support (zero column of P) get a unit diagonal
void galerkin_product(const Bsr &A, const Bsr &P, Bsr *R, Bsr *Ac);

// dense inverse of a small SPD block matrix (coarsest level), n = 6*A.nr; false when A is not positive definite
bool dense_inverse(const Bsr &A, std::vector<double> *inv);

// sliced block ELL image of a BSR matrix in the device layout of plan.hpp (32 block rows per slice; with
// diag_first the diagonal block sits in slot 0 as k_block_jacobi expects); padding slots hold zero blocks and
// point at column pad_col
struct SlicedEll {
    int32_t n_rows = 0, n_pad = 0, n_slices = 0, max_width = 0;
    std::vector<int32_t> slice_width;
    std::vector<int64_t> slice_base;
    std::vector<int32_t> cols;
    ValueArray vals;
};
void pack_sliced_ell(const Bsr &A, bool diag_first, SlicedEll *out);

// symmetric storage of a square operator with a symmetric pattern (coarse level matrices): the diagonal block in slot
// 0 and the blocks (a, c) with c > a; the in-lists tell row c which stored blocks (a, c) act on it through their
// transpose (plan.hpp describes the layout; k_spmv_sym / k_sym_gather multiply with it)
struct SlicedEllSym : SlicedEll {
    int32_t max_in_width = 0;
    std::vector<int32_t> in_width;
    std::vector<int64_t> in_base;
    RawVec<int32_t> in_slots, in_rows;
};
void pack_sliced_ell_sym(const Bsr &A, SlicedEllSym *out);
// in-lists of a diagonal-first upper pattern given as ELL arrays (rows kept their blocks (a, c >= a) only)
void build_in_lists(int32_t n_rows, const std::vector<int32_t> &slice_width, const std::vector<int64_t> &slice_base,
                    const int32_t *cols, const std::vector<uint8_t> &count, SlicedEllSym *out);
// A += strict upper part transposed (A holds the diagonal and upper blocks of a symmetric matrix): full BSR
void mirror_upper(Bsr *A);

} // namespace femshell
