// amg_patch.hpp -- patch smoother of the multigrid preconditioner for shells of poor element quality (round 6).
//
// Why.  On a random-point Delaunay shell (two of the reference's three input formats are unstructured: Gmsh and XDA written by other
// tools, doc/implementation.tex:63-124) nodes a hundredth of the mesh width apart are coupled so rigidly that their common motion
// is a mode of D^-1 A at 1e-2 -- below what the point-block Chebyshev smoother damps, and in no aggregate's coarse space when the
// greedy pass puts the two into different aggregates: > 1000 iterations, on one rank or several.
// What helps, measured in the numpy restatement (profiles/r05_cluster_block_smoother_experiment.txt, tools/lab/
// r06_patch_smoother_experiment.py): ONE smoother block per cluster of rigidly coupled nodes -- the exact inverse of the cluster's
// diagonal block of A where the point-block smoother has six 6 x 6 inverses -- in the level's Chebyshev smoother, in its spectral
// bound and in the smoothing of the prolongator, and the clusters glued into one aggregate each.
//
// When.  Edges above 0.8 are ordinary on stretched structured elements (a third of the edges of the pinched cylinder's 3 : 1 cells
// reach 0.836, the coupled flap's 5 : 1 cells 0.965) where the point-block method works; what it does not cope with are NEARLY
// COINCIDENT nodes, sigma > 0.98: 2.6 % of the edges of the random-point shells, 0.13 % of a jittered-grid Delaunay shell (90
// iterations without cluster blocks), none of any structured mesh.  The method switches itself on when more than 1 % of the pairs
// exceed 0.98 (FEMSHELL_AMG_PATCH_TRIGGER; amg_solve.cpp amg_build_patches) -- the ranks of a row partition decide together.
//
// How.  Clusters: the edges of the block graph with sigma_max(D_i^-1/2 A_ij D_j^-1/2) > tau (0.8), strongest first, united while a
// cluster stays within eight nodes (FEMSHELL_AMG_PATCH_TAU / _MAX; swept on 700 ... 50,000-point shells in four numberings:
// profiles/r06_patch_smoother.txt).  The device keeps its packed 6 x 6 inverses; a cluster c adds the dense matrix
//     M_c = (A_cc)^-1 - blockdiag(D_i^-1, i in c)
// and every application z = D^-1 r of the level is followed by z_c += M_c r_c for the clustered nodes (k_patch_correct: one wave per
// cluster) -- linear, so the Chebyshev steps d = a d + c D^-1 r, x += d take it as d += c M r, x += c M r behind the kernel that did the
// point-block part.  A level without clusters launches nothing: structured meshes keep their hierarchies and iterates bit for bit.
// Restated in oracle/amg_oracle.py (patch_*).  Level 0 only: the coarse operators of such a mesh are not the problem.  A level with
// clusters keeps FP64 copies throughout (the flexible CG broke down under the single-precision ones on every such shell tried).
// Row partitions: every rank clusters its own rows; a cluster that holds a node another rank reads smooths like the others but
// neither widens its members' rows of P nor is glued (PatchView::in_p; amg_solve.cpp says why).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FS_PATCH_HD __host__ __device__
#else
#define FS_PATCH_HD
#endif

namespace femshell {

constexpr int kPatchPowerSteps = 16; // power iteration of patch_sigma2
constexpr int kPatchMaxNodes = 10;   // largest cluster the kernels take (one wave: 60 of 64 lanes); default bound: 8

struct PatchEdge {
    int32_t a, c;   // a < c
    double sigma2;  // estimate of sigma_max^2 (below)
};

// sigma_max^2 of S = D_i^-1/2 A D_j^-1/2 for the 6 x 6 block A = A_ij (row-major) and the inverse diagonal blocks Di, Dj (full 6 x 6,
// symmetric): S S^T is similar to T = Di A Dj A^T.  trace(T) = ||S||_F^2 bounds sigma_max^2 from above and is returned when it is
// below tau2 already (nearly every block of a good mesh); else the norm ratio of the last of kPatchPowerSteps power steps on T from
// the vector of ones.  The same arithmetic on the host (tests, femshell_amg_host_patch_edges) and on the device (k_patch_sigma).
FS_PATCH_HD inline double patch_sigma2(const double Di[36], const double A[36], const double Dj[36], double tau2)
{
    double X[36], Y[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += Di[6 * i + k] * A[6 * k + j];
            X[6 * i + j] = s;
        }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += X[6 * i + k] * Dj[6 * k + j];
            Y[6 * i + j] = s;
        }
    double tr = 0.0;
    for (int e = 0; e < 36; e++) tr += Y[e] * A[e];
    if (!(tr >= tau2)) return tr;
    double v[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0}, lam = 0.0;
    for (int it = 0; it < kPatchPowerSteps; it++) {
        double w[6], u[6], nv = 0.0, nu = 0.0;
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int i = 0; i < 6; i++) s += A[6 * i + j] * v[i];
            w[j] = s;
        }
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
            for (int j = 0; j < 6; j++) s += Y[6 * i + j] * w[j];
            u[i] = s;
        }
        for (int i = 0; i < 6; i++) {
            nv += v[i] * v[i];
            nu += u[i] * u[i];
        }
        if (!(nu > 0.0) || !(nv > 0.0)) return 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
        lam = sqrt(nu / nv);
        const double inv = 1.0 / sqrt(nu);
#else
        lam = __builtin_sqrt(nu / nv);
        const double inv = 1.0 / __builtin_sqrt(nu);
#endif
        for (int i = 0; i < 6; i++) v[i] = u[i] * inv;
    }
    return lam;
}

// what the kernels see of a level's clusters
struct PatchView {
    int32_t n_clusters = 0;
    int32_t n_members = 0;          // nodes in clusters
    const int32_t *ptr = nullptr;   // n_clusters + 1: members of cluster c are nodes[ptr[c] .. ptr[c+1])
    const int32_t *nodes = nullptr; // local rows, ascending inside a cluster
    const int64_t *moff = nullptr;  // n_clusters: offset of M_c, a (6 m) x (6 m) row-major matrix, in M
    const double *M = nullptr;
    const int32_t *cluster_of = nullptr; // per member position: its cluster
    const uint8_t *in_p = nullptr;       // per cluster: does it take part in the smoothing of P (null: all do)
};

} // namespace femshell
