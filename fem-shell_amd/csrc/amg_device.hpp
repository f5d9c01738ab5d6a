// amg_device.hpp -- the multigrid hierarchy in HBM (amg_solve.cpp) as the context sees it.
#pragma once
#include "amg_patch.hpp"

#include <memory>
#include <vector>

#include "amg.hpp"
#include "amg_kernels.hpp"
#include "context.hpp"

namespace femshell {

// one sliced block ELL operator of the hierarchy (level matrix, prolongation or restriction)
struct AmgOperator {
    DevBuf<int32_t> slice_width, cols;
    DevBuf<int64_t> slice_base;
    DevBuf<uint8_t> count;  // real slots per row (operators whose pattern a coarsening step on the device built; else empty)
    DevBuf<double> vals;
    // symmetric storage (square level operators): in-lists and the transposed products beside the slots
    DevBuf<int32_t> in_width, in_slots, in_rows;
    DevBuf<int64_t> in_base;
    DevBuf<double> tbuf;
    DeviceMatrix dm{};      // the view k_spmv / k_spmv_sym / k_block_jacobi take
    int64_t nnzb = 0;       // real blocks
    int32_t n_cols_pad = 0; // padded block columns = nodes of the input vector
};

// Host copy of the PATTERN of a level operator that lives in HBM in sliced block ELL (values stay on the device): what the
// integer work of a device coarsening step needs.  Level 0 takes it from the plan, coarser levels from the pattern the
// previous step built for its Galerkin product.
struct HostEllPattern {
    int32_t n = 0;                     // rows (nodes)
    bool symmetric = false;            // diagonal + upper blocks stored, in-lists below
    std::vector<int32_t> slice_width;
    std::vector<int64_t> slice_base;
    RawVec<int32_t> cols;
    std::vector<uint8_t> count;        // real entries per row (slots 0 .. count-1)
    std::vector<int32_t> in_width;
    RawVec<int32_t> in_slots, in_rows;
    std::vector<int64_t> in_base;
    // Level 0 with its patterns built in HBM (amg_symbolic.hip): the host only runs the greedy passes of the aggregation over K's
    // pattern, which the plan holds already -- `borrowed` then points at the plan and cols / count / in_slots / in_rows above stay
    // empty (their copy was 6-8 ms of a 76 ms setup at 4M triangles: 80 MB).  A step that needs the host's lists after all
    // (clusters, a row too long for the lane sets) fills them first: pattern_of_plan.
    const Plan *borrowed = nullptr;
    bool empty() const { return slice_base.empty(); }
};

// Ghost exchange of the vectors of a row-partitioned level (amg_dist.cpp): which of the rank's rows every neighbour reads,
// and where the rows it reads from them land (behind the padded owned rows, peer after peer).  Level 0 copies the plan's.
struct LevelHalo {
    int32_t n_pad = 0, n_ghost = 0;
    std::vector<HaloPeer> peers;
    std::vector<int32_t> send_offsets; // per peer, in nodes
    int32_t total_send = 0;
    DevBuf<int32_t> send_nodes;        // peer after peer
    DevBuf<double> sendbuf;            // total_send x the widest row sent so far
    int sendbuf_width = 0;
};

// the clusters of a level (amg_patch.hpp) in HBM, and on the host what the aggregation and the inspection export need
struct AmgPatches {
    int32_t n_clusters = 0, n_members = 0, fell_back = 0;
    int64_t edges = 0;          // rigid edges found
    double tau = 0.0;
    int max_nodes = 0;
    bool glue = true;           // the clusters are glued into one node each before the aggregation (off: the second attempt of a
                                // setup whose glued aggregates were seen by more rows than an operator row can hold)
    DevBuf<int32_t> ptr, nodes, cluster_of;
    DevBuf<int64_t> moff;
    DevBuf<double> M;
    std::vector<int32_t> label; // per node: its cluster or -1
    // ... and the same without the clusters that hold a node another rank reads (row-partitioned levels): those smooth -- a purely
    // local matter -- but neither widen their members' rows of P nor are glued (amg_solve.cpp amg_build_patches)
    std::vector<int32_t> label_p;
    DevBuf<uint8_t> in_p;
    std::vector<int32_t> h_ptr, h_nodes;
    PatchView view() const
    {
        PatchView v;
        v.n_clusters = n_clusters;
        v.n_members = n_members;
        v.ptr = ptr.p;
        v.nodes = nodes.p;
        v.moff = moff.p;
        v.M = M.p;
        v.cluster_of = cluster_of.p;
        v.in_p = in_p.p;
        return v;
    }
};

struct AmgLevel {
    int32_t n = 0, n_pad = 0; // nodes (6 dofs each) / padded to whole slices (row-partitioned levels: the rank's own rows)
    // row-partitioned levels (contexts with a communicator, amg_dist.cpp): the rank holds the rows [part[rank], part[rank+1])
    // of the level's global numbering -- aggregates are numbered rank by rank -- and the ghost nodes its blocks reach
    bool dist = false;
    int32_t n_ghost = 0, n_global = 0;
    std::vector<int32_t> part;
    std::vector<int32_t> part_begin, part_end; // the same as two lists of the ranks' first / one-past-last rows (comm_gather_rows)
    std::shared_ptr<LevelHalo> halo;
    // ... and, for levels >= 1, the slices of the level operator ordered for the overlap of its products with their halo
    // exchange: the n_interior slices that read owned columns only first (level 0: the plan's spmv_order)
    DevBuf<int32_t> order;
    int32_t n_interior = -1;
    DevBuf<double> bown;      // last row-partitioned level: the rank's rows of the restricted residual, all-gathered into the
                              // replicated level below
    DevBuf<double> ksums;     // three words for the all-reduce of a K-cycle coefficient step
    int64_t nnzb = 0;         // blocks of the level matrix
    AmgOperator A;            // levels >= 1 (level 0 is the context's K)
    bool A_on_device = false; // the level matrix was computed in HBM (amg_device_setup.cpp), nothing to upload
    HostEllPattern pattern;   // ... and then this is the host copy of its pattern (for a further step on the device)
    AmgOperator P, R;         // to / from the next coarser level (absent on the coarsest)
    DevBuf<float> A32;        // single-precision copy of the level operator's values for the smoothing products
                              // (FEMSHELL_AMG_SMOOTH_F32; level 0: of K)
    DevBuf<float> minv32, P32, R32; // ... of the block-Jacobi inverse the smoothers apply, and of the transfer operators
    DeviceMatrix smooth_dm{}; // the level operator as the smoothers see it (vals32 / minv32 set where the copies exist)
    bool smooth_ready = false;
    DevBuf<double> minv;      // block-Jacobi inverse of A (levels >= 1)
    double lam = 0.0;         // upper bound of the spectrum of D^-1 A used by the smoother
    double inv_theta = 0.0;
    std::vector<double> cheb_a, cheb_c; // d = a d + c D^-1 r of Chebyshev steps 1 .. degree-1
    DevBuf<double> b, x;      // right-hand side and correction (levels >= 1; level 0 works on the CG's r and z)
    DevBuf<double> r, d, q;   // residual, Chebyshev direction, operator product
    DevBuf<double> c1, v1, r2, c2, v2; // K cycle (levels >= 1)
    DevBuf<KcycScalars> ks;
    DevBuf<double> kscratch;
    // host copies for the inspection exports (small problems only)
    Bsr hA, hP;
    std::vector<int32_t> agg;
    std::shared_ptr<AmgPatches> patches; // clusters of rigidly coupled nodes and their smoother blocks (level 0; null: none)
};

// timings of the first coarsening step on the device (amg_device_setup.cpp); zero when the host path ran
struct AmgSetupStats {
    double prolongator_ms = 0, ap_ms = 0, restriction_ms = 0, galerkin_ms = 0;
    double galerkin_useful_flops = 0, galerkin_mfma_flops_issued = 0;
    int galerkin_mfma = 0;
    // coarsening steps by where their patterns were built: in HBM / on the host after the device gave up / on the host by rule
    int symbolic_device = 0, symbolic_fallback = 0, symbolic_host = 0;
};

// Level 0 of a row-partitioned context (see amg_solve.cpp): work vectors in the rank's numbering (x0, d0 with ghost
// space: they are inputs of the halo product) and two fine-level vectors in global numbering for the transfer operators
struct AmgDist {
    DevBuf<double> x0;          // the cycle's iterate on level 0: an input of the halo product, so it carries ghost space (the CG's z does not)
    int dist_levels = 0;        // levels 0 .. dist_levels-1 are row-partitioned, the rest is replicated on every rank
    double hierarchy_bytes_partitioned = 0.0, hierarchy_bytes_replicated = 0.0; // HBM of the rank's operators (femshell_amg_memory)
};

// the dense inverse of a large coarsest operator, computed on the matrix cores (amg_dense.hip); n = 0: the host path ran
struct AmgDenseStats {
    int n = 0, dropped = 0;
    double ms = 0.0, mfma_flops = 0.0, useful_flops = 0.0, bytes = 0.0;
    double f32_defect = -1.0; // ||A (inv32 v) - v|| / ||v|| of the single-precision copy (-1: not taken); above 0.05 the FP64 one is kept
};

struct Amg {
    femshell_pc_options opt{};
    AmgSetupStats stats;
    AmgDenseStats dense;
    std::vector<std::unique_ptr<AmgLevel>> levels;
    DevBuf<double> coarse_inv; // dense inverse of the coarsest operator
    DevBuf<float> coarse_inv32; // ... in single precision (FEMSHELL_AMG_DENSE_F32=1), device path only
    int64_t coarse_lda = 0;    // row stride of the device-computed inverse (0: host path, rows of 6 n doubles)
    bool valid = false;
    double setup_seconds = 0.0;
    std::shared_ptr<AmgDist> dist; // row-partitioned contexts only
};

// K itself (level 0) is the coarsest level -- a dense inverse and nothing else -- only up to this many nodes (amg_solve.cpp)
constexpr int32_t kDirectNodes = 200;
void amg_default_options(femshell_pc_options *o);
// symmetric storage of a coarse level operator of n_nodes nodes (large levels only, see amg_solve.cpp)
bool coarse_symmetric_storage(int32_t n_nodes);
// in-lists of a symmetric-storage operator into HBM and into op.dm (the slot arrays of op are in place already)
int attach_in_lists(AmgOperator &op, const SlicedEllSym &S, int64_t total_slots, hipStream_t st);
// ... the same from in-lists that were built in HBM (amg_symbolic.hip): the buffers become the operator's own
int attach_in_lists_device(AmgOperator &op, DevBuf<int32_t> &in_width, DevBuf<int64_t> &in_base, DevBuf<int32_t> &in_slots, DevBuf<int32_t> &in_rows,
                           int32_t max_in_width, int64_t total_slots, hipStream_t st);
int amg_setup(femshell_ctx *c);
// contexts with a communicator (amg_dist.cpp): levels 0 .. d-1 row-partitioned like K -- aggregates never span ranks, the
// rows of Q, P and A P of the nodes along the cuts are exchanged once so that every rank computes its rows of the Galerkin
// operator itself -- and the first level of at most dist_min nodes all-gathered and replicated, with everything below it
int amg_setup_dist(femshell_ctx *c);
// the replicated part of a hierarchy from level first_level on (A: that level's operator as a host matrix, B: its near-null
// space; empty A: the level is in HBM already with its pattern, Bdev its near-null space): coarsening steps down to the
// coarsest level, dense inverse, Chebyshev coefficients of all levels
int amg_finish_hierarchy(femshell_ctx *c, Bsr &A, std::vector<double> &B, DevBuf<double> &Bdev, int first_level);
int alloc_level_vectors(AmgLevel &L, bool top, bool kcycle, hipStream_t st);
// ghost entries of vec (width doubles per node, ghosts behind the n_pad owned rows) from their owners
int level_halo_exchange(femshell_ctx *c, LevelHalo &H, double *vec, int width, hipStream_t st);
// nodes above which a coarse level stays row-partitioned (FEMSHELL_AMG_DIST_MIN, default 60000)
int32_t amg_dist_min();
// host copies of the level operators for femshell_amg_export: problems of up to 300,000 blocks (40k triangles: the tests), or
// FEMSHELL_AMG_KEEP_HOST=1 (up to 2,000,000 blocks).  Until round 4 every problem of up to 2,000,000 blocks paid for them: a
// quarter of the multigrid setup of the 250k-triangle roof went into downloads nobody read.
bool amg_keep_host(int64_t nnz_blocks);
// does the hierarchy keep single-precision copies (level operators for the smoothers, coarsest inverse)?
bool amg_uses_single_precision(const Amg &H);
// the smoother's upper bound of the spectrum of D^-1 A: safety factor x the estimate of a power iteration of that many steps
// (FEMSHELL_AMG_LAMBDA_SAFETY, FEMSHELL_AMG_POWER_ITS; read per setup)
double amg_lambda_safety();
int amg_power_iterations();
// z = M(r): one multigrid cycle on the context's stream (all launches are no-ops once gate->done != 0)
// (pre_started: the caller's kernel took the first step of the pre-smoothing on level 0 already -- k_pcg_update_start: the
//  direction in the level's d vector, the iterate = that direction in amg_apply_iterate(c, z))
int amg_apply(femshell_ctx *c, const double *r, double *z, const CgScalars *gate, bool pre_started = false);
// the vector the cycle on level 0 iterates in: z itself, or -- row-partitioned contexts -- the work vector with ghost space
double *amg_apply_iterate(femshell_ctx *c, double *z);
// *true_rr_out: ||b - K x||^2 of the returned iterate when the residual replacement computed it, else -1
// *rec_rr_out: recurrence ||r||^2 the stopping rule saw last
// (x0: the iterate to start from -- owned rows, n_pad * 6 doubles in HBM -- or nullptr for zero)
int cg_amg(femshell_ctx *c, const CgVectors &v, double rtol, int32_t max_it, double *true_rr_out, double *rec_rr_out, const double *x0 = nullptr);
// algorithmic HBM bytes of one cycle as it is built (single-precision copies, increments, the real product counts), and per level
double amg_cycle_bytes(const femshell_ctx *c, std::vector<double> *per_level = nullptr);
// K of a single-rank context as host BSR with ascending columns (api.cpp)
int download_matrix(femshell_ctx *c, Bsr *A, int32_t *col_out = nullptr, double *val_out = nullptr);
// Dense inverse of the coarsest operator on the device (amg_dense.hip): symmetric block sweeps on v_mfma_f64_16x16x4_f64.
// Exactly one of inv64 / inv32 is filled (rows of *lda entries); FEMSHELL_ERR_BREAKDOWN for an operator that is not
// positive semi-definite (same pivot rules as the host's dense_inverse)
// Asks, once per context, whether c->stream and c->aux_stream run side by side (the look-ahead of the dense inverse needs that; a
// process with many streams can have two of them on one hardware queue): c->aux_streams_side_by_side = 1 / -1.
int amg_dense_probe_streams(femshell_ctx *c);
int amg_dense_inverse_device(femshell_ctx *c, const Bsr &A, bool single_precision, DevBuf<double> *inv64, DevBuf<float> *inv32,
                             int64_t *lda_out, AmgDenseStats *stats);
// y = Ainv b for such an inverse (one of A64 / A32 non-null); rows [n, n_pad6) of y are set to zero
void launch_dense_gemv_big(const double *A64, const float *A32, int64_t lda, const double *b, double *y, int32_t n, int32_t n_pad6,
                           const CgScalars *gate, hipStream_t st);
// first coarsening step with the numerics on the device (amg_device_setup.cpp)
// (A: the level operator in HBM, block-Jacobi inverse valid; pat: host copy of its pattern; want_host(coarse nodes): bring
//  the coarse operator back as a host matrix -- needed when the next step runs on the host or the level is the coarsest)
// (B: the level's near-null space in HBM -- generated from the mesh on level 0, the previous step's Bc_dev below; the
//  tentative prolongator is factorised there.  lam_of: hands over the spectral bound of the level when the prolongator is
//  smoothed -- its power iteration runs on the device beside this function's host work.  Bc_dev: the coarse level's near-null space in HBM; Bc_out: its host copy,
//  filled under the same condition as Ac_host.  before_qr: called when the tentative prolongator is about to be enqueued -- the
//  last moment for whatever B points at to be on its way into HBM on the context's stream)
int amg_device_coarsen(femshell_ctx *c, const DeviceMatrix &A, const HostEllPattern &pat, AmgLevel &L, AmgLevel &next,
                       const NearNullSrc &B, const std::function<int(double *)> &lam_of, bool keep_host, const std::function<bool(int32_t)> &want_host,
                       Bsr *Ac_host, std::vector<double> *Bc_out, DevBuf<double> *Bc_dev,
                       const std::function<void(const char *)> &lap, const std::function<int()> &before_qr = nullptr);
// the pattern of the context's K (level 0) from the plan (light: the slice arrays only, the slot arrays stay the plan's -- see
// HostEllPattern::borrowed)
void pattern_of_plan(const Plan &p, HostEllPattern *out, bool light = false);

// clusters of rigidly coupled nodes of a level whose operator is in HBM, and their smoother blocks (amg_solve.cpp; amg_patch.hpp).
// collective: the ranks of a row partition decide together whether the mesh needs them.  L.patches stays null when it does not.
// excluded (optional, one flag per own node): nodes no cluster may take -- those another rank reads.
int amg_build_patches(femshell_ctx *c, const DeviceMatrix &A, AmgLevel &L, bool collective, const std::vector<uint8_t> *excluded = nullptr);

} // namespace femshell
