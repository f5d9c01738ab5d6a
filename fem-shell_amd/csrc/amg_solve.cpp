// amg_solve.cpp -- device side of the multigrid preconditioner (amg.hpp): the hierarchy in HBM, the V / K cycle
// and the flexible PCG around it.  Replaces what `-pc_type gamg`-style options select inside PETSc's KSPSolve
// behind equation_systems.solve() (fem-shell.cpp:138, doc/implementation.tex:68-72).
#include "amg_device.hpp"
#include "amg_pattern.hpp"
#include <thread>
#include "trace.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace femshell {

namespace {

bool has_lowp_copy(const AmgLevel &L) { return L.A32.p != nullptr; }

// FEMSHELL_AMG_VEC_F32: what a smoothing product on the single-precision copy of a level operator (symmetric storage) keeps
// in single precision besides the values -- 2 (default): its results (direct part, transposed products) and its input, the
// Chebyshev direction; 1: the results only; 0: nothing.  Residuals and iterates stay FP64.  Read per setup.
int smooth_vectors_f32()
{
    const char *e = getenv("FEMSHELL_AMG_VEC_F32");
    const int v = e ? atoi(e) : 2;
    return v < 0 ? 0 : (v > 2 ? 2 : v);
}

bool setup_verbose_flag()
{
    static const bool verbose = getenv("FEMSHELL_AMG_VERBOSE") && atoi(getenv("FEMSHELL_AMG_VERBOSE")) != 0;
    return verbose;
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int upload_operator(AmgOperator &op, const SlicedEll &S, int32_t n_cols_pad, int64_t nnzb, hipStream_t st)
{
    FS_HIP(op.slice_width.upload(S.slice_width, st));
    FS_HIP(op.slice_base.upload(S.slice_base, st));
    FS_HIP(op.cols.upload(S.cols, st));
    FS_HIP(op.vals.upload(S.vals, st));
    FS_HIP(hipStreamSynchronize(st));
    op.nnzb = nnzb;
    op.n_cols_pad = n_cols_pad;
    op.dm = DeviceMatrix();
    op.dm.n_own = S.n_rows;
    op.dm.n_pad = S.n_pad;
    op.dm.n_slices = S.n_slices;
    op.dm.slice_width = op.slice_width.p;
    op.dm.slice_base = op.slice_base.p;
    op.dm.cols = op.cols.p;
    op.dm.vals = op.vals.p;
    op.dm.max_slice_width = S.max_width;
    return FEMSHELL_OK;
}

} // namespace

int attach_in_lists(AmgOperator &op, const SlicedEllSym &S, int64_t total_slots, hipStream_t st)
{
    FS_HIP(op.in_width.upload(S.in_width, st));
    FS_HIP(op.in_base.upload(S.in_base, st));
    FS_HIP(op.in_slots.upload(S.in_slots, st));
    FS_HIP(op.in_rows.upload(S.in_rows, st));
    FS_HIP(op.tbuf.alloc((size_t)total_slots * 6));
    FS_HIP(op.tbuf.zero(st));
    FS_HIP(hipStreamSynchronize(st));
    op.dm.symmetric = 1;
    op.dm.max_in_width = S.max_in_width;
    op.dm.in_width = op.in_width.p;
    op.dm.in_base = op.in_base.p;
    op.dm.in_slots = op.in_slots.p;
    op.dm.in_rows = op.in_rows.p;
    op.dm.gat_slots = op.in_slots.p; // (level operators: every transposed product through tbuf)
    op.dm.tbuf = op.tbuf.p;
    return FEMSHELL_OK;
}

int attach_in_lists_device(AmgOperator &op, DevBuf<int32_t> &in_width, DevBuf<int64_t> &in_base, DevBuf<int32_t> &in_slots, DevBuf<int32_t> &in_rows,
                           int32_t max_in_width, int64_t total_slots, hipStream_t st)
{
    std::swap(op.in_width.p, in_width.p);
    std::swap(op.in_width.n, in_width.n);
    std::swap(op.in_base.p, in_base.p);
    std::swap(op.in_base.n, in_base.n);
    std::swap(op.in_slots.p, in_slots.p);
    std::swap(op.in_slots.n, in_slots.n);
    std::swap(op.in_rows.p, in_rows.p);
    std::swap(op.in_rows.n, in_rows.n);
    FS_HIP(op.tbuf.alloc((size_t)total_slots * 6));
    FS_HIP(op.tbuf.zero(st));
    FS_HIP(hipStreamSynchronize(st));
    op.dm.symmetric = 1;
    op.dm.max_in_width = max_in_width;
    op.dm.in_width = op.in_width.p;
    op.dm.in_base = op.in_base.p;
    op.dm.in_slots = op.in_slots.p;
    op.dm.in_rows = op.in_rows.p;
    op.dm.gat_slots = op.in_slots.p; // (level operators: every transposed product through tbuf)
    op.dm.tbuf = op.tbuf.p;
    return FEMSHELL_OK;
}

namespace {

// lambda_max(D^-1 A) of a level by power iteration on the device (x <- D^-1 A x, ratio of consecutive norms).  In two halves:
// the launches, and -- when lambda is needed -- the one synchronisation that brings the last two norms back; a coarsening
// step on the device does its host-side graph work (aggregation, patterns: 0.16 s on the 4M-triangle meshes) in between
// while the GPU iterates.
struct PowerIteration {
    DevBuf<double> part, sums;
    int G = 0, iterations = 0;
    hipStream_t st = nullptr; // the stream its launches went to
};

// FEMSHELL_AMG_PATCH_TAU (default 0.8; 0: no patch smoother) and FEMSHELL_AMG_PATCH_MAX (nodes per cluster, default 8): amg_patch.hpp
double patch_tau()
{
    const char *e = getenv("FEMSHELL_AMG_PATCH_TAU"); // (read per setup: the tests switch it inside one process)
    return e && *e ? atof(e) : 0.8;
}
int patch_max_nodes()
{
    const char *e = getenv("FEMSHELL_AMG_PATCH_MAX");
    const int m = e && *e ? atoi(e) : 8;
    return m < 2 ? 2 : (m > kPatchMaxNodes ? kPatchMaxNodes : m);
}

// When is a mesh one of poor element quality?  Edges above tau = 0.8 are ordinary on stretched structured elements (a third of the
// edges of the pinched cylinder's 3 : 1 cells reach 0.836, the coupled flap's 5 : 1 cells 0.965) where the point-block method works;
// what it does not cope with are NEARLY COINCIDENT nodes, sigma > 0.98: 2.6 % of the edges of the random-point shells, 0.13 % of a
// jittered-grid Delaunay shell (which converges in 90 iterations without cluster blocks), none of any structured mesh.  The
// clusters are built when more than FEMSHELL_AMG_PATCH_TRIGGER (default 0.01) of the pairs exceed 0.98; 0: whenever an edge exceeds tau.
double patch_trigger()
{
    const char *e = getenv("FEMSHELL_AMG_PATCH_TRIGGER");
    return e && *e ? atof(e) : 0.01;
}
constexpr double kPatchTriggerSigma = 0.98;

} // namespace

// The clusters of rigidly coupled nodes of a level whose operator is in HBM (block-Jacobi inverse valid) and their smoother
// blocks M_c: detection on the device (k_patch_sigma), the union of the edges and the 36 x 36 inverses on the host.  A level
// without a rigid edge -- every structured mesh -- costs one kernel over its blocks and a four-byte copy, and keeps L.patches null.
int amg_build_patches(femshell_ctx *c, const DeviceMatrix &A, AmgLevel &L, bool collective, const std::vector<uint8_t> *excluded)
{
    L.patches.reset();
    const double tau = patch_tau();
    if (!(tau > 0.0) || A.n_own < 2 || A.minv == nullptr) return FEMSHELL_OK;
    hipStream_t st = c->stream;
    DevBuf<unsigned int> counter;
    DevBuf<PatchEdge> edges;
    FS_HIP(counter.alloc(4));
    unsigned int cap = (unsigned int)std::max<int64_t>(65536, A.n_own), found = 0;
    unsigned int counts[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        FS_HIP(edges.alloc(cap));
        FS_HIP(counter.zero(st));
        launch_patch_sigma(A, tau, kPatchTriggerSigma, edges.p, counter.p, cap, st);
        FS_HIP(hipGetLastError());
        FS_HIP(hipMemcpyAsync(counts, counter.p, sizeof counts, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        found = counts[0];
        if (found <= cap) break;
        cap = found; // (more rigid edges than nodes: once more with room for all of them)
    }
    // the decision belongs to the whole mesh: the ranks of a row partition add their counts up
    double high = counts[1], pairs = counts[2];
    if (collective && c->comm.active()) {
        DevBuf<double> sums;
        FS_HIP(sums.alloc(2));
        const double mine[2] = {high, pairs};
        FS_HIP(hipMemcpyAsync(sums.p, mine, sizeof mine, hipMemcpyHostToDevice, st));
        std::string e;
        if (!comm_allreduce_sum(c->comm, sums.p, 2, st, &e)) return set_err(FEMSHELL_ERR_COMM, e);
        double all[2] = {0.0, 0.0};
        FS_HIP(hipMemcpyAsync(all, sums.p, sizeof all, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        high = all[0];
        pairs = all[1];
    }
    if (setup_verbose_flag())
        fprintf(stderr, "[femshell amg setup] patch smoother: %u of %.0f pairs above sigma %.2f (%.0f above %.2f: %s)\n", found, pairs, tau, high,
                kPatchTriggerSigma, high > patch_trigger() * pairs || (patch_trigger() <= 0.0 && found > 0) ? "a mesh of poor element quality" : "no clusters");
    if (patch_trigger() > 0.0 ? !(high > patch_trigger() * pairs) : found == 0) return FEMSHELL_OK;
    if (found == 0) return FEMSHELL_OK;
    std::vector<PatchEdge> h((size_t)found);
    FS_HIP(hipMemcpyAsync(h.data(), edges.p, h.size() * sizeof(PatchEdge), hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    auto P = std::make_shared<AmgPatches>();
    P->tau = tau;
    P->max_nodes = patch_max_nodes();
    P->edges = found;
    P->n_clusters = patch_clusters(A.n_own, std::move(h), P->max_nodes, &P->label, &P->h_ptr, &P->h_nodes);
    if (P->n_clusters == 0) return FEMSHELL_OK;
    P->n_members = (int32_t)P->h_nodes.size();
    // Row-partitioned levels: a cluster that holds a node another rank reads (the nodes of the halo's send lists) smooths like the
    // others -- z_c += M_c r_c is local -- but stays out of the smoothing of P and of the gluing.  A cluster couples the row of P of
    // each of its members to the aggregates ALL its members see; were a member adjacent to another rank's nodes, a cluster-mate one
    // node further from the cut would get a block in a column of that rank -- a fine row the rank does not hold among its ghosts,
    // whose share of P^T A P it could not add (measured: "coarsest operator is not positive definite").
    P->label_p = P->label;
    std::vector<uint8_t> in_p; // (source of an upload: lives until the synchronisations below)
    if (excluded != nullptr && !excluded->empty()) {
        in_p.assign((size_t)P->n_clusters, 1);
        for (int32_t i = 0; i < A.n_own; i++)
            if (P->label[(size_t)i] >= 0 && (*excluded)[(size_t)i]) in_p[(size_t)P->label[(size_t)i]] = 0;
        for (int32_t i = 0; i < A.n_own; i++)
            if (P->label[(size_t)i] >= 0 && !in_p[(size_t)P->label[(size_t)i]]) P->label_p[(size_t)i] = -1;
        FS_HIP(P->in_p.upload(in_p, st));
    }
    std::vector<int64_t> moff((size_t)P->n_clusters);
    std::vector<int32_t> cluster_of((size_t)P->n_members);
    int64_t total = 0;
    for (int32_t k = 0; k < P->n_clusters; k++) {
        const int64_t m = P->h_ptr[(size_t)k + 1] - P->h_ptr[(size_t)k];
        moff[(size_t)k] = total;
        total += 36 * m * m;
        for (int32_t t = P->h_ptr[(size_t)k]; t < P->h_ptr[(size_t)k + 1]; t++) cluster_of[(size_t)t] = k;
    }
    FS_HIP(P->ptr.upload(P->h_ptr, st));
    FS_HIP(P->nodes.upload(P->h_nodes, st));
    FS_HIP(P->moff.upload(moff, st));
    FS_HIP(P->cluster_of.upload(cluster_of, st));
    FS_HIP(P->M.alloc((size_t)total));
    DevBuf<double> dinv;
    FS_HIP(dinv.alloc((size_t)P->n_members * 36));
    launch_patch_gather(A, P->view(), P->M.p, dinv.p, st);
    FS_HIP(hipGetLastError());
    std::vector<double> hB((size_t)total), hD((size_t)P->n_members * 36);
    FS_HIP(hipMemcpyAsync(hB.data(), P->M.p, hB.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    FS_HIP(hipMemcpyAsync(hD.data(), dinv.p, hD.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    P->fell_back = patch_matrices(P->h_ptr, moff, hD.data(), hB.data());
    FS_HIP(hipMemcpyAsync(P->M.p, hB.data(), hB.size() * sizeof(double), hipMemcpyHostToDevice, st));
    FS_HIP(hipStreamSynchronize(st)); // (hB goes out of scope)
    if (setup_verbose_flag())
        fprintf(stderr, "[femshell amg setup] patch smoother: %lld rigid edges (sigma > %.2f), %d clusters of %d nodes (at most %d each), %d not positive definite\n",
                (long long)P->edges, tau, P->n_clusters, P->n_members, P->max_nodes, P->fell_back);
    L.patches = P;
    return FEMSHELL_OK;
}

namespace {

// (beside: on the context's second stream, behind everything the first one holds at this moment -- the symbolic kernels of the
//  coarsening step, which the first stream gets next, then run beside the products instead of behind them: 30 products of K are
//  22 ms at 4M triangles, the patterns 10)
int power_iteration_start(femshell_ctx *c, AmgLevel &L, const DeviceMatrix &A, int iterations, PowerIteration *pw, bool beside = false)
{
    hipStream_t st = c->stream;
    if (beside && c->aux_stream != nullptr) {
        hipEvent_t ev;
        FS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        FS_HIP(hipEventRecord(ev, c->stream));
        FS_HIP(hipStreamWaitEvent(c->aux_stream, ev, 0));
        FS_HIP(hipEventDestroy(ev));
        st = c->aux_stream;
    }
    pw->st = st;
    pw->G = slice_grid(A);
    // lambda is the ratio of the last two norms: only those come back to the host (one synchronisation instead of one per
    // step; without normalisation the iterate grows like lambda^k, lambda ~ 2, which 30 steps of FP64 take easily)
    FS_HIP(pw->part.alloc(2 * (size_t)pw->G));
    double *x = L.d.p, *z = L.r.p;
    launch_fill_hash(x, 6ll * L.n, 6ll * L.n_pad, st);
    if (iterations < 2) iterations = 2;
    pw->iterations = iterations;
    for (int it = 0; it < iterations; it++) {
        launch_spmv(A, x, L.q.p, nullptr, nullptr, st);
        launch_minv_apply_norm(A, L.q.p, z, pw->part.p + (size_t)(it & 1) * pw->G, st);
        if (L.patches) { // the level's smoother applies the cluster blocks: lambda_max of THAT operator
            launch_patch_correct(L.patches->view(), L.q.p, 1.0, z, false, nullptr, nullptr, st);
            launch_sqnorm_partials(z, 6ll * L.n_pad, pw->part.p + (size_t)(it & 1) * pw->G, pw->G, st);
        }
        std::swap(x, z);
    }
    FS_HIP(hipGetLastError());
    return FEMSHELL_OK;
}

int power_iteration_finish(femshell_ctx *c, PowerIteration &pw, double *lam_out)
{
    hipStream_t st = pw.st != nullptr ? pw.st : c->stream;
    const int G = pw.G, iterations = pw.iterations;
    // the two norms are sums over the G partials in index order, taken on the device (launch_sums_in_order: the host's loop of
    // rounds 3-5, same bits): 16 bytes come back instead of 16 G
    const size_t last = (size_t)((iterations - 1) & 1) * G, prev = (size_t)((iterations - 2) & 1) * G;
    FS_HIP(pw.sums.alloc(2));
    launch_sums_in_order(pw.part.p, G, pw.sums.p, st); // (sums[0] over part[0, G), sums[1] over part[G, 2 G))
    FS_HIP(hipGetLastError());
    double h[2] = {0.0, 0.0};
    FS_HIP(hipMemcpyAsync(h, pw.sums.p, sizeof h, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    const double s_last = h[last / (size_t)std::max(G, 1)], s_prev = h[prev / (size_t)std::max(G, 1)];
    const double n_last = std::sqrt(s_last), n_prev = std::sqrt(s_prev);
    if (!(n_prev > 0.0) || !std::isfinite(n_last) || !(n_last > 0.0))
        return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: power iteration broke down");
    *lam_out = n_last / n_prev;
    return FEMSHELL_OK;
}

} // namespace

// (row-partitioned levels: every vector carries ghost space -- any of them may be the input of a halo product)
int alloc_level_vectors(AmgLevel &L, bool top, bool kcycle, hipStream_t st)
{
    const size_t n6 = (size_t)(L.n_pad + (L.dist ? L.n_ghost : 0)) * 6;
    if (!top) {
        FS_HIP(L.b.alloc(n6));
        FS_HIP(L.x.alloc(n6));
        FS_HIP(L.b.zero(st));
        FS_HIP(L.x.zero(st));
    }
    FS_HIP(L.r.alloc(n6));
    FS_HIP(L.d.alloc(n6));
    FS_HIP(L.q.alloc(n6));
    FS_HIP(L.r.zero(st));
    FS_HIP(L.d.zero(st));
    FS_HIP(L.q.zero(st));
    if (!top && kcycle) {
        FS_HIP(L.c1.alloc(n6));
        FS_HIP(L.v1.alloc(n6));
        FS_HIP(L.r2.alloc(n6));
        FS_HIP(L.c2.alloc(n6));
        FS_HIP(L.v2.alloc(n6));
        FS_HIP(L.c1.zero(st));
        FS_HIP(L.v1.zero(st));
        FS_HIP(L.r2.zero(st));
        FS_HIP(L.c2.zero(st));
        FS_HIP(L.v2.zero(st));
        FS_HIP(L.ks.alloc(1));
        FS_HIP(L.ks.zero(st));
        FS_HIP(L.kscratch.alloc(3 * 128));
        if (L.dist) {
            FS_HIP(L.ksums.alloc(8)); // (three sums, a spare; word 4: the ticket of the last-workgroup reduction, amg_kernels.hip)
            FS_HIP(L.ksums.zero(st));
        }
    }
    return FEMSHELL_OK;
}

double amg_lambda_safety()
{
    const char *e = getenv("FEMSHELL_AMG_LAMBDA_SAFETY");
    const double v = e ? atof(e) : 1.1;
    return v >= 1.0 && v <= 4.0 ? v : 1.1;
}

int amg_power_iterations()
{
    const char *e = getenv("FEMSHELL_AMG_POWER_ITS");
    const int v = e ? atoi(e) : 30;
    return v >= 2 && v <= 10000 ? v : 30;
}

bool coarse_symmetric_storage(int32_t n_nodes)
{
    // levels of at least 100,000 nodes (FEMSHELL_AMG_COARSE_SYM overrides; 1 = all levels, 0 = none) are stored like K,
    // diagonal + upper blocks: their products are HBM-bound (4M-triangle panel, level 1 of 222k nodes: 1.63 s against
    // 1.71 s per solve); the smaller levels are launch- and latency-bound, where the two-phase product loses (all
    // levels symmetric: 1.80 s)
    const char *e = getenv("FEMSHELL_AMG_COARSE_SYM"); // (read per call: the tests switch it inside one process)
    const long min_nodes = e ? atol(e) : 100000l;
    return min_nodes > 0 && n_nodes >= min_nodes && default_symmetric_storage();
}

void amg_default_options(femshell_pc_options *o)
{
    std::memset(o, 0, sizeof *o);
    o->type = FEMSHELL_PC_AMG;
    o->cycle = FEMSHELL_CYCLE_K;
    o->smoother_degree = 3; // tools/lab/amg_degree_probe.py, 4M triangles, rtol 1e-10 with the refinement pass: degrees 3 / 4
    o->coarse_degree = 4;   // against 2 / 4: panel 104 against 132 iterations, 0.98 against 1.04 s; cylinder 97 / 119, 0.91 /
                            // 0.94 s; roof 99 / 112, 0.087 / 0.089 s (3 / 3: 1.04 s on the panel; V cycles: twice the time)
    o->coarsest_nodes = 1400; // the K cycle's last levels cost launches, not bytes: end with an exact solve early (amg_dense.hip)
    o->max_levels = 12;
    o->refine_passes = 1;
    o->eig_ratio = 30.0;
}

const DeviceMatrix &amg_level_matrix(const femshell_ctx *c, int l) { return l == 0 ? c->dm : c->amg->levels[l]->A.dm; }

// Builds the hierarchy for the matrix currently in HBM.  Host: aggregation, prolongators, Galerkin products
// (amg_setup.cpp); device: lambda_max of every level, block-Jacobi inverses of the coarse operators.
namespace {

// which levels are coarsened where, and where the hierarchy ends
struct SetupRules {
    femshell_pc_options opt{};
    bool host_only = false;
    int32_t device_min = 5000;
    // Levels of more than device_min nodes are coarsened with the numerics on the device (amg_device_setup.cpp): their
    // operator is in HBM already and only its pattern is needed on the host.  The coarse operator of such a step comes
    // back as a host matrix only when the next step runs on the host (a small level, the coarsest one, the last allowed).
    bool device_step(int l, int32_t n_nodes) const
    {
        return !host_only && n_nodes > std::max(device_min, opt.coarsest_nodes) && l + 2 < opt.max_levels;
    }
    // The hierarchy ends at the first level of at most coarsest_nodes nodes -- but K itself is only "solved" by a dense
    // inverse when it is really small (kDirectNodes): the explicit FP64 inverse of a thin-shell K of a thousand nodes is not
    // a positive definite operator any more (coupled flap of 2,000 triangles, 6666 dofs: one pivot dropped, the CG residual
    // grows to 1e14), while the Galerkin operators below it are harmless (7386 dofs at 4M triangles).
    bool is_coarsest(int l, int32_t n_nodes) const
    {
        const int32_t limit = l == 0 ? std::min(opt.coarsest_nodes, kDirectNodes) : opt.coarsest_nodes;
        return n_nodes <= limit || l + 1 >= opt.max_levels;
    }
};

SetupRules setup_rules(const femshell_pc_options &opt)
{
    SetupRules r;
    r.opt = opt;
    // the first coarsening step runs its numerics on the device unless FEMSHELL_AMG_SETUP=host (amg_device_setup.cpp)
    r.host_only = getenv("FEMSHELL_AMG_SETUP") && std::string(getenv("FEMSHELL_AMG_SETUP")) == "host";
    const char *dmin_env = getenv("FEMSHELL_AMG_DEVICE_MIN"); // nodes; levels at or below it are coarsened on the host
    // (5000 since round 4: the 18.6k-node level of the 4M-triangle meshes on the device as well, 46 -> 32 ms for the two
    //  steps below level 0)
    r.device_min = dmin_env ? (int32_t)atol(dmin_env) : (int32_t)5000;
    return r;
}

bool setup_verbose()
{
    static const bool verbose = getenv("FEMSHELL_AMG_VERBOSE") && atoi(getenv("FEMSHELL_AMG_VERBOSE")) != 0;
    return verbose;
}

} // namespace

namespace {
// ||A (inv32 v) - v|| / ||v|| for one vector v of mixed frequencies: what single precision did to the coarsest inverse
int dense_inverse_defect(femshell_ctx *c, AmgLevel &L, const DeviceMatrix &A, Amg &H, double *rel_out)
{
    hipStream_t st = c->stream;
    const int64_t n6 = 6ll * L.n_pad;
    launch_fill_hash(L.r.p, 6ll * L.n, n6, st);
    launch_dense_gemv_big(H.coarse_inv.p, H.coarse_inv32.p, H.coarse_lda, L.r.p, L.d.p, 6 * L.n, 6 * L.n_pad, nullptr, st);
    launch_spmv(A, L.d.p, L.q.p, nullptr, nullptr, st);
    launch_sub(L.q.p, L.r.p, L.q.p, n6, st);
    DevBuf<double> scratch;
    FS_HIP(scratch.alloc(3 * 128));
    const int groups = launch_two_dots(L.q.p, L.q.p, L.r.p, L.r.p, n6, scratch.p, st);
    double hp[2 * 128];
    FS_HIP(hipMemcpyAsync(hp, scratch.p, sizeof hp, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    FS_HIP(hipGetLastError());
    double ee = 0.0, vv = 0.0;
    for (int g = 0; g < groups; g++) {
        ee += hp[g];
        vv += hp[128 + g];
    }
    *rel_out = vv > 0.0 ? std::sqrt(ee / vv) : 0.0; // (NaN: the caller's test fails and the FP64 inverse is taken)
    return FEMSHELL_OK;
}
} // namespace

bool amg_keep_host(int64_t nnz_blocks)
{
    const char *e = getenv("FEMSHELL_AMG_KEEP_HOST"); // (read per setup)
    if (e && atoi(e) != 0) return nnz_blocks <= (int64_t)2000000;
    return nnz_blocks <= (int64_t)300000;
}

bool amg_uses_single_precision(const Amg &H)
{
    if (H.coarse_inv32.p != nullptr) return true;
    for (const auto &L : H.levels)
        if (L->A32.p != nullptr) return true;
    return false;
}

int amg_setup(femshell_ctx *c)
{
    TraceRange trace("femshell multigrid setup");
    const double t0 = now_s();
    DevPool::Defer no_sync_per_free; // (buffers released during the setup join the pool at its end: context.hpp)
    hipStream_t st = c->stream;
    const femshell_pc_options opt = c->pc;
    const bool kcycle = opt.cycle == FEMSHELL_CYCLE_K;
    c->amg.reset(new Amg());
    Amg &H = *c->amg;
    H.opt = opt;
    const Plan &pl = c->plan;
    if (setup_verbose_flag()) {
        double ps[6];
        DevPool::get().stats(ps); // (clears the counters: what follows belongs to this setup)
    }

    double tl = now_s();
    auto lap = [&](const char *what, int level) {
        const double t = now_s();
        if (setup_verbose()) fprintf(stderr, "[femshell amg setup] level %d %-28s %.3f s\n", level, what, t - tl);
        tl = t;
        CommWatch::heartbeat(); // (progress of the phase the watchdog of a multi-rank context times)
    };
    const SetupRules rules = setup_rules(opt);
    const bool host_only = rules.host_only;
    const bool keep_host = amg_keep_host(pl.nnz_blocks); // inspection exports (tests)
    Bsr A;
    std::vector<double> B;  // near-null space of the current level on the host (levels coarsened on the host) ...
    DevBuf<double> Bdev;    // ... and in HBM (levels coarsened on the device, from the second one on)
    // FEMSHELL_AMG_PLAIN_RBM=1: the six plain rigid-body modes (A/B runs)
    static const bool plain = getenv("FEMSHELL_AMG_PLAIN_RBM") && atoi(getenv("FEMSHELL_AMG_PLAIN_RBM")) != 0;
    // the node normals (48 MB to bring over at 4M triangles) on a thread of their own, beside the search for
    // clusters, the copy of the pattern and the greedy passes of the aggregation: the thread computes them and -- when a coarsening
    // step on the device will read them -- copies them into HBM on a stream of its own; whoever needs them first waits for the
    // thread (normals_ready: the host array is complete and the device copy has arrived)
    RawVec<double> normals;
    DevBuf<double> d_normals;
    std::thread normals_thread;
    hipError_t normals_err = hipSuccess;
    const bool normals_to_device = !plain && !host_only && opt.max_levels > 1 && pl.n_own > kDirectNodes;
    if (normals_to_device) FS_HIP(d_normals.alloc((size_t)pl.n_own * 3));
    if (!plain)
        normals_thread = std::thread([&] {
            node_normals_plan(pl, &normals);
            if (!normals_to_device) return;
            normals_err = hipSetDevice(c->device);
            if (normals_err == hipSuccess && c->copy_stream == nullptr) normals_err = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
            hipStream_t s = c->copy_stream; // (see context.hpp)
            // (through the pinned staging buffer, region 1: `normals` is freed when the first step is through -- context.hpp stage_host)
            if (normals_err == hipSuccess && staged_upload(c, d_normals.p, normals.data(), normals.size() * sizeof(double), s, 1) != FEMSHELL_OK)
                normals_err = hipErrorUnknown;
        });
    auto normals_ready = [&] {
        if (normals_thread.joinable()) normals_thread.join();
    };
    struct JoinAtExit { // (every early return below)
        std::function<void()> f;
        ~JoinAtExit() { f(); }
    } join_at_exit{normals_ready};
    // (decided by amg_device_coarsen for the level it creates: it knows the coarse size only after the aggregation)
    auto want_host_matrix = [&](int next_level) {
        return std::function<bool(int32_t)>([&, next_level](int32_t na) { return !rules.device_step(next_level, na); });
    };
    int rc = FEMSHELL_OK;
    int first_level = 0;
    // clusters of rigidly coupled nodes (amg_patch.hpp: none on a mesh of decent element quality); a mesh that has them takes the
    // device path below whatever its size -- the host path has no cluster blocks
    std::shared_ptr<AmgPatches> patches0;
    // A mesh that is certain to take its first coarsening step on the device starts the power iteration of level 0 -- 30 products
    // of K on the second stream, 22 ms at 4M triangles, the longest thing the GPU has to do in a setup -- before anything else; the
    // search for clusters, the host's copy of the pattern, the aggregation and the symbolic kernels all run beside it.  (A mesh that
    // turns out to have clusters starts it again: lambda_max is then that of the smoother with the cluster blocks.)
    PowerIteration pw_early;
    bool early = false;
    if (!host_only && opt.max_levels > 1 && pl.n_own > opt.coarsest_nodes && pl.n_own > kDirectNodes) {
        H.levels.emplace_back(new AmgLevel());
        AmgLevel &L0 = *H.levels.back();
        L0.n = pl.n_own;
        L0.n_pad = pl.n_pad;
        L0.nnzb = pl.nnz_blocks;
        rc = alloc_level_vectors(L0, true, kcycle, st);
        if (rc) return rc;
        rc = power_iteration_start(c, L0, c->dm, amg_power_iterations(), &pw_early, true);
        if (rc) return rc;
        early = true;
    }
    if (!host_only && opt.max_levels > 1 && pl.n_own > kDirectNodes) {
        AmgLevel probe;
        probe.n = pl.n_own;
        rc = amg_build_patches(c, c->dm, probe, false);
        if (rc) return rc;
        patches0 = probe.patches;
        lap("patch smoother: clusters", 0);
    }
    if (!host_only && (pl.n_own > opt.coarsest_nodes || patches0) && opt.max_levels > 1) { // (small meshes: host algebra below)
        if (!early) {
            H.levels.emplace_back(new AmgLevel());
            AmgLevel &Lnew = *H.levels.back();
            Lnew.n = pl.n_own;
            Lnew.n_pad = pl.n_pad;
            Lnew.nnzb = pl.nnz_blocks;
            rc = alloc_level_vectors(Lnew, true, kcycle, st);
            if (rc) return rc;
        }
        AmgLevel &L0 = *H.levels[0];
        L0.patches = patches0;
        if (early && patches0) { // (the early power iteration ran without the cluster blocks)
            FS_HIP(hipStreamSynchronize(pw_early.st));
            early = false;
        }
        H.levels.emplace_back(new AmgLevel());
        AmgLevel &L1 = *H.levels.back();
        std::vector<double> Bc;
        pattern_of_plan(pl, &L0.pattern, /*light=*/true); // (amg_device_coarsen fills the slot arrays if it needs the host's lists)
        lap("pattern of K", 0);
        // the near-null space of the finest level is generated from the mesh in HBM (never stored: n x 36 doubles)
        NearNullSrc src;
        src.xyz = c->xyz.p;
        src.dmask = c->dmask.p;
        if (!plain) src.normals = d_normals.p;
        // (amg_device_coarsen calls this when the tentative prolongator is about to read them)
        auto upload_normals = [&]() -> int {
            if (plain) return (int)FEMSHELL_OK;
            normals_ready();
            FS_HIP(normals_err);
            return (int)FEMSHELL_OK;
        };
        double ctr[3];
        mesh_centre(pl.n_own, pl.xyz_local.data(), ctr);
        src.cx = ctr[0];
        src.cy = ctr[1];
        src.cz = ctr[2];
        // (with clusters: glued into aggregates; should an aggregate then be seen by more fine rows than a row of R holds, once more
        //  without the gluing, and without the cluster blocks at all if that fails too)
        for (int attempt = 0;; attempt++) {
            PowerIteration pw_now; // (its launches now, its result when the prolongator is smoothed: amg_device_coarsen asks for it)
            const bool reuse_early = attempt == 0 && early;
            PowerIteration &pw0 = reuse_early ? pw_early : pw_now;
            if (!reuse_early) {
                rc = power_iteration_start(c, L0, c->dm, amg_power_iterations(), &pw0, true);
                if (rc) return rc;
            }
            auto lam0 = [&](double *out) {
                double lam = 0.0;
                const int r2 = power_iteration_finish(c, pw0, &lam);
                if (r2) return r2;
                *out = L0.lam = amg_lambda_safety() * lam;
                return (int)FEMSHELL_OK;
            };
            rc = amg_device_coarsen(c, c->dm, L0.pattern, L0, L1, src, lam0, keep_host, want_host_matrix(1), &A, &Bc, &Bdev,
                                    [&](const char *what) { lap(what, 0); }, upload_normals);
            if (rc == FEMSHELL_ERR_UNSUPPORTED && L0.patches && attempt < 2) {
                FS_HIP(hipStreamSynchronize(pw0.st)); // (the power iteration of this attempt)
                FS_HIP(hipStreamSynchronize(st));
                if (L0.patches->glue) L0.patches->glue = false;
                else L0.patches.reset();
                if (setup_verbose_flag()) fprintf(stderr, "[femshell amg setup] patch smoother: %s\n", L0.patches ? "once more without gluing the clusters" : "given up for this mesh");
                continue;
            }
            break;
        }
        if (rc) return rc;
        L0.pattern = HostEllPattern(); // (the plan holds it)
        L1.A_on_device = true;
        if (keep_host) {
            rc = download_matrix(c, &L0.hA);
            if (rc) return rc;
        }
        B.swap(Bc);
        first_level = 1;
    } else {
        rc = download_matrix(c, &A);
        if (rc) return rc;
        normals_ready();
        rigid_body_modes(pl.n_own, pl.xyz_local.data(), c->dmask_global.data() + pl.row_begin, &B, plain ? nullptr : normals.data());
        lap("download K", 0);
    }
    normals_ready();
    RawVec<double>().swap(normals);

    rc = amg_finish_hierarchy(c, A, B, Bdev, first_level);
    if (rc) return rc;
    H.setup_seconds = now_s() - t0;
    if (setup_verbose_flag()) {
        double ps[6];
        DevPool::get().stats(ps);
        fprintf(stderr, "[femshell amg setup] device allocator during this setup (and whatever ran since the last one): %.0f hipMalloc %.1f ms, %.0f hipFree %.1f ms, "
                        "%.0f blocks to the pool behind a device synchronisation %.1f ms\n", ps[0], ps[1], ps[2], ps[3], ps[4], ps[5]);
    }
    return FEMSHELL_OK;
}

// The levels from first_level on, on one rank or replicated on every rank of a row partition (amg_dist.cpp hands over the
// all-gathered operator of the first replicated level).  A: that level's operator as a host matrix (empty: the level is in
// HBM with its pattern and Bdev is its near-null space), B: its near-null space on the host.
int amg_finish_hierarchy(femshell_ctx *c, Bsr &A, std::vector<double> &B, DevBuf<double> &Bdev, int first_level)
{
    hipStream_t st = c->stream;
    Amg &H = *c->amg;
    const femshell_pc_options opt = H.opt;
    const bool kcycle = opt.cycle == FEMSHELL_CYCLE_K;
    const Plan &pl = c->plan;
    const SetupRules rules = setup_rules(opt);
    auto device_step = [&](int l, int32_t n_nodes) { return rules.device_step(l, n_nodes); };
    auto is_coarsest = [&](int l, int32_t n_nodes) { return rules.is_coarsest(l, n_nodes); };
    auto want_host_matrix = [&](int next_level) {
        return std::function<bool(int32_t)>([&, next_level](int32_t na) { return !device_step(next_level, na); });
    };
    const bool keep_host = amg_keep_host(pl.nnz_blocks); // inspection exports (tests)
    double tl = now_s();
    auto lap = [&](const char *what, int level) {
        const double t = now_s();
        if (setup_verbose()) fprintf(stderr, "[femshell amg setup] level %d %-28s %.3f s\n", level, what, t - tl);
        tl = t;
        CommWatch::heartbeat(); // (progress of the phase the watchdog of a multi-rank context times)
    };
    int rc = FEMSHELL_OK;
    for (int l = first_level;; l++) {
        if ((int)H.levels.size() <= l) H.levels.emplace_back(new AmgLevel());
        AmgLevel &L = *H.levels[l];
        const bool have_host = A.nr > 0 || !L.A_on_device; // the level's operator as a host matrix (else: in HBM + pattern)
        if (have_host) {
            L.n = A.nr;
            L.nnzb = A.nnzb();
        } else {
            L.n = L.pattern.n;
            int64_t stored = 0;
            for (int32_t a = 0; a < L.n; a++) stored += L.pattern.count[(size_t)a];
            L.nnzb = L.pattern.symmetric ? 2 * stored - L.n : stored;
        }
        L.n_pad = (L.n + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
        if (l > 0) {
            if (!L.A_on_device) {
                // the level operators are symmetric: diagonal and upper blocks only, like K (FEMSHELL_SYMMETRIC=0: full)
                if (coarse_symmetric_storage(A.nr)) {
                    SlicedEllSym S;
                    pack_sliced_ell_sym(A, &S);
                    rc = upload_operator(L.A, S, S.n_pad, (A.nnzb() + A.nr) / 2, st); // stored: diagonal + one block per pair
                    if (!rc) rc = attach_in_lists(L.A, S, S.slice_base.back(), st);
                } else {
                    SlicedEll S;
                    pack_sliced_ell(A, true, &S);
                    rc = upload_operator(L.A, S, S.n_pad, A.nnzb(), st);
                }
                if (rc) return rc;
            }
            FS_HIP(L.minv.alloc((size_t)L.A.dm.n_slices * 21 * kSliceNodes));
            L.A.dm.minv = L.minv.p;
            L.A.dm.status = c->status_word;
            launch_block_jacobi(L.A.dm, st);
            FS_HIP(hipGetLastError());
            int32_t status_now = 0;
            {
                const int rcs = fetch_status_word(c, st, &status_now);
                if (rcs) return rcs;
            }

            if (status_now != 0) {
                {
                    const int rcc = clear_status_word(c, st);
                    if (rcc) return rcc;
                }
                return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: a diagonal block of coarse level " + std::to_string(l) +
                                                           " is not positive definite");
            }
        }
        rc = alloc_level_vectors(L, l == 0, kcycle, st);
        if (rc) return rc;
        lap("upload + block-Jacobi", l);
        const DeviceMatrix &Adev = amg_level_matrix(c, l);
        const bool coarsest = is_coarsest(l, L.n);
        if (coarsest) {
            std::vector<double> inv;
            if (L.n > 4096) return set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: coarsest level too large for a dense inverse");
            if (!have_host) return set_err(FEMSHELL_ERR_INVALID, "multigrid setup: the coarsest operator was not brought to the host");
            // the smallest operators are inverted on the host; beyond FEMSHELL_AMG_DENSE_DEVICE_MIN nodes (default 64; 250
            // until the host's 0.19 s at 145 nodes and 0.35 s at 204 showed up as most of the setup of the small coupled
            // examples) the inverse is computed on the matrix cores (amg_dense.hip: n^3 flops, 0.4 TFLOP at 1231 nodes)
            // (read per setup: the tests switch them inside one process)
            const long dense_device_min = getenv("FEMSHELL_AMG_DENSE_DEVICE_MIN") ? atol(getenv("FEMSHELL_AMG_DENSE_DEVICE_MIN")) : 64l;
            // (stored and applied in single precision unless FEMSHELL_AMG_DENSE_F32=0: the coarsest solve sits inside a K cycle
            //  inside a flexible Krylov method, 1e-7 there moves the iteration count by one or two and halves the 436 MB the
            //  four visits per iteration stream: panel 0.809 -> 0.785 s, cylinder 0.755 -> 0.744 s)
            const bool dense_f32 = !c->amg_fp64_only && !(getenv("FEMSHELL_AMG_DENSE_F32") && atoi(getenv("FEMSHELL_AMG_DENSE_F32")) == 0);
            H.coarse_lda = 0;
            H.dense = AmgDenseStats();
            H.coarse_inv32.release();
            if (L.n > dense_device_min) {
                rc = amg_dense_inverse_device(c, A, dense_f32, &H.coarse_inv, &H.coarse_inv32, &H.coarse_lda, &H.dense);
                if (rc) return rc;
                if (dense_f32) {
                    // Is the rounded inverse still an inverse?  Its entries are off by 6e-8 of the largest, the smallest
                    // eigenvalue of A^-1 lies kappa(A) below that: beyond kappa ~ 1e6 the float copy is not positive definite
                    // any more (a cantilever strip of t / L = 1 / 1600: CG breakdown).  ||A (inv32 v) - v|| / ||v|| on one
                    // vector of mixed frequencies shows kappa eps; above 0.05 the inverse is computed again and kept FP64.
                    double rel = 0.0;
                    rc = dense_inverse_defect(c, L, Adev, H, &rel);
                    if (rc) return rc;
                    H.dense.f32_defect = rel;
                    if (setup_verbose())
                        fprintf(stderr, "[femshell amg setup] coarsest inverse rounded to single precision: ||A inv32 v - v|| / ||v|| = %.2e (%s)\n", rel,
                                rel <= 0.05 ? "kept" : "above 0.05: the FP64 inverse is kept instead");
                    if (!(rel <= 0.05)) {
                        H.coarse_inv32.release();
                        rc = amg_dense_inverse_device(c, A, false, &H.coarse_inv, &H.coarse_inv32, &H.coarse_lda, &H.dense);
                        if (rc) return rc;
                        H.dense.f32_defect = rel;
                    }
                }
                lap("dense inverse on the matrix cores", l);
            } else {
                if (!dense_inverse(A, &inv))
                    return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: coarsest operator is not positive definite");
                FS_HIP(H.coarse_inv.upload(inv, st));
                FS_HIP(hipStreamSynchronize(st));
                lap("dense inverse on the host", l);
            }
            L.hA = std::move(A); // (kept at every size: a few MB, and the operator the dense inverse is checked against)
            break;
        }
        PowerIteration pw;
        rc = power_iteration_start(c, L, Adev, amg_power_iterations(), &pw);
        if (rc) return rc;
        bool lam_known = false;
        auto lam_of = [&](double *out) {
            if (!lam_known) {
                double lam = 0.0;
                const int r2 = power_iteration_finish(c, pw, &lam);
                if (r2) return r2;
                L.lam = amg_lambda_safety() * lam; // the power iteration approaches from below
                lam_known = true;
            }
            *out = L.lam;
            return (int)FEMSHELL_OK;
        };
        // coarsen
        if (L.A_on_device && !L.pattern.empty() && device_step(l, L.n)) {
            // on the device: the operator is in HBM, its pattern on the host (amg_device_setup.cpp)
            if ((int)H.levels.size() <= l + 1) H.levels.emplace_back(new AmgLevel());
            AmgLevel &N = *H.levels[(size_t)l + 1];
            std::vector<double> Bc;
            DevBuf<double> Bnext;
            Bsr Anext;
            NearNullSrc src;
            if (Bdev.p == nullptr) { // (the level came from a host step)
                FS_HIP(Bdev.upload(B, st));
            }
            src.B = Bdev.p;
            rc = amg_device_coarsen(c, L.A.dm, L.pattern, L, N, src, lam_of, keep_host, want_host_matrix(l + 1), &Anext, &Bc, &Bnext,
                                    [&](const char *what) { lap(what, l); });
            if (rc) return rc;
            std::swap(Bdev.p, Bnext.p);
            std::swap(Bdev.n, Bnext.n);
            N.A_on_device = true;
            if (keep_host && have_host) L.hA = std::move(A);
            L.pattern = HostEllPattern();
            A = std::move(Anext);
            B.swap(Bc);
            continue;
        }
        if (!have_host) return set_err(FEMSHELL_ERR_INVALID, "multigrid setup: level operator neither on the host nor coarsened on the device");
        {
            double lam_now = 0.0;
            rc = lam_of(&lam_now);
            if (rc) return rc;
            lap("power iteration", l);
        }
        std::vector<int32_t> agg;
        const int32_t na = aggregate_nodes(A, &agg);
        lap("aggregation", l);
        std::vector<double> Q, Bc, Dinv;
        tentative_prolongator(agg, na, B, &Q, &Bc);
        block_diagonal_inverse(A, &Dinv);
        lap("tentative P, D^-1", l);
        Bsr P, R, Ac;
        smoothed_prolongator(A, Dinv, agg, na, Q, (4.0 / 3.0) / L.lam, &P);
        std::vector<double>().swap(Q);
        std::vector<double>().swap(Dinv);
        lap("smoothed P", l);
        galerkin_product(A, P, &R, &Ac);
        lap("Galerkin product", l);
        {
            SlicedEll S;
            const int32_t nc_pad = (na + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
            pack_sliced_ell(P, false, &S);
            rc = upload_operator(L.P, S, nc_pad, P.nnzb(), st);
            if (rc) return rc;
            pack_sliced_ell(R, false, &S);
            rc = upload_operator(L.R, S, L.n_pad, R.nnzb(), st);
            if (rc) return rc;
        }
        lap("pack + upload P, R", l);
        if (keep_host) {
            L.hA = std::move(A);
            L.hP = std::move(P);
        }
        if (keep_host || l == 0) L.agg = std::move(agg);
        A = std::move(Ac);
        B.swap(Bc);
        Bdev.release();
    }
    // Chebyshev coefficients per level
    for (size_t l = 0; l + 1 < H.levels.size(); l++) {
        AmgLevel &L = *H.levels[l];
        const int deg = std::max(1, l == 0 ? opt.smoother_degree : opt.coarse_degree);
        const double lmax = L.lam, lmin = L.lam / opt.eig_ratio;
        const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
        L.inv_theta = 1.0 / theta;
        L.cheb_a.clear();
        L.cheb_c.clear();
        double rho = 1.0 / sigma;
        for (int k = 1; k < deg; k++) {
            const double rho_new = 1.0 / (2.0 * sigma - rho);
            L.cheb_a.push_back(rho_new * rho);
            L.cheb_c.push_back(2.0 * rho_new / delta);
            rho = rho_new;
        }
    }
    {
        // FEMSHELL_AMG_SMOOTH_F32: the Chebyshev products of the levels stream a single-precision copy of the operator's
        // values (half the bytes of the HBM-bound part of the cycle); residuals, the Krylov product, the K cycle's products,
        // the Galerkin operators themselves and the transfers stay FP64.  The soft modes of a shell sit twelve decades
        // below ||K||: only the smoother -- a polynomial in D^-1 A whose job is the upper end of the spectrum -- tolerates
        // 1e-7, and the flexible Krylov method around the cycle does not care that the preconditioner moved a little.
        // 1 (default): levels of at least 4096 nodes; 2: level 0 only, 3: every level, whatever their size (A/B runs, tests); 0: off.
        const char *e = getenv("FEMSHELL_AMG_SMOOTH_F32");
        const int mode = c->amg_fp64_only ? 0 : (e ? atoi(e) : 1);
        // (what the coarsening steps released -- A P, Q, the symbolic buffers -- joins the pool's kept blocks now: the copies below are
        //  carved from them instead of being fresh requests to the driver, context.hpp DevPool::alloc (3))
        if (mode != 0) (void)DevPool::get().flush_pending();
        for (size_t l = 0; l + 1 < H.levels.size(); l++) {
            AmgLevel &L = *H.levels[l];
            L.A32.release();
            const DeviceMatrix &A = amg_level_matrix(c, (int)l);
            // (the size of the WHOLE level decides on a row-partitioned one: every rank takes the same decision, which the
            //  fallback of femshell_solve -- a collective rebuild -- relies on)
            const int32_t level_nodes = L.dist ? std::max(L.n, L.n_global) : L.n;
            if (mode == 0 || (mode == 2 && l > 0) || (mode == 1 && level_nodes < 4096) || A.vals == nullptr) continue;
            // a level with clusters of rigidly coupled nodes keeps everything in double precision: what makes the clusters -- pairs of
            // blocks that nearly cancel -- is what a product in single precision loses (the flexible CG broke down under the copies on
            // every such shell tried, and the all-FP64 rebuild of femshell_solve did what this line does at once)
            if (L.patches) continue;
            const int64_t nv = (l == 0 ? (int64_t)pl.total_slots() : (int64_t)L.A.vals.n / 36) * 36;
            {
                FS_HIP(L.A32.alloc((size_t)nv));
                launch_to_f32(A.vals, L.A32.p, nv, st);
                if (const char *sb = getenv("FEMSHELL_AMG_SMOOTH_SIGBITS")) launch_round_sig(L.A32.p, nv, atoi(sb), st); // (experiment)
            }
            // the block-Jacobi inverse the smoothers apply, and the transfer operators (R = P^T value by value, so the
            // rounded pair is still a transposed pair and the cycle stays symmetric)
            const int64_t nm = (int64_t)A.n_slices * 21 * kSliceNodes;
            FS_HIP(L.minv32.alloc((size_t)nm));
            launch_to_f32(A.minv, L.minv32.p, nm, st);
            // (transfers onto a small level are bound by latency, where half-width loads lose: R onto the 1231-node level
            //  22 -> 36 us in single precision)
            const bool big_coarse = mode == 3 || H.levels[l + 1]->n >= 4096 || H.levels[l + 1]->n_global >= 4096;
            if (mode != 2 && big_coarse && L.P.vals.n > 0 && L.R.vals.n > 0) {
                FS_HIP(L.P32.alloc(L.P.vals.n));
                FS_HIP(L.R32.alloc(L.R.vals.n));
                launch_to_f32(L.P.vals.p, L.P32.p, (int64_t)L.P.vals.n, st);
                launch_to_f32(L.R.vals.p, L.R32.p, (int64_t)L.R.vals.n, st);
            }
            FS_HIP(hipGetLastError());
        }
        for (size_t l = 0; l + 1 < H.levels.size(); l++) { // what the smoothers and the transfers of the cycle multiply with
            AmgLevel &L = *H.levels[l];
            L.smooth_dm = amg_level_matrix(c, (int)l);
            L.smooth_dm.vals32 = L.A32.p;
            L.smooth_dm.minv32 = L.minv32.p;
            // a level whose vectors are exchanged with the neighbour ranks keeps the input of its products FP64 (the halo
            // exchange moves six doubles per node)
            L.smooth_dm.vec32 = !has_lowp_copy(L) ? 0 : std::min(smooth_vectors_f32(), L.dist ? 1 : 2);
            L.P.dm.vals32 = L.P32.p;
            L.R.dm.vals32 = L.R32.p;
            L.smooth_ready = true;
        }
    }
    FS_HIP(hipStreamSynchronize(st));
    lap("smoother coefficients, single-precision copies", (int)H.levels.size() - 1);
    H.valid = true;
    return FEMSHELL_OK;
}

// ---- the cycle -------------------------------------------------------------------------------------------------------
// One code path for single-rank contexts and for row-partitioned ones (amg_dist.cpp).  On a row-partitioned level every
// operator product is preceded by a halo exchange of its input (level 0: beside the interior slices, cg_driver.cpp), a
// restriction by one of the residual and a prolongation by one of the coarse correction (P was smoothed across the cuts);
// the K cycle's dot products are all-reduced; below the last row-partitioned level the rank's rows of the restricted
// residual are all-gathered and the replicated levels run as on one rank.
namespace {

// FEMSHELL_AMG_FUSED_CHEB=0: product and Chebyshev step of the full-storage levels as two launches (A/B runs)
bool fused_cheb()
{
    const char *e = getenv("FEMSHELL_AMG_FUSED_CHEB");
    return !(e && atoi(e) == 0);
}

// FEMSHELL_AMG_POST_INCREMENT=0: the residual the post-smoothing starts from as b - A x, an FP64 product with the iterate, instead
// of (restricted residual) - A (P x_c) on the smoother's copy of the operator (A/B runs)
bool post_increment()
{
    const char *e = getenv("FEMSHELL_AMG_POST_INCREMENT");
    return !(e && atoi(e) == 0);
}

// FEMSHELL_AMG_RESIDUAL_INCREMENT=0: the residual in front of a restriction as b - A x, a product of its own in FP64 (A/B runs)
bool residual_increment()
{
    const char *e = getenv("FEMSHELL_AMG_RESIDUAL_INCREMENT");
    return !(e && atoi(e) == 0);
}

// does the matrix view carry the single-precision copy of its values for the smoothing products?
bool has_lowp(const DeviceMatrix &A) { return A.vals32 != nullptr; }

// FEMSHELL_AMG_FUSE (bit mask; A/B runs and tests): the first step of a Chebyshev smoothing runs in the epilogue of the kernel that
// produces its residual.  1: k_pcg_update_start (the update of the flexible PCG + the second phase of q = K p + the start of the
// cycle's pre-smoothing on level 0); 2: k_sym_gather_start (second phase of the increment product in front of a post-smoothing,
// symmetric-storage levels) -- both repeat the arithmetic of the passes they replace bit for bit; 4: the same in the epilogue of
// k_spmv on full-storage levels (rounds x + c z as one multiply-add: not the same bits).  Default 3: measured on the 4M-triangle
// panel, with the one-lane-per-node vector kernels the fused passes take as long as the passes they replace (147 against 65 + 81
// us, 226 against 82 + 85 + 81 us: these kernels are bound by their load instructions, not by their bytes), so the gain is the
// launches only; bit 2 changed the iteration count by rounding luck (104 -> 106) and stays off.
int fuse_mask()
{
    const char *e = getenv("FEMSHELL_AMG_FUSE");
    return e ? atoi(e) : 3;
}

// FEMSHELL_AMG_K_MASK (experiments; round 6): bit l set = the cycle of level l is wrapped into the K cycle's two Krylov steps; a
// level whose bit is clear is visited once per visit of its parent, as in a V cycle.  Default: every level above the coarsest.
// MEASURED (profiles/r06_kcycle_levels.txt): the Krylov steps cannot be taken off any level -- 4M-triangle panel 100 -> 196
// iterations with the K cycle on level 1 only.
int k_cycle_mask()
{
    const char *e = getenv("FEMSHELL_AMG_K_MASK");
    return e && *e ? atoi(e) : ~0;
}

struct Cycle {
    femshell_ctx *c;
    Amg &H;
    const CgScalars *gate;
    hipStream_t st;
    int rc = FEMSHELL_OK;
    const int fuse = fuse_mask();
    const int k_mask = k_cycle_mask();
    // level 0: the caller's kernel (k_pcg_update_start) has taken the first step of the pre-smoothing already -- d in L.d, x = d
    bool pre_started0 = false;

    bool dist(int l) const { return H.levels[(size_t)l]->dist; }
    void halo(int l, double *x)
    {
        if (rc || !dist(l)) return;
        rc = level_halo_exchange(c, *H.levels[(size_t)l]->halo, x, 6, st);
    }
    // Products of a row-partitioned level >= 1 with the halo exchange of their input BESIDE the slices that read owned columns only
    // (round 5; level 0 has done so since round 2, cg_driver.cpp spmv_with_halo): pack and grouped send/recv on the halo stream,
    // the interior slices on the main stream meanwhile, the slices with ghost columns behind the exchange.  FEMSHELL_HALO_OVERLAP=0,
    // or a level without the slice order: the blocking sequence halo(l, x) + product.
    bool overlap(int l) const
    {
        const AmgLevel &L = *H.levels[(size_t)l];
        return L.dist && l >= 1 && c->halo_overlap && c->halo_stream != nullptr && L.order.p != nullptr && L.n_interior >= 0;
    }
    // span(begin, count): launches the product over the slices order[begin, begin + count) of level l on the main stream
    template <class Span> void overlapped(int l, double *x, Span span)
    {
        if (rc) return;
        AmgLevel &L = *H.levels[(size_t)l];
        const int ns = L.n_pad / kSliceNodes;
        hipError_t e = hipEventRecord(c->ev_p_ready, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->halo_stream, c->ev_p_ready, 0);
        if (e == hipSuccess) {
            rc = level_halo_exchange(c, *L.halo, x, 6, c->halo_stream);
            if (rc) return;
            e = hipEventRecord(c->ev_halo_done, c->halo_stream);
        }
        if (e == hipSuccess) {
            span(0, L.n_interior);
            e = hipStreamWaitEvent(st, c->ev_halo_done, 0);
        }
        if (e != hipSuccess) {
            rc = set_err(FEMSHELL_ERR_HIP, std::string("multigrid cycle, overlapped halo exchange: ") + hipGetErrorString(e));
            return;
        }
        span(L.n_interior, ns - L.n_interior);
    }
    // y = op(A x) of level l (>= 1 row-partitioned: overlapped when possible): first phase of a symmetric-storage product, or a
    // full-storage product with epilogue e
    void sym_phase1(int l, const DeviceMatrix &A, double *x, double *y)
    {
        if (overlap(l)) {
            const int32_t *order = H.levels[(size_t)l]->order.p;
            overlapped(l, x, [&](int b, int n) { (void)launch_spmv_span(A, x, y, nullptr, gate, order, b, n, 0, st); });
            return;
        }
        halo(l, x);
        launch_spmv_direct(A, x, y, nullptr, gate, st, has_lowp(A));
    }
    void full_product(int l, const DeviceMatrix &A, double *x, double *y, const SpmvEpilogue &e)
    {
        if (overlap(l)) {
            const int32_t *order = H.levels[(size_t)l]->order.p;
            overlapped(l, x, [&](int b, int n) { launch_spmv_epilogue_span(A, x, y, e, order, b, n, gate, st); });
            return;
        }
        halo(l, x);
        if (e.d_out != nullptr && e.start) launch_spmv_start(A, x, e.base_vec, y, e.d_out, e.xsol, e.c, gate, st);
        else if (e.d_out != nullptr) launch_spmv_cheb(A, x, e.base_vec, y, e.d_out, e.xsol, e.a, e.c, gate, st);
        else if (e.base_vec != nullptr) launch_spmv_axpy(A, x, y, e.base_vec, e.sign, gate, st);
        else launch_spmv(A, x, y, nullptr, gate, st);
    }
    // y = K x on level 0 of a row-partitioned context: the halo exchange beside the interior slices (symmetric storage with
    // defer: the direct part only, the consumer collects the transposed products)
    // (vals32: a smoothing product on the single-precision copy of K's values)
    // (lowp: a smoothing product on the single-precision copy of K's values that matrix view carries)
    void product0(double *x, double *y, bool defer, const DeviceMatrix *lowp = nullptr, int vec32 = 0)
    {
        if (rc) return;
        CgVectors vv;
        vv.s = const_cast<CgScalars *>(gate);
        int np = 0;
        rc = spmv_with_halo(c, vv, x, y, nullptr, &np, defer, lowp != nullptr ? lowp->vals32 : nullptr, vec32);
    }
    // out = b - A_l x
    void residual(int l, const double *b, double *x, double *out)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        const DeviceMatrix &A = amg_level_matrix(c, l);
        if (l == 0 && dist(0)) {
            if (A.symmetric) {
                product0(x, out, true);
                launch_sym_gather(A, out, b, -1.0, gate, st);
            } else {
                product0(x, L.q.p, false);
                launch_sub(b, L.q.p, out, 6ll * A.n_pad, st);
            }
            return;
        }
        halo(l, x);
        launch_spmv_axpy(A, x, out, b, -1.0, gate, st);
    }

    // (r_given: the residual of x, where the caller has it -- in L.r or a vector of its own, never L.q or L.d)
    // (d_started: the kernel that produced the residual took the first step too -- d = inv_theta D^-1 r in this vector, L.d or, on a
    //  full-storage level, L.q; x updated)
    // the cluster blocks' share of a smoothing step that applied c D^-1 r with the point blocks (amg_patch.hpp): d += c M r, x += c M r
    void patch(int l, const double *r, double coef, double *d, bool d_float, double *x)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        if (rc || !L.patches) return;
        launch_patch_correct(L.patches->view(), r, coef, d, d_float, x, gate, st);
    }
    void smooth(int l, const double *b, double *x, bool zero_guess, const double *r_given = nullptr, double *d_started = nullptr)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        // (the operator as the smoother sees it: single-precision copies of the values and of D^-1 where the level has them)
        const DeviceMatrix &A = L.smooth_ready ? L.smooth_dm : amg_level_matrix(c, l);
        // what the smoothing products keep in single precision besides the operator's values (DeviceMatrix::vec32)
        const int v32 = (A.symmetric && has_lowp(A)) ? A.vec32 : 0;
        const double *rcur = b;
        if (!zero_guess) {
            if (r_given == nullptr) residual(l, b, x, L.r.p);
            rcur = r_given != nullptr ? r_given : L.r.p;
        }
        if (d_started == nullptr) launch_cheb_start(A, rcur, L.d.p, x, L.inv_theta, !zero_guess, gate, st, v32);
        // full-storage levels: the direction alternates between the two vectors
        double *d_cur = d_started != nullptr ? d_started : L.d.p, *d_next = d_cur == L.d.p ? L.q.p : L.d.p;
        const bool d_is_float = v32 == 2; // (the direction of a symmetric-storage level that keeps it in single precision)
        patch(l, rcur, L.inv_theta, d_cur, d_is_float, x); // (behind a fused start as well: the kernel that took it applied the point blocks)
        for (size_t k = 0; k < L.cheb_a.size(); k++) {
            if (l == 0 && dist(0)) {
                product0(L.d.p, L.q.p, A.symmetric != 0, &A, v32);
                launch_cheb_step(A, rcur, L.q.p, L.r.p, L.d.p, x, L.cheb_a[k], L.cheb_c[k], gate, st, A.symmetric != 0, v32);
                patch(l, L.r.p, L.cheb_c[k], L.d.p, d_is_float, x);
            } else if (A.symmetric) { // first phase of the product; the step kernel collects the transposed products
                sym_phase1(l, A, L.d.p, L.q.p);
                launch_cheb_step(A, rcur, L.q.p, L.r.p, L.d.p, x, L.cheb_a[k], L.cheb_c[k], gate, st, true, v32);
                patch(l, L.r.p, L.cheb_c[k], L.d.p, d_is_float, x);
            } else if (fused_cheb()) { // product and step in one launch (the small levels are bound by launch latency)
                SpmvEpilogue e;
                e.base_vec = rcur;
                e.sign = -1.0;
                e.d_out = d_next;
                e.xsol = x;
                e.a = L.cheb_a[k];
                e.c = L.cheb_c[k];
                full_product(l, A, d_cur, L.r.p, e);
                patch(l, L.r.p, L.cheb_c[k], d_next, false, x);
                std::swap(d_cur, d_next);
            } else {
                halo(l, L.d.p);
                launch_spmv(amg_level_matrix(c, l), L.d.p, L.q.p, nullptr, gate, st);
                launch_cheb_step(A, rcur, L.q.p, L.r.p, L.d.p, x, L.cheb_a[k], L.cheb_c[k], gate, st);
                patch(l, L.r.p, L.cheb_c[k], L.d.p, false, x);
            }
            rcur = L.r.p;
        }
        last_r = rcur;
        last_d = A.symmetric || (l == 0 && dist(0)) ? L.d.p : d_cur;
    }
    // what the last smooth() left: the residual of its iterate BEFORE the last direction was added, and that direction
    const double *last_r = nullptr;
    double *last_d = nullptr;

    // r_base - A dvec without a product with the iterate.  Right behind a pre-smoothing from a zero guess the smoother carries
    // the residual of its iterate up to the last direction d it added, so b - A x = last_r - A d; behind the coarse-grid
    // correction e = P x_c the residual is the one that was restricted minus A e.  Either way the product is with an
    // INCREMENT of the iterate, its result of the size of the residual itself -- not of ||A|| ||x||, seven to nine decades
    // above it -- and the single-precision copy of the values serves (relative error 1e-7 of the increment; the numpy
    // restatement with these products rounded needs the same iterations, tools/lab/f32_vectors_experiment.py).
    // Returns where the result is: L.r, or L.q for the first of the two on a symmetric-storage level in FP64.
    // (start_x: the result is the residual a post-smoothing of the iterate start_x begins with -- where the kernel that finishes
    //  the product can take the smoothing's first step as well it does, d = inv_theta D^-1 r and start_x += d, and *d_started
    //  receives the vector that holds d; else *d_started stays null)
    double *residual_minus_product(int l, const double *r_base, double *dvec, double *start_x = nullptr, double **d_started = nullptr)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        const DeviceMatrix &A = L.smooth_ready ? L.smooth_dm : amg_level_matrix(c, l);
        // (a product that left its results in single precision -- DeviceMatrix::vec32 -- is collected into L.r, FP64)
        const bool q32 = A.symmetric && has_lowp(A) && A.vec32 >= 1;
        const bool d32 = q32 && A.vec32 == 2;
        const bool start = start_x != nullptr && d_started != nullptr && (fuse & (A.symmetric ? 2 : 4)) != 0;
        // the buffer the direct part of a symmetric product lands in: not the one the base vector lives in
        double *pb = (!q32 && r_base == L.q.p) ? L.r.p : L.q.p;
        // second phase of a symmetric product, with the smoothing's first step where asked for (dvec has been read by then: d
        // goes to L.d, where the step kernels of a symmetric-storage level expect it)
        auto gather = [&]() -> double * {
            double *out = q32 ? L.r.p : pb;
            if (start) {
                launch_sym_gather_start(A, pb, out, r_base, -1.0, L.d.p, start_x, L.inv_theta, q32, d32, gate, st);
                *d_started = L.d.p;
            } else if (q32) {
                launch_sym_gather(A, pb, r_base, -1.0, gate, st, true, L.r.p);
            } else {
                launch_sym_gather(A, pb, r_base, -1.0, gate, st);
            }
            return out;
        };
        if (l == 0 && dist(0)) {
            if (A.symmetric) {
                product0(dvec, pb, true, &A, A.vec32);
                return gather();
            }
            product0(dvec, L.q.p, false);
            launch_sub(r_base, L.q.p, L.r.p, 6ll * A.n_pad, st);
            return L.r.p;
        }
        if (A.symmetric) {
            sym_phase1(l, A, dvec, pb);
            return gather();
        }
        // (dvec is L.d or L.q, never L.r; r_base may be L.r)
        SpmvEpilogue e;
        e.base_vec = r_base;
        e.sign = -1.0;
        if (start) {
            // the product's Chebyshev epilogue as a first step: r = r_base - A dvec, d = inv_theta D^-1 r into the other direction vector
            double *d_out = dvec == L.d.p ? L.q.p : L.d.p;
            e.d_out = d_out;
            e.xsol = start_x;
            e.c = L.inv_theta;
            e.start = 1;
            *d_started = d_out;
        }
        full_product(l, A, dvec, L.r.p, e);
        return L.r.p;
    }

    // x = M_l(b): one cycle on level l
    void cycle(int l, const double *b, double *x)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        if ((size_t)l + 1 == H.levels.size()) {
            if (H.coarse_lda > 0) launch_dense_gemv_big(H.coarse_inv.p, H.coarse_inv32.p, H.coarse_lda, b, x, 6 * L.n, 6 * L.n_pad, gate, st);
            else launch_dense_gemv(H.coarse_inv.p, b, x, 6 * L.n, 6 * L.n_pad, gate, st);
            return;
        }
        AmgLevel &N = *H.levels[(size_t)l + 1];
        if (l == 0 && pre_started0) {
            pre_started0 = false;
            smooth(l, b, x, true, nullptr, L.d.p);
        } else {
            smooth(l, b, x, true);
        }
        double *rf = L.r.p; // r = b - A x
        const bool increments = residual_increment();
        if (increments) rf = residual_minus_product(l, last_r, last_d);
        else residual(l, b, x, L.r.p);
        // b_c = R r
        if (dist(l)) {
            halo(l, rf); // (R = P^T reaches the rows of the neighbours' nodes along the cut)
            if (N.dist) {
                launch_spmv(L.R.dm, rf, N.b.p, nullptr, gate, st);
            } else {
                // the rank's rows of the restricted residual, all-gathered into the replicated level
                launch_spmv(L.R.dm, rf, L.bown.p, nullptr, gate, st);
                if (N.part_begin.size() + 1 != N.part.size()) { // (once per hierarchy, not per restriction)
                    N.part_begin.assign(N.part.begin(), N.part.end() - 1);
                    N.part_end.assign(N.part.begin() + 1, N.part.end());
                }
                std::string e;
                if (!rc && !comm_gather_rows(c->comm, L.bown.p, N.b.p, N.part_begin, N.part_end, st, &e)) rc = set_err(FEMSHELL_ERR_COMM, e);
            }
        } else {
            launch_spmv(L.R.dm, rf, N.b.p, nullptr, gate, st);
        }
        const bool next_is_coarsest = (size_t)l + 2 == H.levels.size();
        if (H.opt.cycle == FEMSHELL_CYCLE_K && !next_is_coarsest && ((k_mask >> (l + 1)) & 1)) kcycle(l + 1);
        else cycle(l + 1, N.b.p, N.x.p);
        if (N.dist) halo(l + 1, N.x.p);
        if (increments && post_increment()) {
            // x += e, e = P x_c kept in the direction vector -- as floats where the smoothing products read theirs as floats --
            // and the residual the post-smoothing starts from is the restricted one minus A e
            const DeviceMatrix &A = L.smooth_ready ? L.smooth_dm : amg_level_matrix(c, l);
            const bool e32 = A.symmetric && has_lowp(A) && A.vec32 == 2;
            launch_spmv_axpy_keep(L.P.dm, N.x.p, x, x, 1.0, L.d.p, e32, gate, st);
            double *d_started = nullptr;
            const double *r2 = residual_minus_product(l, rf, L.d.p, x, &d_started);
            smooth(l, b, x, false, r2, d_started);
            return;
        }
        launch_spmv_axpy(L.P.dm, N.x.p, x, x, 1.0, gate, st); // x += P x_c
        smooth(l, b, x, false);
    }

    // y = A_l x on the FP64 operator of level l (the K cycle's own products)
    void krylov_product(int l, const DeviceMatrix &A, double *x, double *y)
    {
        if (A.symmetric) {
            sym_phase1(l, A, x, y);
            launch_sym_gather(A, y, nullptr, 1.0, gate, st);
        } else {
            full_product(l, A, x, y, SpmvEpilogue());
        }
    }

    // two steps of flexible CG on A_l x = b_l preconditioned by the cycle (Notay & Vassilevski's K cycle)
    void kcycle(int l)
    {
        AmgLevel &L = *H.levels[(size_t)l];
        const DeviceMatrix &A = amg_level_matrix(c, l);
        const int64_t n6 = 6ll * L.n_pad;
        const bool small = n6 <= kKcycSmall && !L.dist; // coefficient steps as single launches (amg_kernels.hpp)
        // row-partitioned level: the rank's sums are in L.ksums (left there by the last workgroup of the dot-product kernel); the
        // kernel that applies the coefficients forms them from the all-reduced sums itself -- three launches per coefficient step
        // where rounds 4-5 had five (two one-workgroup kernels around the collective: 12.7 launches of 4.3 us per outer iteration
        // on the 1/8 strip of the 4M panel, profiles/r06_dist_budget_N8.txt)
        auto reduce = [&]() {
            std::string e;
            if (!rc && !comm_allreduce_sum(c->comm, L.ksums.p, 3, st, &e)) rc = set_err(FEMSHELL_ERR_COMM, e);
        };
        cycle(l, L.b.p, L.c1.p);
        krylov_product(l, A, L.c1.p, L.v1.p);
        if (small) {
            launch_kcyc_step1_small(L.c1.p, L.v1.p, L.b.p, L.r2.p, n6, L.ks.p, gate, st);
        } else {
            if (L.dist) {
                launch_kcyc_dots_local(1, L.c1.p, L.v1.p, L.c1.p, L.b.p, nullptr, nullptr, n6, L.kscratch.p, L.ksums.p, gate, st);
                reduce();
            } else {
                launch_kcyc_dots(1, L.c1.p, L.v1.p, L.c1.p, L.b.p, nullptr, nullptr, n6, L.ks.p, L.kscratch.p, gate, st);
            }
            launch_kcyc_r2(L.b.p, L.v1.p, L.r2.p, n6, L.ks.p, gate, st, L.dist ? L.ksums.p : nullptr);
        }
        cycle(l, L.r2.p, L.c2.p);
        krylov_product(l, A, L.c2.p, L.v2.p);
        if (small) {
            launch_kcyc_step2_small(L.c1.p, L.c2.p, L.v1.p, L.v2.p, L.r2.p, L.x.p, n6, L.ks.p, gate, st);
        } else {
            if (L.dist) {
                launch_kcyc_dots_local(2, L.c2.p, L.v1.p, L.c2.p, L.v2.p, L.c2.p, L.r2.p, n6, L.kscratch.p, L.ksums.p, gate, st);
                reduce();
            } else {
                launch_kcyc_dots(2, L.c2.p, L.v1.p, L.c2.p, L.v2.p, L.c2.p, L.r2.p, n6, L.ks.p, L.kscratch.p, gate, st);
            }
            launch_kcyc_combine(L.c1.p, L.c2.p, L.x.p, n6, L.ks.p, gate, st, L.dist ? L.ksums.p : nullptr);
        }
    }
};

// host side of the stopping test
struct AmgPoll {
    int32_t next_check = 1, step = 1;
    int operator()(femshell_ctx *c, const CgVectors &v, int32_t it, int32_t max_it, CgScalars *hs)
    {
        if (it + 1 != next_check && it + 1 < max_it) return 0;
        FS_HIP(hipMemcpyAsync(hs, v.s, sizeof *hs, hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        CommWatch::heartbeat(); // (progress: the watchdog of multi-rank contexts counts from here again)
        if (hs->done != 0) return 1;
        if (step < 4) step *= 2;
        next_check += step;
        return 0;
    }
};

} // namespace

double *amg_apply_iterate(femshell_ctx *c, double *z) { return c->amg->dist ? c->amg->dist->x0.p : z; }

int amg_apply(femshell_ctx *c, const double *r, double *z, const CgScalars *gate, bool pre_started)
{
    Cycle cy{c, *c->amg, gate, c->stream};
    cy.pre_started0 = pre_started;
    if (c->amg->dist) {
        // the iterate of level 0 is an input of the halo product: it needs ghost space, which the CG's z does not have
        double *x = c->amg->dist->x0.p;
        cy.cycle(0, r, x);
        launch_copy(x, z, 6ll * c->plan.n_pad, gate, c->stream);
    } else {
        cy.cycle(0, r, z);
    }
    if (cy.rc) return cy.rc;
    FS_HIP(hipGetLastError());
    return FEMSHELL_OK;
}

// ||a||^2 and ||b||^2 over the owned rows of all ranks (two partial arrays of 128 groups, room for a third, and behind them the
// two-word staging slot of the all-reduce)
static int two_squared_norms(femshell_ctx *c, const double *a, const double *b, int64_t n6, double sums[2])
{
    hipStream_t st = c->stream;
    constexpr size_t kDotsStage = 3 * 128;
    FS_HIP(c->dots_scratch.alloc(kDotsStage + 2));
    const int groups = launch_two_dots(a, a, b, b, n6, c->dots_scratch.p, st);
    double hp[2 * 128];
    FS_HIP(hipMemcpyAsync(hp, c->dots_scratch.p, sizeof hp, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    sums[0] = sums[1] = 0.0;
    for (int g = 0; g < groups; g++) {
        sums[0] += hp[g];
        sums[1] += hp[128 + g];
    }
    if (c->comm.active()) { // every rank holds its own rows
        FS_HIP(hipMemcpyAsync(c->dots_scratch.p + kDotsStage, sums, 2 * sizeof(double), hipMemcpyHostToDevice, st));
        std::string e;
        if (!comm_allreduce_sum(c->comm, c->dots_scratch.p + kDotsStage, 2, st, &e)) return set_err(FEMSHELL_ERR_COMM, e);
        FS_HIP(hipMemcpyAsync(sums, c->dots_scratch.p + kDotsStage, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
    }
    return FEMSHELL_OK;
}
static int squared_norm(femshell_ctx *c, const double *a, int64_t n6, double *out)
{
    double sums[2];
    const int rc = two_squared_norms(c, a, a, n6, sums);
    *out = sums[0];
    return rc;
}

// Flexible preconditioned CG (beta = z.(r - r_old) / r_old.z_old, so that the K cycle's slightly varying operator
// does not break the recurrence); stopping rule and scalars as in cg_classic.
//
// Iterative refinement: on the thin-shell systems ||K|| ||x|| exceeds ||b|| by seven to nine orders of magnitude.
// Every product K p then carries rounding noise of eps ||K|| ||p||, part of which falls on the soft modes: the
// displacement error against a direct solve stalls at 2e-10 on the 250k-triangle roof however far the recurrence
// residual is driven.  After the recurrence has converged the residual of the iterate is therefore evaluated in
// double-double (k_residual_dd; in plain FP64 it is noise itself, and restarting from it made the error worse:
// 2e-10 -> 2e-9) and the correction equation K e = b - K x is solved by the same method from e = 0 in a vector of
// its own -- there the noise scales with ||e||, not ||x|| -- and added once: x += e.  (Continuing the recurrence on
// x itself does not help: every update x += alpha p rounds at eps ||x||.)  Measured on the roof: 2e-10 -> 2e-13
// with one pass.  femshell_pc_options::refine_passes passes at most (default 1; 0 = off).  A pass stops on the drop of its
// own right-hand side, whatever the residual tolerance of the solve: ||e|| / ||x|| of the pass is the displacement error of
// the iterate before it (manufactured solutions at 4M triangles: 4.8e-8 estimated, 4.8e-8 true) and the pass leaves about
// that times its drop -- it runs until that product, with the ||e_k|| so far, is a fifth of rtol (kernels.hip: kRefineTarget;
// 1e-4 flat with FEMSHELL_REFINE_ADAPTIVE=0); passes after the first run while the product exceeds rtol.
// *true_rr_out = ||b - K x||^2 (double-double) of the returned iterate.
int cg_amg(femshell_ctx *c, const CgVectors &v0, double rtol, int32_t max_it, double *true_rr_out, double *rec_rr_out, const double *x0)
{
    const DeviceMatrix &m = c->dm;
    hipStream_t st = c->stream;
    const int64_t n6 = 6ll * m.n_pad;
    *true_rr_out = -1.0;
    *rec_rr_out = -1.0;
    const int refine = c->amg->opt.refine_passes;
    // With a refinement pass to follow, the first phase stops a factor 100 = 1 / sqrt(drop of a pass) above the tolerance:
    // what the tolerance is for is the displacement error, the pass reduces the error of the iterate it starts from by
    // its drop whatever that iterate is, and the digits the recurrence adds beyond 100 rtol are the ones its rounding noise
    // spoils anyway (manufactured solutions at 4M triangles, first phase to 1e-10 / 1e-8: 99 / 76 iterations, error 2e-14 /
    // 1e-12 on the panel; 86 / 70 iterations, 2e-13 / 2e-11 on the cylinder; to 1e-7 the cylinder's error is 1.3e-10).
    // Should the estimate of the first pass still exceed the tolerance, one pass more than refine_passes may run.
    const bool loosened = refine >= 1 && rtol > 0.0;
    const double rtol_first = loosened ? std::min(100.0 * rtol, 1e-2) : rtol;
    const int pass_limit = refine + (loosened ? 1 : 0);
    // FEMSHELL_REFINE_ADAPTIVE=0: every pass to a drop of 1e-4 of its own right-hand side, as before round 5 (A/B runs)
    const bool adaptive_pass = !(getenv("FEMSHELL_REFINE_ADAPTIVE") && atoi(getenv("FEMSHELL_REFINE_ADAPTIVE")) == 0);
    CgVectors v = v0;
    CgScalars hs{};
    double pass_host[3] = {0.0, 0.0, 0.0}; // source of an asynchronous copy below: lives until the function's next synchronisation
    int32_t it = 0;
    c->refine = femshell_ctx::RefineStats();
    double pass_rhs_rr = 0.0; // ||rhs||^2 of the running refinement pass
    for (int pass = 0;; pass++) {
        int rc;
        if (pass == 0) {
            launch_pcg_init(m, v, st);
            rc = scalar_step(c, v, 1, CG_PHASE_FLEX_INIT, rtol_first);
            if (rc) return rc;
            if (x0 != nullptr) {
                // A solve from an initial guess (femshell_set_initial_guess): x0 becomes the accumulated iterate and the first phase
                // solves the correction equation K e = b - K x0 -- its right-hand side evaluated in double-double, as a refinement
                // pass's -- down to the threshold the step above derived from b.  What follows is the solve from zero: x0 + e takes
                // the place of the first phase's iterate, the refinement passes correct it.
                FS_HIP(c->xacc.alloc((size_t)n6 + 6 * (size_t)m.n_ghost));
                FS_HIP(c->xacc.zero(st));
                FS_HIP(c->rres.alloc((size_t)n6));
                launch_copy(x0, c->xacc.p, n6, nullptr, st);
                rc = halo_exchange(c, c->xacc.p, st);
                if (rc) return rc;
                launch_residual_dd(m, c->xacc.p, v0.b, c->rres.p, st);
                v.b = c->rres.p;
                launch_pcg_init(m, v, st); // x = 0, r = b - K x0, partial sums of r.r
                rc = scalar_step(c, v, 1, CG_PHASE_FLEX_WARM, rtol_first);
                if (rc) return rc;
            }
        } else {
            // correction equation: right-hand side = residual of the accumulated solution, evaluated in double-double
            TraceRange trace("femshell refinement pass: double-double residual");
            rc = halo_exchange(c, c->xacc.p, st); // no-op without a communicator
            if (rc) return rc;
            launch_residual_dd(m, c->xacc.p, v0.b, c->rres.p, st);
            v.b = c->rres.p;
            if (adaptive_pass) {
                // ||x||^2 of the iterate this pass corrects, for the pass's own stopping rule (kernels.hip: kRefineTarget)
                double xx = 0.0;
                rc = squared_norm(c, c->xacc.p, n6, &xx);
                if (rc) return rc;
                pass_host[0] = xx;
                pass_host[1] = 0.0;
                pass_host[2] = rtol;
                FS_HIP(hipMemcpyAsync(reinterpret_cast<char *>(v.s) + offsetof(CgScalars, pass_xx), pass_host, sizeof pass_host, hipMemcpyHostToDevice, st));
            }
            launch_pcg_init(m, v, st); // x = 0, r = rhs, partial sums of r.r
            // (rtol = 0: the pass stops on the relative drop kRefineDrop of its own right-hand side alone -- the residual
            //  tolerance of the solve says little about the displacement error on these systems: at 4M triangles a
            //  manufactured solution is 4e-9 off at a double-double residual of 8.7e-11 ||b||, and under uniform pressure
            //  the double-double residual of a converged iterate is 5.6e-5 ||b||, rounding noise of K x)
            rc = scalar_step(c, v, 1, CG_PHASE_FLEX_RESTART, 0.0);
            if (rc) return rc;
            FS_HIP(hipMemcpyAsync(&hs, v.s, sizeof hs, hipMemcpyDeviceToHost, st));
            FS_HIP(hipStreamSynchronize(st));
            *true_rr_out = hs.rr;
            pass_rhs_rr = hs.rr;
            // passes beyond the first run only while the error estimate of the previous one is above the tolerance
            const bool accurate = pass >= 2 && c->refine.correction_rel * c->refine.residual_reduction <= rtol;
            if (hs.done != 0 || pass > pass_limit || it >= max_it || accurate) {
                const int32_t one = 1; // the solve as a whole has converged
                FS_HIP(hipMemcpyAsync(reinterpret_cast<char *>(v.s) + offsetof(CgScalars, done), &one, sizeof one, hipMemcpyHostToDevice, st));
                launch_copy(c->xacc.p, v.x, n6, nullptr, st);
                FS_HIP(hipStreamSynchronize(st));
                return FEMSHELL_OK;
            }
        }
        rc = amg_apply(c, v.r, v.z, v.s);
        if (rc) return rc;
        launch_pcg_dots(m, v, st);
        rc = scalar_step(c, v, 1, CG_PHASE_FLEX_RZ0, rtol);
        if (rc) return rc;
        launch_copy(v.z, v.p, n6, v.s, st);
        AmgPoll poll;
        poll.next_check = it + 1;
        bool finished = false;
        // FEMSHELL_AMG_FUSE bit 0: the update kernel also finishes q = K p (second phase of the symmetric-storage product) and
        // takes the first step of the cycle's pre-smoothing on the new residual (k_pcg_update_start: three passes become one)
        AmgLevel &L0 = *c->amg->levels[0];
        const bool fused_update = (fuse_mask() & 1) != 0 && c->amg->levels.size() > 1;
        const DeviceMatrix &S0 = L0.smooth_ready ? L0.smooth_dm : m;
        const bool d32_0 = S0.symmetric && has_lowp(S0) && S0.vec32 == 2;
        for (; it < max_it; it++) {
            int n_partials = 0;
            const bool defer = fused_update && m.symmetric != 0;
            if (c->comm.active()) rc = spmv_with_halo(c, v, v.p, v.q, v.partials, &n_partials, defer);
            else if (defer) launch_spmv_direct(m, v.p, v.q, v.partials, v.s, st);
            else launch_spmv(m, v.p, v.q, v.partials, v.s, st);
            if (rc) return rc;
            rc = scalar_step(c, v, 1, CG_PHASE_ALPHA, rtol, n_partials);
            if (rc) return rc;
            if (fused_update) launch_pcg_update_start(S0, v, L0.d.p, amg_apply_iterate(c, v.z), L0.inv_theta, defer, d32_0, st);
            else launch_pcg_update(m, v, st);
            rc = scalar_step(c, v, 2, CG_PHASE_FLEX_CONV, rtol); // (r.r and, for the stopping rule of a refinement pass, x.x)
            if (rc) return rc;
            rc = amg_apply(c, v.r, v.z, v.s, fused_update);
            if (rc) return rc;
            launch_pcg_dots(m, v, st);
            rc = scalar_step(c, v, 2, CG_PHASE_FLEX_BETA, rtol);
            if (rc) return rc;
            launch_cg_direction(m, v, st);
            rc = poll(c, v, it, max_it, &hs);
            if (rc < 0) return rc;
            if (rc == 1) {
                finished = true;
                break;
            }
        }
        if (!finished) {
            FS_HIP(hipMemcpyAsync(&hs, v.s, sizeof hs, hipMemcpyDeviceToHost, st));
            FS_HIP(hipStreamSynchronize(st));
        }
        it = hs.iters;
        if (pass == 0 && x0 != nullptr) {
            // the iterate of the first phase is x0 + e, whatever ended the phase
            launch_add(v.x, c->xacc.p, n6, st);
            launch_copy(c->xacc.p, v.x, n6, nullptr, st);
            v.b = v0.b;
        }
        if (pass == 0) *rec_rr_out = hs.rr; // what the stopping rule of the first phase saw (a refinement pass stops on
                                            // the drop of its own right-hand side, see CG_PHASE_FLEX_RESTART)
        if (pass > 0) {
            // error estimate of the solve: ||e|| / ||x|| of this pass (the error of the iterate before it) and the factor
            // by which the pass reduced the residual of its correction equation
            double sums[2];
            rc = two_squared_norms(c, v.x, c->xacc.p, n6, sums);
            if (rc) return rc;
            c->refine.passes = pass;
            c->refine.correction_rel = sums[1] > 0.0 ? std::sqrt(sums[0] / sums[1]) : 0.0;
            c->refine.residual_reduction = pass_rhs_rr > 0.0 ? std::sqrt(hs.rr / pass_rhs_rr) : 0.0;
            // x += e; the context's x is the accumulated solution again
            launch_add(v.x, c->xacc.p, n6, st);
            launch_copy(c->xacc.p, v.x, n6, nullptr, st);
        }
        if (hs.done != 1 || rtol <= 0.0) return FEMSHELL_OK; // iteration limit or breakdown: no refinement
        if (pass == 0) {
            FS_HIP(c->xacc.alloc((size_t)n6 + 6 * (size_t)m.n_ghost)); // ghost space: input of the double-double residual
            FS_HIP(c->xacc.zero(st));
            FS_HIP(c->rres.alloc((size_t)n6));
            launch_copy(v.x, c->xacc.p, n6, nullptr, st);
        }
    }
    return FEMSHELL_OK;
}

// Algorithmic HBM bytes of one multigrid cycle, level by level (per_level[l]: everything the cycle does ON level l over all its
// visits per outer iteration: smoothing products and steps, the two residual increments, the transfers to and from level l+1,
// the K cycle's own products and vector passes; the dense solve on the coarsest level).  Counted is what the kernels have to
// stream as the cycle is built today (round 5) -- NOT 292 B per block for every product as rounds 2-4 counted:
//   * a smoothing product or a residual increment reads the smoother's copy of the operator: 148 B per stored block where the
//     level keeps the single-precision copy (144 B of values + 4 B column index), 292 B else; symmetric storage stores (and streams)
//     the diagonal and one block per pair; its input and output vectors are floats where DeviceMatrix::vec32 says so
//   * the K cycle's two products per coarse solve and nothing else of the cycle read the FP64 operator
//   * the transposed-product buffers of symmetric storage are overhead of the method and are not counted (as for bytes_spmv)
//   * vector passes: every vector a kernel reads or writes once, 48 B per node (24 B as floats), the block-Jacobi inverse 168 B
//     per node (84 B from its single-precision copy)
double amg_cycle_bytes(const femshell_ctx *c, std::vector<double> *per_level)
{
    const Amg &H = *c->amg;
    const size_t nl = H.levels.size();
    std::vector<double> visits(nl, 0.0), bytes(nl, 0.0);
    visits[0] = 1.0;
    const bool kcyc = H.opt.cycle == FEMSHELL_CYCLE_K;
    for (size_t l = 0; l + 1 < nl; l++) {
        const AmgLevel &L = *H.levels[l], &N = *H.levels[l + 1];
        const DeviceMatrix &A = amg_level_matrix(c, (int)l);
        const double n = (double)L.n;
        const bool lowp = L.A32.p != nullptr;
        const int v32 = (A.symmetric && lowp) ? L.smooth_dm.vec32 : 0;
        // blocks a product of the level operator streams (symmetric storage: the stored ones)
        const double blocks = l == 0 ? (double)c->plan.stored_blocks : (A.symmetric ? 0.5 * ((double)L.nnzb + n) : (double)L.nnzb);
        const double deg = (double)L.cheb_a.size() + 1.0;
        const double vec = 48.0 * n, fvec = 24.0 * n;
        const double minv = (lowp ? 84.0 : 168.0) * n;
        const double in_vec = v32 == 2 ? fvec : vec, out_vec = v32 >= 1 ? fvec : vec;
        const double smooth_product = blocks * (lowp ? 148.0 : 292.0) + in_vec + out_vec;
        // per visit: 2 (deg - 1) products inside the two smoothings + the two increments
        double per_visit = 2.0 * deg * smooth_product;
        // Chebyshev steps (2 (deg - 1)): rin, q, d, x, D^-1 read; rout, d, x written
        per_visit += 2.0 * (deg - 1.0) * (vec + out_vec + in_vec + vec + minv + vec + in_vec + vec);
        // the two starts: r, D^-1 (and x behind the coarse correction) read; d, x written -- and the two second phases / axpys
        // of the increments: product and base vector read, residual written
        per_visit += 2.0 * (vec + minv + in_vec + vec) + vec + 2.0 * (out_vec + vec + vec);
        // transfers: R (rows of level l+1) and P (rows of level l), single-precision copies where the setup made them
        const double rp_block = L.P32.p != nullptr ? 148.0 : 292.0;
        // (the prolongation is a product over the rows of this level; the restriction one over the rows of the next and is booked
        //  there -- which is also where a kernel trace by grid size finds it, tools/amg_level_times.py)
        per_visit += rp_block * (double)L.P.nnzb + (48.0 * N.n + 2.0 * vec + in_vec);
        bytes[l + 1] += visits[l] * (rp_block * (double)L.R.nnzb + vec + 48.0 * N.n);
        // the K cycle on this level (levels >= 1 that are not the coarsest): per coarse solve = per two visits, two FP64 products
        // and the vector passes of the two coefficient steps
        if (kcyc && l >= 1) per_visit += 0.5 * (2.0 * (292.0 * blocks + 2.0 * vec) + 14.0 * vec);
        bytes[l] += visits[l] * per_visit;
        const bool next_k = kcyc && l + 2 < nl;
        visits[l + 1] = visits[l] * (next_k ? 2.0 : 1.0);
    }
    const AmgLevel &C = *H.levels.back();
    bytes[nl - 1] += visits.back() * ((H.coarse_inv32.p != nullptr ? 4.0 : 8.0) * 36.0 * (double)C.n * (double)C.n + 96.0 * C.n);
    double total = 0.0;
    for (double b : bytes) total += b;
    if (per_level != nullptr) *per_level = bytes;
    return total;
}

} // namespace femshell
