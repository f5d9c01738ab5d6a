// amg_dist.cpp -- the multigrid hierarchy of a row-partitioned context (one rank per GPU).
//
// The reference partitions its elements over the MPI ranks and leaves the preconditioner to PETSc, which builds it on the
// distributed matrix (doc/implementation.tex:463-472: only the mesh is replicated).  Here: levels 0 .. d-1 of the hierarchy
// are row-partitioned like K, the first level of at most FEMSHELL_AMG_DIST_MIN nodes is all-gathered once and it and
// everything below run replicated on every rank (amg_solve.cpp: amg_finish_hierarchy).
//
// A row-partitioned coarsening step.  Only the aggregation knows about the ranks: every rank aggregates the graph of its own
// rows without the edges that leave it, so aggregates never span ranks and the coarse rows are numbered rank by rank.
// Tentative prolongator (QR per aggregate), smoothing P = (I - w D^-1 A) P0 with the WHOLE A and the Galerkin operator
// A_c = P^T A P are the single-rank ones -- the hierarchy equals the one a single rank builds from the same aggregates
// (oracle/amg_oracle.py: aggregate_by_rank) -- and every rank computes its rows of P and A_c itself.  What it needs from its
// neighbours for that are rows of the nodes along the cuts, which travel the way ghost entries of vectors do, only wider:
//   1. Q and the aggregate id of every ghost node                       (36 + 1 doubles per node)
//      -> own rows of P: the product A P0 reads the ghost rows of P0
//   2. the rows of P of the ghost nodes                                 (Wp x 37 doubles per node)
//      -> own rows of A P: (A P)_i = sum_j A_ij P_j, j owned or ghost
//   3. the rows of A P of the ghost nodes                               (Wap x 37 doubles per node)
//      -> own rows of A_c: (A_c)_IJ = sum_i P_iI^T (A P)_iJ over the fine rows i that see aggregate I -- the rank's own and,
//         since P was smoothed across the cut, ghost rows
// Column keys are GLOBAL coarse ids until the values are complete; then the columns of P and A_c become local ids (own
// aggregates, padding, ghost aggregates) and the ghost lists are exchanged once so that every rank knows which of its rows
// its neighbours read (level_halo).  In the cycle a restriction reads the ghost entries of the residual and a prolongation
// the ghost entries of the coarse correction: one halo exchange each, no reduction across ranks.
#include "amg_device.hpp"
#include "amg_pattern.hpp"
#include "trace.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace femshell {

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every rank's value (one-hot all-reduce)
int allgather_i64(femshell_ctx *c, int64_t mine, std::vector<int64_t> *all)
{
    const int world = c->comm.world;
    std::vector<double> h((size_t)world, 0.0);
    h[(size_t)c->comm.rank] = (double)mine;
    DevBuf<double> d;
    FS_HIP(d.upload(h, c->stream));
    std::string e;
    if (!comm_allreduce_sum(c->comm, d.p, world, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    FS_HIP(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    CommWatch::heartbeat(); // (every rank got here: progress of the phase the watchdog times)
    all->resize((size_t)world);
    for (int r = 0; r < world; r++) (*all)[(size_t)r] = (int64_t)std::llround(h[(size_t)r]);
    return FEMSHELL_OK;
}

// the ranks' lists of doubles, rank after rank (*offsets: world + 1 positions)
int allgather_doubles(femshell_ctx *c, const double *mine, int64_t count, std::vector<double> *all, std::vector<int64_t> *offsets)
{
    std::vector<int64_t> counts;
    int rc = allgather_i64(c, count, &counts);
    if (rc) return rc;
    const int world = c->comm.world;
    std::vector<int64_t> begin((size_t)world), end((size_t)world);
    offsets->assign((size_t)world + 1, 0);
    for (int r = 0; r < world; r++) {
        begin[(size_t)r] = (*offsets)[(size_t)r];
        end[(size_t)r] = (*offsets)[(size_t)r + 1] = begin[(size_t)r] + counts[(size_t)r];
    }
    const int64_t total = offsets->back();
    all->assign((size_t)total, 0.0);
    if (total == 0) return FEMSHELL_OK;
    DevBuf<double> dm, df;
    FS_HIP(dm.alloc((size_t)std::max<int64_t>(count, 1)));
    if (count > 0) FS_HIP(hipMemcpyAsync(dm.p, mine, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FS_HIP(df.alloc((size_t)total));
    std::string e;
    if (!comm_gather_pieces(c->comm, dm.p, df.p, begin, end, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    FS_HIP(hipMemcpyAsync(all->data(), df.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    return FEMSHELL_OK;
}

// rows of `width` doubles of the ghost nodes of a level into their places behind the owned rows of src (in place)
int exchange_rows(femshell_ctx *c, LevelHalo &H, double *vec, int width) { return level_halo_exchange(c, H, vec, width, c->stream); }

// Ghost exchange of the coarse level a step creates.  ghost_keys: the global ids of the coarse nodes this rank reads besides
// its own, ascending (= grouped by owner).  Every rank learns all lists, so it knows what its neighbours read of its rows.
int build_level_halo(femshell_ctx *c, const std::vector<int32_t> &part, int32_t n_pad, const std::vector<int32_t> &ghost_keys,
                     LevelHalo *out)
{
    LevelHalo &H = *out;
    const int world = c->comm.world, me = c->comm.rank;
    H.n_pad = n_pad;
    H.n_ghost = (int32_t)ghost_keys.size();
    H.peers.clear();
    std::vector<double> mine(ghost_keys.begin(), ghost_keys.end()), all;
    std::vector<int64_t> off;
    int rc = allgather_doubles(c, mine.data(), (int64_t)mine.size(), &all, &off);
    if (rc) return rc;
    for (int r = 0; r < world; r++) {
        if (r == me) continue;
        HaloPeer p;
        p.rank = r;
        // what I read of rank r's rows: a contiguous stretch of my (ascending) list
        const auto lo = std::lower_bound(ghost_keys.begin(), ghost_keys.end(), part[(size_t)r]);
        const auto hi = std::lower_bound(ghost_keys.begin(), ghost_keys.end(), part[(size_t)r + 1]);
        p.recv_offset = (int32_t)(lo - ghost_keys.begin());
        p.recv_count = (int32_t)(hi - lo);
        // what rank r reads of mine, in the order of its list
        for (int64_t q = off[(size_t)r]; q < off[(size_t)r + 1]; q++) {
            const int32_t key = (int32_t)all[(size_t)q];
            if (key >= part[(size_t)me] && key < part[(size_t)me + 1]) p.send_nodes.push_back(key - part[(size_t)me]);
        }
        if (p.recv_count > 0 || !p.send_nodes.empty()) H.peers.push_back(std::move(p));
    }
    H.send_offsets.clear();
    std::vector<int32_t> flat;
    for (const HaloPeer &p : H.peers) {
        H.send_offsets.push_back((int32_t)flat.size());
        flat.insert(flat.end(), p.send_nodes.begin(), p.send_nodes.end());
    }
    H.total_send = (int32_t)flat.size();
    if (flat.empty()) flat.push_back(0);
    FS_HIP(H.send_nodes.upload(flat, c->stream));
    FS_HIP(H.sendbuf.alloc((size_t)std::max(H.total_send, 1) * 6));
    H.sendbuf_width = 6;
    FS_HIP(hipStreamSynchronize(c->stream));
    return FEMSHELL_OK;
}

// fixed-width ghost rows behind the own rows of a pattern: slices of width W, cols / count filled once the keys are known
void append_ghost_rows(EllPattern &E, int32_t n_ghost, int W)
{
    const int32_t g_pad = (n_ghost + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    const int32_t g_slices = g_pad / kSliceNodes;
    for (int32_t s = 0; s < g_slices; s++) {
        E.slice_width.push_back(W);
        E.slice_base.push_back(E.slice_base.back() + (int64_t)W * kSliceNodes);
    }
    E.n_slices += g_slices;
    E.n_rows = E.n_pad + n_ghost; // (rows between the owned ones and n_pad have no entries)
    E.n_pad += g_pad;
    E.count.resize((size_t)E.n_pad, 0);
    E.cols.resize((size_t)E.total(), 0);
    E.max_width = std::max(E.max_width, W);
}

// keys of the received rows (n_ghost x W, -1 = none) into the ghost rows of E, compacted to the front of each row
void fill_ghost_rows(EllPattern &E, int32_t first_row, int32_t n_ghost, int W, const std::vector<int32_t> &keys)
{
    for (int32_t g = 0; g < n_ghost; g++) {
        const int32_t r = first_row + g;
        const int s = r / kSliceNodes, n = r % kSliceNodes;
        int k = 0;
        for (int q = 0; q < W; q++) {
            const int32_t key = keys[(size_t)g * W + q];
            if (key >= 0) E.cols[(size_t)(E.slice_base[s] + (int64_t)k++ * kSliceNodes + n)] = key;
        }
        E.count[(size_t)r] = (uint8_t)k;
        E.nnzb += k;
    }
}

// the largest of the ranks' values -- and the ranks' agreement on a failure only some of them see: a rank hands over -1 for
// "my part failed" (its own error text is set), and every rank leaves with an error instead of walking on into the next
// exchange, where the healthy ones would wait for the failed one until the watchdog ends them
int global_max(femshell_ctx *c, int64_t mine, int64_t *out, const char *what)
{
    std::vector<int64_t> all;
    const std::string local = mine < 0 ? last_err() : std::string();
    const int rc = allgather_i64(c, mine, &all);
    if (rc) return rc;
    int failed = 0;
    for (int64_t v : all) failed += v < 0 ? 1 : 0;
    if (mine < 0) return set_err(FEMSHELL_ERR_UNSUPPORTED, local);
    if (failed)
        return set_err(FEMSHELL_ERR_COMM, std::string("multigrid setup: ") + what + " failed on " + std::to_string(failed) +
                                              " other rank(s) of the row partition (their own message says why)");
    *out = *std::max_element(all.begin(), all.end());
    return FEMSHELL_OK;
}

struct StepResult {
    bool next_dist = false;
    Bsr A_global;                  // the coarse operator on every rank (next level replicated)
    std::vector<double> B_global;  // ... and its near-null space
    DevBuf<double> Bc_dev;         // own rows of the coarse near-null space (next level row-partitioned)
};

// One coarsening step of a row-partitioned level (see the head of this file).
// A: the rank's rows of the level operator in HBM (block-Jacobi inverse valid), pat: host copy of its pattern (columns
// local: own rows, padding, ghosts), Bsrc: near-null space of the own rows.
int dist_coarsen(femshell_ctx *c, const femshell_pc_options &opt, int level, const DeviceMatrix &Adev, const HostEllPattern &pat,
                 AmgLevel &L, AmgLevel &N, const NearNullSrc &Bsrc, bool keep_host, int max_level_index, StepResult *res,
                 const std::function<void(const char *)> &lap)
{
    hipStream_t st = c->stream;
    const int world = c->comm.world, me = c->comm.rank;
    LevelHalo &FH = *L.halo;
    const int32_t n = pat.n, n_pad = FH.n_pad, n_ghost = FH.n_ghost, n_local = n_pad + n_ghost;
    // A failure only this rank sees (an allocation, an upload, a launch) must not send it home while its peers sit in the next
    // exchange -- a mismatched collective that only the watchdog would end.  The local work between two collectives therefore
    // runs as blocks whose first failure is kept in `pending` (its text in last_err()), and every collective of this step is
    // preceded by an agreement of the ranks on it: all of them leave with an error, or all of them go on.
    int pending = FEMSHELL_OK;
    auto local = [&](const std::function<int()> &block) {
        if (pending) return;
        pending = block();
    };
    auto agree = [&](const char *what) -> int {
        int64_t unused = 0;
        const int r = global_max(c, pending ? -1 : 0, &unused, what);
        pending = FEMSHELL_OK;
        return r;
    };
    // ---- rank-local aggregation
    Bsr G; // rows: own nodes; columns: local ids, ghosts included
    graph_of_pattern(pat, &G);
    std::vector<int32_t> agg;
    int32_t na = 0;
    {
        Bsr Gown;
        Gown.nr = Gown.nc = n;
        Gown.ptr.assign((size_t)n + 1, 0);
        for (int32_t a = 0; a < n; a++) {
            int64_t cnt = 0;
            for (int64_t q = G.ptr[a]; q < G.ptr[a + 1]; q++) cnt += G.col[(size_t)q] < n;
            Gown.ptr[(size_t)a + 1] = Gown.ptr[(size_t)a] + cnt;
        }
        Gown.col.resize((size_t)Gown.ptr[(size_t)n]);
        parallel_chunks(n, [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                int64_t w = Gown.ptr[(size_t)a];
                for (int64_t q = G.ptr[a]; q < G.ptr[a + 1]; q++)
                    if (G.col[(size_t)q] < n) Gown.col[(size_t)w++] = G.col[(size_t)q];
            }
        });
        // (clusters of rigidly coupled nodes -- amg_patch.hpp, own rows only -- are glued into one node each first)
        na = L.patches && L.patches->glue ? aggregate_nodes_glued(Gown, L.patches->label_p, &agg) : aggregate_nodes(Gown, &agg);
    }
    std::vector<int64_t> counts;
    int rc = allgather_i64(c, na, &counts);
    if (rc) return rc;
    std::vector<int32_t> cpart((size_t)world + 1, 0);
    for (int r = 0; r < world; r++) cpart[(size_t)r + 1] = cpart[(size_t)r] + (int32_t)counts[(size_t)r];
    const int32_t na_global = cpart.back(), key0 = cpart[(size_t)me];
    const int32_t na_pad = (na + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    // does the coarse level stay row-partitioned?  (never the coarsest one, never the last level allowed)
    res->next_dist = na_global > std::max(amg_dist_min(), opt.coarsest_nodes) && level + 2 < max_level_index;
    lap("graph + rank-local aggregation");

    // ---- tentative prolongator of the own rows, Q and aggregate keys of the ghost rows from their owners
    DevBuf<double> d_Q, d_keyf;
    DevBuf<int32_t> d_key;
    std::vector<int32_t> key((size_t)n_local, -1); // global coarse id of every local fine node
    {
        std::vector<int32_t> gptr((size_t)na + 1, 0), order((size_t)n);
        for (int32_t i = 0; i < n; i++) gptr[(size_t)agg[(size_t)i] + 1]++;
        int32_t largest = 0;
        for (int32_t I = 0; I < na; I++) {
            largest = std::max(largest, gptr[(size_t)I + 1]);
            gptr[(size_t)I + 1] += gptr[(size_t)I];
        }
        {
            std::vector<int32_t> fill(gptr.begin(), gptr.end() - 1);
            for (int32_t i = 0; i < n; i++) order[(size_t)fill[(size_t)agg[(size_t)i]]++] = i;
        }
        DevBuf<int32_t> d_gptr, d_order;
        std::vector<double> keyf((size_t)n_local, -1.0);
        for (int32_t i = 0; i < n; i++) keyf[(size_t)i] = (double)(key[(size_t)i] = key0 + agg[(size_t)i]);
        local([&]() -> int {
            FS_HIP(d_gptr.upload(gptr, st));
            FS_HIP(d_order.upload(order, st));
            FS_HIP(d_Q.alloc((size_t)n_local * 36));
            FS_HIP(d_Q.zero(st));
            FS_HIP(res->Bc_dev.alloc((size_t)std::max(na, 1) * 36));
            const bool in_memory = getenv("FEMSHELL_AMG_QR") && std::string(getenv("FEMSHELL_AMG_QR")) == "memory";
            launch_amg_tentative_qr(Bsrc, d_gptr.p, d_order.p, na, largest, d_Q.p, res->Bc_dev.p, in_memory, st);
            FS_HIP(hipGetLastError());
            FS_HIP(d_keyf.upload(keyf, st));
            return FEMSHELL_OK;
        });
        rc = agree("the tentative prolongator");
        if (rc) return rc;
        rc = exchange_rows(c, FH, d_Q.p, 36);
        if (rc) return rc;
        rc = exchange_rows(c, FH, d_keyf.p, 1);
        if (rc) return rc;
        local([&]() -> int {
            FS_HIP(hipMemcpyAsync(keyf.data() + n_pad, d_keyf.p + n_pad, (size_t)n_ghost * sizeof(double), hipMemcpyDeviceToHost, st));
            FS_HIP(hipStreamSynchronize(st));
            for (int32_t g = 0; g < n_ghost; g++) key[(size_t)n_pad + g] = (int32_t)keyf[(size_t)n_pad + g];
            std::vector<int32_t> keyd(key);
            for (int32_t &k : keyd) k = std::max(k, 0); // (padding rows)
            FS_HIP(d_key.upload(keyd, st));
            FS_HIP(hipStreamSynchronize(st));
            return FEMSHELL_OK;
        });
        // (the keys of the ghost rows are read on the host below: the ranks agree on their arrival first)
        rc = agree("the aggregate keys of the ghost rows");
        if (rc) return rc;
    }
    lap("tentative P, ghost rows of Q");

    // ---- P: per own row the sorted distinct aggregates (keys) of its neighbours, ghost neighbours included
    std::vector<int64_t> pptr((size_t)n + 1, 0);
    RawVec<int32_t> pcol;
    {
        std::vector<uint8_t> cnt((size_t)n, 0);
        RawVec<int32_t> tmp_all((size_t)G.ptr[(size_t)n]);
        // rows of clustered nodes: what ALL the cluster's members see (amg_device_setup.cpp: P = P0 - omega B^-1 A P0)
        const AmgPatches *patches = L.patches.get();
        std::vector<std::vector<int32_t>> cluster_rows(patches ? (size_t)patches->n_clusters : 0);
        if (patches)
            parallel_chunks(patches->n_clusters, [&](int64_t k0, int64_t k1) {
                for (int64_t k = k0; k < k1; k++) {
                    std::vector<int32_t> &r = cluster_rows[(size_t)k];
                    for (int32_t t = patches->h_ptr[(size_t)k]; t < patches->h_ptr[(size_t)k + 1]; t++) {
                        const int32_t j = patches->h_nodes[(size_t)t];
                        for (int64_t q = G.ptr[(size_t)j]; q < G.ptr[(size_t)j + 1]; q++) r.push_back(key[(size_t)G.col[(size_t)q]]);
                    }
                    std::sort(r.begin(), r.end());
                    r.erase(std::unique(r.begin(), r.end()), r.end());
                }
            }, 64);
        parallel_chunks(n, [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                if (patches && patches->label_p[(size_t)a] >= 0) {
                    cnt[(size_t)a] = (uint8_t)std::min<size_t>(cluster_rows[(size_t)patches->label_p[(size_t)a]].size(), 255);
                    continue;
                }
                int32_t *t = &tmp_all[(size_t)G.ptr[(size_t)a]];
                int m = 0;
                for (int64_t q = G.ptr[(size_t)a]; q < G.ptr[(size_t)a + 1]; q++) t[m++] = key[(size_t)G.col[(size_t)q]];
                std::sort(t, t + m);
                m = (int)(std::unique(t, t + m) - t);
                cnt[(size_t)a] = (uint8_t)std::min(m, 255);
            }
        });
        for (int32_t a = 0; a < n; a++) pptr[(size_t)a + 1] = pptr[(size_t)a] + cnt[(size_t)a];
        pcol.resize((size_t)pptr[(size_t)n]);
        parallel_chunks(n, [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                if (patches && patches->label_p[(size_t)a] >= 0)
                    std::copy_n(cluster_rows[(size_t)patches->label_p[(size_t)a]].data(), cnt[(size_t)a], &pcol[(size_t)pptr[(size_t)a]]);
                else std::copy_n(&tmp_all[(size_t)G.ptr[(size_t)a]], cnt[(size_t)a], &pcol[(size_t)pptr[(size_t)a]]);
            }
        });
    }
    auto p_index = [&](int32_t row, int32_t J) -> int {
        const int32_t *b = &pcol[(size_t)pptr[(size_t)row]], *e = &pcol[(size_t)pptr[(size_t)row + 1]];
        return (int)(std::lower_bound(b, e, J) - b);
    };
    std::vector<uint8_t> pmap_own((size_t)pat.slice_base.back(), 0), pmap_in(std::max<size_t>(pat.in_slots.size(), 1), 0);
    parallel_chunks(n, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), nn = (int)(a % kSliceNodes);
            for (int k = 0; k < pat.count[(size_t)a]; k++) {
                const int64_t slot = pat.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn;
                pmap_own[(size_t)slot] = (uint8_t)p_index((int32_t)a, key[(size_t)pat.cols[(size_t)slot]]);
            }
            if (pat.symmetric)
                for (int k = 0; k < pat.in_width[(size_t)s]; k++) {
                    const size_t e = (size_t)(pat.in_base[(size_t)s] + (int64_t)k * kSliceNodes + nn);
                    if (pat.in_slots[e] >= 0) pmap_in[e] = (uint8_t)p_index((int32_t)a, key[(size_t)pat.in_rows[e]]);
                }
        }
    });
    EllPattern eP, eAP, eR, eAc;
    bool ok = pack_pattern(n, pptr.data(), pcol.data(), false, &eP);
    if (!ok) (void)set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: a row of the prolongator has more than 255 blocks");
    int64_t Wp = 0;
    rc = global_max(c, ok ? eP.max_width : -1, &Wp, "the pattern of the prolongator");
    if (rc) return rc;
    const int64_t own_total_P = eP.total();
    append_ghost_rows(eP, n_ghost, (int)Wp);
    DevBuf<double> vP, vAP, vR, vAc;
    DevPattern dP, dAP, dR, dAc;
    EllView wP, wAP, wR, wAc;
    DevBuf<uint8_t> d_pmap_own, d_pmap_in;
    local([&]() -> int {
        FS_HIP(vP.alloc((size_t)eP.total() * 36));
        const int r = upload_pattern(eP, dP, vP.p, &wP, st);
        if (r) return r;
        FS_HIP(d_pmap_own.upload(pmap_own, st));
        FS_HIP(d_pmap_in.upload(pmap_in, st));
        EllView own = wP; // the kernel writes the own rows: one lane per slot
        own.n_rows = n;
        own.n_slices = n_pad / kSliceNodes;
        own.total = own_total_P;
        launch_amg_prolongator(Adev, d_key.p, d_Q.p, (4.0 / 3.0) / L.lam, d_pmap_own.p, d_pmap_in.p, own, st);
        if (L.patches) { // the cluster blocks' share of the smoothing (the keys are the global coarse ids here, ghosts included)
            int width = 0;
            for (int32_t sl = 0; sl < n_pad / kSliceNodes; sl++) width = std::max(width, (int)eP.slice_width[(size_t)sl]);
            launch_patch_prolongator(Adev, d_key.p, d_Q.p, (4.0 / 3.0) / L.lam, own, L.patches->view(), width, st);
        }
        FS_HIP(hipGetLastError());
        return FEMSHELL_OK;
    });
    // ---- the rows of P of the ghost nodes
    DevBuf<double> rowbuf; // received rows (reused for A P)
    DevBuf<int32_t> d_keys;
    std::vector<int32_t> hkeys;
    // (local work in front of the exchange, agreement, the exchange, local work behind it -- whose failure the next
    //  agreement of this step picks up)
    auto fetch_ghost_rows = [&](const EllView &M, bool contig, EllPattern &E, DevPattern &D, int W, const char *what) -> int {
        // send: W x 37 doubles per node of the send lists; receive the same per ghost node
        const size_t per = (size_t)W * 37;
        local([&]() -> int {
            if ((size_t)FH.sendbuf_width < per) {
                FS_HIP(FH.sendbuf.alloc((size_t)std::max(FH.total_send, 1) * per));
                FH.sendbuf_width = (int)per;
            }
            launch_pack_ell_rows(M, contig, FH.send_nodes.p, FH.total_send, W, FH.sendbuf.p, st);
            FS_HIP(hipGetLastError());
            FS_HIP(rowbuf.alloc((size_t)std::max(n_ghost, 1) * per));
            return FEMSHELL_OK;
        });
        const int arc = agree(what);
        if (arc) return arc;
        std::string e;
        if (!comm_halo(c->comm, FH.peers, FH.send_offsets, FH.sendbuf.p, rowbuf.p, st, &e, (int)per)) return set_err(FEMSHELL_ERR_COMM, e);
        local([&]() -> int {
            const int64_t entries = (int64_t)n_ghost * W;
            FS_HIP(d_keys.alloc((size_t)std::max<int64_t>(entries, 1)));
            launch_extract_keys(rowbuf.p, entries, d_keys.p, st);
            FS_HIP(hipGetLastError());
            hkeys.assign((size_t)entries, -1);
            if (entries) FS_HIP(hipMemcpyAsync(hkeys.data(), d_keys.p, (size_t)entries * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            FS_HIP(hipStreamSynchronize(st));
            fill_ghost_rows(E, n_pad, n_ghost, W, hkeys);
            // cols / count of the ghost rows into the device copy of the pattern, then the values
            FS_HIP(hipMemcpyAsync(D.cols.p, E.cols.data(), E.cols.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            FS_HIP(hipMemcpyAsync(D.count.p, E.count.data(), E.count.size() * sizeof(uint8_t), hipMemcpyHostToDevice, st));
            launch_unpack_ell_rows(rowbuf.p, n_ghost, W, M, contig, n_pad, st);
            FS_HIP(hipGetLastError());
            FS_HIP(hipStreamSynchronize(st));
            return FEMSHELL_OK;
        });
        return FEMSHELL_OK;
    };
    rc = fetch_ghost_rows(wP, false, eP, dP, (int)Wp, "the rank's rows of the prolongator");
    if (rc) return rc;
    // (the host reads the ghost rows' keys next: agreement on their arrival)
    rc = agree("the ghost rows of the prolongator");
    if (rc) return rc;
    lap("P and its ghost rows");

    // ---- A P: per own row the union of the P rows of its neighbours
    auto p_row = [&](int32_t r, std::vector<int32_t> &out) { // keys of row r of P (own or ghost), appended
        const int s = r / kSliceNodes, nn = r % kSliceNodes;
        for (int k = 0; k < eP.count[(size_t)r]; k++) out.push_back(eP.cols[(size_t)(eP.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)]);
    };
    std::vector<int64_t> aptr;
    RawVec<int32_t> acol;
    build_rows(n, [&](int32_t a, std::vector<int32_t> &out) {
        for (int64_t q = G.ptr[(size_t)a]; q < G.ptr[(size_t)a + 1]; q++) p_row(G.col[(size_t)q], out);
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
    }, &aptr, &acol);
    ok = pack_pattern(n, aptr.data(), acol.data(), false, &eAP);
    if (!ok) (void)set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: a row of A P has more than 255 blocks");
    int64_t Wap = 0;
    rc = global_max(c, ok ? eAP.max_width : -1, &Wap, "the pattern of A P");
    if (rc) return rc;
    const int64_t own_total_AP = eAP.total();
    append_ghost_rows(eAP, n_ghost, (int)Wap);
    local([&]() -> int {
        FS_HIP(vAP.alloc((size_t)eAP.total() * 36));
        const int r = upload_pattern(eAP, dAP, vAP.p, &wAP, st);
        if (r) return r;
        EllView own = wAP;
        own.n_rows = n;
        own.n_slices = n_pad / kSliceNodes;
        own.total = own_total_AP;
        launch_amg_ap(Adev, wP, own, st);
        FS_HIP(hipGetLastError());
        return FEMSHELL_OK;
    });
    rc = fetch_ghost_rows(wAP, true, eAP, dAP, (int)Wap, "the rank's rows of A P");
    if (rc) return rc;
    rc = agree("the ghost rows of A P");
    if (rc) return rc;
    lap("A P and its ghost rows");

    // ---- R = P^T for the own aggregates: the fine rows (own and ghost, ascending local ids) that see aggregate I
    std::vector<int64_t> rptr((size_t)na + 1, 0);
    std::vector<int32_t> rrow;
    std::vector<uint8_t> rk;
    {
        auto own_key = [&](int32_t k) { return k >= key0 && k < key0 + na; };
        auto each_entry = [&](auto &&f) {
            for (int32_t r = 0; r < eP.n_rows; r++) {
                if (r >= n && r < n_pad) continue;
                const int s = r / kSliceNodes, nn = r % kSliceNodes;
                for (int k = 0; k < eP.count[(size_t)r]; k++) {
                    const int32_t kk = eP.cols[(size_t)(eP.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)];
                    if (own_key(kk)) f(r, k, kk - key0);
                }
            }
        };
        each_entry([&](int32_t, int, int32_t I) { rptr[(size_t)I + 1]++; });
        for (int32_t I = 0; I < na; I++) rptr[(size_t)I + 1] += rptr[(size_t)I];
        rrow.resize((size_t)rptr[(size_t)na]);
        rk.resize((size_t)rptr[(size_t)na]);
        std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
        each_entry([&](int32_t r, int k, int32_t I) {
            const int64_t d = fill[(size_t)I]++;
            rrow[(size_t)d] = r;
            rk[(size_t)d] = (uint8_t)k;
        });
    }
    // ---- A_c: per own aggregate the union of the A P rows of those fine rows; symmetric storage keeps the diagonal, the
    // own columns above it and every ghost column (the owner of a ghost column has its own copy of the block)
    const bool sym_coarse = res->next_dist && coarse_symmetric_storage(na);
    std::vector<int64_t> cptr;
    RawVec<int32_t> ccol;
    build_rows(na, [&](int32_t I, std::vector<int32_t> &out) {
        for (int64_t q = rptr[(size_t)I]; q < rptr[(size_t)I + 1]; q++) {
            const int32_t r = rrow[(size_t)q];
            const int s = r / kSliceNodes, nn = r % kSliceNodes;
            for (int k = 0; k < eAP.count[(size_t)r]; k++)
                out.push_back(eAP.cols[(size_t)(eAP.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)]);
        }
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
        const int32_t Ig = key0 + I;
        if (sym_coarse) out.erase(std::remove_if(out.begin(), out.end(), [&](int32_t J) { return J >= key0 && J < Ig; }), out.end());
        if (!std::binary_search(out.begin(), out.end(), Ig)) // (an aggregate without any support keeps a unit diagonal)
            out.insert(std::lower_bound(out.begin(), out.end(), Ig), Ig);
    }, &cptr, &ccol);
    ok = pack_pattern(na, cptr.data(), ccol.data(), true, &eAc, key0);
    if (!ok) (void)set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: a row of a coarse operator has more than 255 blocks");
    if (ok) {
        ok = pack_pattern(na, rptr.data(), rrow.data(), false, &eR); // (R's columns are the fine rows, ascending per aggregate)
        if (!ok) (void)set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: an aggregate is seen by more than 255 fine rows");
    }
    {
        int64_t unused = 0; // (the ranks agree before the step goes on: the exchanges below are collective)
        rc = global_max(c, ok ? 0 : -1, &unused, "the patterns of R and the coarse operator");
        if (rc) return rc;
    }
    DevBuf<int64_t> d_rptr;
    DevBuf<int32_t> d_rrow;
    DevBuf<uint8_t> d_rk;
    ValueArray h;
    L.agg.assign((size_t)n, 0);
    for (int32_t i = 0; i < n; i++) L.agg[(size_t)i] = key[(size_t)i];
    local([&]() -> int {
        std::vector<int32_t> rr(rrow);
        std::vector<uint8_t> kk(rk);
        if (rr.empty()) {
            rr.push_back(0);
            kk.push_back(0);
        }
        FS_HIP(d_rptr.upload(rptr, st));
        FS_HIP(d_rrow.upload(rr, st));
        FS_HIP(d_rk.upload(kk, st));
        FS_HIP(vR.alloc((size_t)std::max<int64_t>(eR.total(), 1) * 36));
        FS_HIP(vAc.alloc((size_t)std::max<int64_t>(eAc.total(), 1) * 36));
        int r = upload_pattern(eR, dR, vR.p, &wR, st);
        if (!r) r = upload_pattern(eAc, dAc, vAc.p, &wAc, st);
        if (r) return r;
        launch_amg_restriction(wP, d_rptr.p, d_rrow.p, d_rk.p, wR, st);
        launch_amg_galerkin(wP, wAP, d_rptr.p, d_rrow.p, d_rk.p, wAc, st, false, key0);
        FS_HIP(hipGetLastError());
        FS_HIP(hipStreamSynchronize(st));
        vAP.release();
        // ---- exports for the tests (small problems): the rank's rows with global column keys
        if (keep_host) {
            r = download_vals(vP, &h, st);
            if (r) return r;
            EllPattern own = eP; // own rows only
            own.n_rows = n;
            ell_to_bsr(own, h.data(), na_global, &L.hP);
        }
        return FEMSHELL_OK;
    });
    lap("R, A_c on the device");

    // ---- the coarse level
    N.n_global = na_global;
    N.part = cpart;
    if (!res->next_dist) {
        // all-gather: every rank gets the whole coarse operator (host BSR, global ids) and its near-null space; P keeps
        // global column ids (the correction it prolongates is replicated), R writes the rank's rows of the restricted
        // residual, which the cycle all-gathers
        Bsr mine;
        std::vector<double> cntf((size_t)na), colf, allc, allcol, allval, bc((size_t)na * 36), allb;
        local([&]() -> int {
            const int r = download_vals(vAc, &h, st);
            if (r) return r;
            ell_to_bsr(eAc, h.data(), na_global, &mine); // (columns ascending: the keys are global ids)
            colf.assign(mine.col.begin(), mine.col.end());
            for (int32_t I = 0; I < na; I++) cntf[(size_t)I] = (double)(mine.ptr[(size_t)I + 1] - mine.ptr[(size_t)I]);
            if (na) FS_HIP(hipMemcpy(bc.data(), res->Bc_dev.p, bc.size() * sizeof(double), hipMemcpyDeviceToHost));
            return FEMSHELL_OK;
        });
        rc = agree("the rank's rows of the coarse operator");
        if (rc) return rc;
        std::vector<int64_t> off;
        rc = allgather_doubles(c, cntf.data(), na, &allc, &off);
        if (!rc) rc = allgather_doubles(c, colf.data(), (int64_t)colf.size(), &allcol, &off);
        if (!rc) rc = allgather_doubles(c, mine.val.data(), (int64_t)mine.val.size(), &allval, &off);
        if (rc) return rc;
        rc = allgather_doubles(c, bc.data(), (int64_t)bc.size(), &allb, &off);
        if (rc) return rc;
        Bsr &A = res->A_global;
        A = Bsr();
        A.nr = A.nc = na_global;
        A.ptr.assign((size_t)na_global + 1, 0);
        for (int32_t I = 0; I < na_global; I++) A.ptr[(size_t)I + 1] = A.ptr[(size_t)I] + (int64_t)allc[(size_t)I];
        A.col.resize(allcol.size());
        for (size_t q = 0; q < allcol.size(); q++) A.col[q] = (int32_t)allcol[q];
        A.val.resize(allval.size());
        std::copy(allval.begin(), allval.end(), A.val.begin());
        res->B_global.swap(allb);
        res->Bc_dev.release();
        adopt(L.P, eP, dP, vP, (na_global + kSliceNodes - 1) / kSliceNodes * kSliceNodes);
        L.P.dm.n_own = n; // the cycle multiplies the own rows only
        L.P.dm.n_pad = n_pad;
        L.P.dm.n_slices = n_pad / kSliceNodes;
        adopt(L.R, eR, dR, vR, n_local);
        local([&]() -> int {
            FS_HIP(L.bown.alloc((size_t)na_pad * 6));
            FS_HIP(L.bown.zero(st));
            FS_HIP(hipStreamSynchronize(st));
            return FEMSHELL_OK;
        });
        rc = agree("the buffers of the replicated level");
        if (rc) return rc;
        lap("all-gather of the coarse operator");
        return FEMSHELL_OK;
    }
    // ---- the coarse level stays row-partitioned: local column ids, ghost lists, halo
    std::vector<int32_t> ghost_keys;
    {
        auto collect = [&](const EllPattern &E, int32_t rows) {
            for (int32_t r = 0; r < rows; r++) {
                const int s = r / kSliceNodes, nn = r % kSliceNodes;
                for (int k = 0; k < E.count[(size_t)r]; k++) {
                    const int32_t kk = E.cols[(size_t)(E.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)];
                    if (kk < key0 || kk >= key0 + na) ghost_keys.push_back(kk);
                }
            }
        };
        collect(eP, n);
        collect(eAc, na);
        std::sort(ghost_keys.begin(), ghost_keys.end());
        ghost_keys.erase(std::unique(ghost_keys.begin(), ghost_keys.end()), ghost_keys.end());
    }
    auto to_local = [&](EllPattern &E, int32_t rows) {
        for (int32_t r = 0; r < rows; r++) {
            const int s = r / kSliceNodes, nn = r % kSliceNodes;
            for (int k = 0; k < E.count[(size_t)r]; k++) {
                int32_t &kk = E.cols[(size_t)(E.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)];
                if (kk >= key0 && kk < key0 + na) kk -= key0;
                else kk = na_pad + (int32_t)(std::lower_bound(ghost_keys.begin(), ghost_keys.end(), kk) - ghost_keys.begin());
            }
        }
    };
    // (the ghost rows of P are setup intermediates: the cycle's P covers the own rows)
    to_local(eP, n);
    to_local(eAc, na);
    local([&]() -> int {
        FS_HIP(hipMemcpyAsync(dP.cols.p, eP.cols.data(), eP.cols.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        FS_HIP(hipMemcpyAsync(dAc.cols.p, eAc.cols.data(), eAc.cols.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        FS_HIP(hipStreamSynchronize(st));
        return FEMSHELL_OK;
    });
    rc = agree("the rank's rows of R and of the coarse operator"); // (build_level_halo is collective)
    if (rc) return rc;
    N.halo.reset(new LevelHalo());
    rc = build_level_halo(c, cpart, na_pad, ghost_keys, N.halo.get());
    if (rc) return rc;
    N.dist = true;
    N.n = na;
    N.n_pad = na_pad;
    N.n_ghost = (int32_t)ghost_keys.size();
    const int32_t nc_local = na_pad + N.n_ghost;
    adopt(L.P, eP, dP, vP, nc_local);
    L.P.dm.n_own = n;
    L.P.dm.n_pad = n_pad;
    L.P.dm.n_slices = n_pad / kSliceNodes;
    adopt(L.R, eR, dR, vR, n_local);
    if (keep_host) { // own rows of the coarse operator with global column ids, before the pattern arrays move on
        local([&]() -> int { return download_vals(vAc, &h, st); });
        EllPattern g = eAc;
        for (int32_t r = 0; r < na; r++) {
            const int s = r / kSliceNodes, nn = r % kSliceNodes;
            for (int k = 0; k < g.count[(size_t)r]; k++) {
                int32_t &kk = g.cols[(size_t)(g.slice_base[(size_t)s] + (int64_t)k * kSliceNodes + nn)];
                kk = kk < na_pad ? key0 + kk : ghost_keys[(size_t)(kk - na_pad)];
            }
        }
        if (!pending) ell_to_bsr(g, h.data(), na_global, &N.hA);
    }
    {
        // slices of the coarse operator that read owned columns only first: its products run those while their halo is in flight
        const int32_t ns = na_pad / kSliceNodes;
        std::vector<int32_t> interior, boundary;
        for (int32_t sl = 0; sl < ns; sl++) {
            bool ghost = false;
            for (int32_t nn = 0; nn < kSliceNodes && !ghost; nn++) {
                const int32_t r = sl * kSliceNodes + nn;
                if (r >= na) break;
                for (int k = 0; k < eAc.count[(size_t)r] && !ghost; k++)
                    ghost = eAc.cols[(size_t)(eAc.slice_base[(size_t)sl] + (int64_t)k * kSliceNodes + nn)] >= na_pad;
            }
            (ghost ? boundary : interior).push_back(sl);
        }
        N.n_interior = (int32_t)interior.size();
        interior.insert(interior.end(), boundary.begin(), boundary.end());
        local([&]() -> int {
            FS_HIP(N.order.upload(interior, st));
            return FEMSHELL_OK;
        });
    }
    adopt(N.A, eAc, dAc, vAc, nc_local);
    N.A.dm.n_ghost = N.n_ghost;
    N.pattern = HostEllPattern();
    N.pattern.n = na;
    N.pattern.symmetric = sym_coarse;
    if (sym_coarse) {
        SlicedEllSym S;
        build_in_lists(na, eAc.slice_width, eAc.slice_base, eAc.cols.data(), eAc.count, &S);
        local([&]() -> int { return attach_in_lists(N.A, S, eAc.total(), st); });
        N.pattern.in_width.swap(S.in_width);
        N.pattern.in_base.swap(S.in_base);
        N.pattern.in_slots.swap(S.in_slots);
        N.pattern.in_rows.swap(S.in_rows);
    }
    N.pattern.slice_width.swap(eAc.slice_width);
    N.pattern.slice_base.swap(eAc.slice_base);
    N.pattern.cols.swap(eAc.cols);
    N.pattern.count.swap(eAc.count);
    N.A_on_device = true;
    rc = agree("the coarse level's halo and in-lists"); // (the caller's next step is collective again)
    if (rc) return rc;
    lap("coarse halo");
    return FEMSHELL_OK;
}

// lambda_max(D^-1 A) of a row-partitioned level: the power iteration of amg_solve.cpp with the halo exchange in front of
// every product and the two last norms summed over the ranks
int power_iteration_dist(femshell_ctx *c, AmgLevel &L, const DeviceMatrix &A, int iterations, double *lam_out)
{
    hipStream_t st = c->stream;
    const int G = slice_grid(A);
    DevBuf<double> part;
    FS_HIP(part.alloc(2 * (size_t)G + 2));
    double *x = L.d.p, *z = L.r.p;
    launch_fill_hash(x, 6ll * L.n, 6ll * L.n_pad, st);
    if (iterations < 2) iterations = 2;
    for (int it = 0; it < iterations; it++) {
        int rc = level_halo_exchange(c, *L.halo, x, 6, st);
        if (rc) return rc;
        launch_spmv(A, x, L.q.p, nullptr, nullptr, st);
        launch_minv_apply_norm(A, L.q.p, z, part.p + (size_t)(it & 1) * G, st);
        if (L.patches) { // the level's smoother applies the cluster blocks: lambda_max of THAT operator (amg_solve.cpp)
            launch_patch_correct(L.patches->view(), L.q.p, 1.0, z, false, nullptr, nullptr, st);
            launch_sqnorm_partials(z, 6ll * L.n_pad, part.p + (size_t)(it & 1) * G, G, st);
        }
        std::swap(x, z);
    }
    FS_HIP(hipGetLastError());
    // the rank's two norms: sums over its G partials in index order, taken on the device (launch_sums_in_order -- the host's loop of
    // rounds 4-5, same bits, without the megabyte that went to the host and the sixteen bytes that came back): stage[j] = the sum
    // over part[j G, (j + 1) G); all-reduced as they lie, which of the two is the last norm is sorted out afterwards
    double s[2] = {0.0, 0.0};
    double *stage = part.p + 2 * (size_t)G;
    launch_sums_in_order(part.p, G, stage, st);
    FS_HIP(hipGetLastError());
    std::string e;
    if (!comm_allreduce_sum(c->comm, stage, 2, st, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    FS_HIP(hipMemcpyAsync(s, stage, sizeof s, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    const double n_last = std::sqrt(s[(iterations - 1) & 1]), n_prev = std::sqrt(s[(iterations - 2) & 1]);
    if (!(n_prev > 0.0) || !std::isfinite(n_last) || !(n_last > 0.0))
        return set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: power iteration broke down");
    *lam_out = n_last / n_prev;
    return FEMSHELL_OK;
}

double operator_bytes(const AmgOperator &op)
{
    return (double)op.vals.n * 8.0 + (double)op.cols.n * 4.0 + (double)op.tbuf.n * 8.0 + (double)op.in_slots.n * 8.0;
}

} // namespace

int32_t amg_dist_min()
{
    const char *e = getenv("FEMSHELL_AMG_DIST_MIN"); // read per setup: the tests switch it inside one process
    return e ? (int32_t)atol(e) : (int32_t)60000;
}

int level_halo_exchange(femshell_ctx *c, LevelHalo &H, double *vec, int width, hipStream_t st)
{
    if (!c->comm.active() || c->comm.world == 1) return FEMSHELL_OK; // (a rank without neighbours still joins the group)
    if (H.sendbuf_width < width) {
        FS_HIP(hipStreamSynchronize(st));
        FS_HIP(H.sendbuf.alloc((size_t)std::max(H.total_send, 1) * (size_t)width));
        H.sendbuf_width = width;
    }
    launch_pack(vec, H.send_nodes.p, H.total_send, H.sendbuf.p, st, width);
    std::string e;
    if (!comm_halo(c->comm, H.peers, H.send_offsets, H.sendbuf.p, vec + (int64_t)width * H.n_pad, st, &e, width))
        return set_err(FEMSHELL_ERR_COMM, e);
    return FEMSHELL_OK;
}

int amg_setup_dist(femshell_ctx *c)
{
    TraceRange trace("femshell multigrid setup (row-partitioned)");
    const double t0 = now_s();
    DevPool::Defer no_sync_per_free; // (buffers released during the setup join the pool at its end: context.hpp)
    hipStream_t st = c->stream;
    const femshell_pc_options opt = c->pc;
    const bool kcycle = opt.cycle == FEMSHELL_CYCLE_K;
    const Plan &pl = c->plan;
    if (opt.max_levels < 2) return set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid on a row-partitioned context needs at least two levels");
    if ((int)pl.part_bounds.size() != c->comm.world + 1)
        return set_err(FEMSHELL_ERR_INVALID, "multigrid setup: the plan holds no row partition");
    c->amg.reset(new Amg());
    Amg &H = *c->amg;
    H.opt = opt;
    H.dist.reset(new AmgDist());
    static const bool verbose = getenv("FEMSHELL_AMG_VERBOSE") && atoi(getenv("FEMSHELL_AMG_VERBOSE")) != 0;
    double tl = now_s();
    int lap_level = 0;
    auto lap = [&](const char *what) {
        const double t = now_s();
        if (verbose) fprintf(stderr, "[femshell amg setup, rank %d] level %d %-32s %.3f s\n", c->comm.rank, lap_level, what, t - tl);
        tl = t;
        CommWatch::heartbeat(); // (progress of the phase the watchdog times)
    };
    const bool keep_host = amg_keep_host(pl.nnz_blocks); // inspection exports (tests)
    static const bool plain = getenv("FEMSHELL_AMG_PLAIN_RBM") && atoi(getenv("FEMSHELL_AMG_PLAIN_RBM")) != 0;

    // ---- level 0: the rank's rows of K
    H.levels.emplace_back(new AmgLevel());
    {
        AmgLevel &L0 = *H.levels.back();
        L0.dist = true;
        L0.n = pl.n_own;
        L0.n_pad = pl.n_pad;
        L0.n_ghost = pl.n_ghost;
        L0.n_global = pl.n_nodes;
        L0.part = pl.part_bounds;
        L0.nnzb = pl.nnz_blocks;
        L0.halo.reset(new LevelHalo());
        LevelHalo &h = *L0.halo;
        h.n_pad = pl.n_pad;
        h.n_ghost = pl.n_ghost;
        h.peers = pl.peers;
        std::vector<int32_t> flat;
        for (const HaloPeer &p : h.peers) {
            h.send_offsets.push_back((int32_t)flat.size());
            flat.insert(flat.end(), p.send_nodes.begin(), p.send_nodes.end());
        }
        h.total_send = (int32_t)flat.size();
        if (flat.empty()) flat.push_back(0);
        FS_HIP(h.send_nodes.upload(flat, st));
        FS_HIP(h.sendbuf.alloc((size_t)std::max(h.total_send, 1) * 6));
        h.sendbuf_width = 6;
        FS_HIP(hipStreamSynchronize(st));
    }
    DevBuf<double> d_normals, Bdev;
    NearNullSrc src;
    {
        // near-null space of the finest level from the mesh in HBM: rigid-body modes about the centre of the WHOLE mesh (the
        // same modes on every rank), rotations projected onto the tangent planes of the own nodes (every element touching an
        // owned node is local, so their normals are the single-rank ones; ghost rows of Q come from their owners)
        src.xyz = c->xyz.p;
        src.dmask = c->dmask.p;
        if (!plain) {
            // (from the gather lists of the diagonal slots -- amg_setup.cpp node_normals_plan, the bits of node_normals -- and through
            //  the pinned staging buffer: the host array is freed right below, context.hpp stage_host has why that matters)
            RawVec<double> normals;
            node_normals_plan(pl, &normals);
            FS_HIP(d_normals.alloc((size_t)pl.n_local_nodes() * 3));
            FS_HIP(d_normals.zero(st)); // (ghost rows: their normals are never read, their Q rows come from the owners)
            {
                const int rcu = staged_upload(c, d_normals.p, normals.data(), normals.size() * sizeof(double), st);
                if (rcu) return rcu;
            }
            src.normals = d_normals.p;
        }
        double ctr[3];
        if (c->have_mesh_centre) std::copy(c->mesh_centre, c->mesh_centre + 3, ctr);
        else if (c->comm.world == 1) mesh_centre(pl.n_own, pl.xyz_local.data(), ctr); // (a one-rank communicator: tests)
        else return set_err(FEMSHELL_ERR_INVALID, "multigrid setup: the context does not know the centre of the mesh");
        src.cx = ctr[0];
        src.cy = ctr[1];
        src.cz = ctr[2];
    }
    HostEllPattern pat0;
    pattern_of_plan(pl, &pat0);
    Bsr A;
    std::vector<double> B;
    int first_replicated = 0;
    for (int l = 0;; l++) {
        lap_level = l;
        AmgLevel &L = *H.levels[(size_t)l];
        const DeviceMatrix &Adev = l == 0 ? c->dm : L.A.dm;
        if (l > 0) {
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
            const bool bad = status_now != 0; // (seen by the rank that owns the block only: the ranks agree below)
            if (bad) {
                {
                    const int rcc = clear_status_word(c, st);
                    if (rcc) return rcc;
                }
                (void)set_err(FEMSHELL_ERR_BREAKDOWN, "multigrid setup: a diagonal block of coarse level " + std::to_string(l) +
                                                          " is not positive definite");
            }
            int64_t unused = 0;
            const int arc = global_max(c, bad ? -1 : 0, &unused, "the block-Jacobi inverse of a coarse level");
            if (arc) return bad ? FEMSHELL_ERR_BREAKDOWN : arc;
            int64_t stored = 0;
            for (int32_t a = 0; a < L.n; a++) stored += L.pattern.count[(size_t)a];
            L.nnzb = stored;
            L.A.nnzb = stored;
        }
        int rc = alloc_level_vectors(L, l == 0, kcycle, st);
        if (rc) return rc;
        if (l == 0) {
            // clusters of rigidly coupled nodes (amg_patch.hpp): every rank among its own rows, the ranks decide together whether
            // the mesh is one that needs them
            std::vector<uint8_t> read_by_others((size_t)L.n, 0);
            for (const HaloPeer &pr : L.halo->peers)
                for (int32_t v : pr.send_nodes)
                    if (v >= 0 && v < L.n) read_by_others[(size_t)v] = 1;
            rc = amg_build_patches(c, Adev, L, true, &read_by_others);
            int64_t unused = 0;
            const int arc = global_max(c, rc ? -1 : 0, &unused, "the clusters of the patch smoother");
            if (arc) return rc ? rc : arc;
            lap("patch smoother: clusters");
        }
        double lam = 0.0;
        rc = power_iteration_dist(c, L, Adev, amg_power_iterations(), &lam);
        if (rc) return rc;
        L.lam = amg_lambda_safety() * lam;
        lap("block-Jacobi, power iteration");
        H.levels.emplace_back(new AmgLevel());
        AmgLevel &N = *H.levels.back();
        StepResult res;
        NearNullSrc bsrc = src;
        if (l > 0) {
            bsrc = NearNullSrc();
            bsrc.B = Bdev.p;
        }
        rc = dist_coarsen(c, opt, l, Adev, l == 0 ? pat0 : L.pattern, L, N, bsrc, keep_host, opt.max_levels, &res, lap);
        if (rc) return rc;
        if (l == 0) pat0 = HostEllPattern();
        else L.pattern = HostEllPattern();
        if (!res.next_dist) {
            A = std::move(res.A_global);
            B = std::move(res.B_global);
            first_replicated = l + 1;
            break;
        }
        std::swap(Bdev.p, res.Bc_dev.p);
        std::swap(Bdev.n, res.Bc_dev.n);
    }
    H.dist->dist_levels = first_replicated;
    FS_HIP(H.dist->x0.alloc((size_t)(pl.n_pad + pl.n_ghost) * 6));
    FS_HIP(H.dist->x0.zero(st));
    Bdev.release();
    DevBuf<double> none;
    int rc = amg_finish_hierarchy(c, A, B, none, first_replicated);
    if (rc) return rc;
    // HBM of the rank's operators: what shrinks with the rank count and what every rank holds in full
    double part_b = 0.0, repl_b = 0.0;
    for (size_t l = 0; l < H.levels.size(); l++) {
        const AmgLevel &L = *H.levels[l];
        const double b = operator_bytes(L.A) + operator_bytes(L.P) + operator_bytes(L.R) + (double)L.minv.n * 8.0;
        ((int)l < first_replicated ? part_b : repl_b) += b;
    }
    repl_b += (double)H.coarse_inv.n * 8.0 + (double)H.coarse_inv32.n * 4.0;
    H.dist->hierarchy_bytes_partitioned = part_b;
    H.dist->hierarchy_bytes_replicated = repl_b;
    H.setup_seconds = now_s() - t0;
    return FEMSHELL_OK;
}

} // namespace femshell
