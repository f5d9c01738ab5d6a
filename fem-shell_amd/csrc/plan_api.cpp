// plan_api.cpp -- host-only inspection of the symbolic phase (include/femshell_plan.h); touches no GPU state.
#include "femshell_plan.h"

#include <climits>
#include <cstring>
#include <string>
#include <vector>

#include "amg.hpp"
#include "errors.hpp"
#include "plan.hpp"
#include "reorder.hpp"

using namespace femshell;

struct femshell_plan {
    Plan p;
};

extern "C" {

int femshell_plan_create(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                         const int32_t *quad, int32_t rank, int32_t world_size, femshell_plan **out)
{
    if (!out || !xyz) return set_err(FEMSHELL_ERR_INVALID, "femshell_plan_create: null argument");
    *out = nullptr;
    femshell_plan *pl = new femshell_plan();
    std::string e;
    if (!build_plan(n_nodes, xyz, n_tri, tri, n_quad, quad, rank, world_size, &pl->p, &e, default_symmetric_storage())) {
        delete pl;
        return set_err(FEMSHELL_ERR_MESH, "femshell_plan_create: " + e);
    }
    *out = pl;
    return FEMSHELL_OK;
}

void femshell_plan_destroy(femshell_plan *plan) { delete plan; }

int femshell_plan_info(const femshell_plan *plan, int64_t *info)
{
    if (!plan || !info) return set_err(FEMSHELL_ERR_INVALID, "femshell_plan_info: null argument");
    const Plan &p = plan->p;
    info[FEMSHELL_PLAN_N_OWN] = p.n_own;
    info[FEMSHELL_PLAN_N_PAD] = p.n_pad;
    info[FEMSHELL_PLAN_N_GHOST] = p.n_ghost;
    info[FEMSHELL_PLAN_N_SLICES] = p.n_slices;
    info[FEMSHELL_PLAN_N_LTRI] = p.n_ltri();
    info[FEMSHELL_PLAN_N_LQUAD] = p.n_lquad();
    info[FEMSHELL_PLAN_TOTAL_SLOTS] = p.total_slots();
    info[FEMSHELL_PLAN_N_PAIRS] = (int64_t)p.pairs.size();
    info[FEMSHELL_PLAN_N_PEERS] = (int64_t)p.peers.size();
    info[FEMSHELL_PLAN_ROW_BEGIN] = p.row_begin;
    info[FEMSHELL_PLAN_ROW_END] = p.row_end;
    info[FEMSHELL_PLAN_NNZ_BLOCKS] = p.nnz_blocks;
    info[FEMSHELL_PLAN_N_INTERIOR_SLICES] = p.n_interior_slices;
    info[FEMSHELL_PLAN_N_ITEMS] = (int64_t)p.items.size();
    int64_t multi = 0;
    for (int32_t s = 0; s < p.n_slices; s++) multi += (p.item_ptr[s + 1] - p.item_ptr[s]) > (p.pipe ? 192 : 256);
    info[FEMSHELL_PLAN_N_MULTI_ROUND_SLICES] = multi;
    info[FEMSHELL_PLAN_MAX_SLICE_ELEMS] = p.max_slice_elems;
    info[FEMSHELL_PLAN_MAX_SLICE_WIDTH] = p.max_slice_width;
    info[FEMSHELL_PLAN_SYMMETRIC] = p.symmetric ? 1 : 0;
    info[FEMSHELL_PLAN_STORED_BLOCKS] = p.stored_blocks;
    info[FEMSHELL_PLAN_PIPE] = p.pipe ? 1 : 0;
    return FEMSHELL_OK;
}

int64_t femshell_plan_array(const femshell_plan *plan, int which, void *out)
{
    if (!plan) return -1;
    const Plan &p = plan->p;
    auto give = [&](const auto &v) -> int64_t {
        if (out && !v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(v[0]));
        return (int64_t)v.size();
    };
    std::vector<int32_t> tmp;
    switch (which) {
    case FEMSHELL_PLAN_GHOST_GLOBAL: return give(p.ghost_global);
    case FEMSHELL_PLAN_TRI_LOCAL: return give(p.tri_local);
    case FEMSHELL_PLAN_TRI_GLOBAL_ID: return give(p.tri_global_id);
    case FEMSHELL_PLAN_QUAD_LOCAL: return give(p.quad_local);
    case FEMSHELL_PLAN_QUAD_GLOBAL_ID: return give(p.quad_global_id);
    case FEMSHELL_PLAN_SLICE_WIDTH: return give(p.slice_width);
    case FEMSHELL_PLAN_SLICE_BASE: return give(p.slice_base);
    case FEMSHELL_PLAN_COLS: return give(p.cols);
    case FEMSHELL_PLAN_PAIR_PTR: return give(p.pair_ptr);
    case FEMSHELL_PLAN_PAIRS: return give(p.pairs);
    case FEMSHELL_PLAN_XYZ_LOCAL: return give(p.xyz_local);
    case FEMSHELL_PLAN_SPMV_ORDER: return give(p.spmv_order);
    case FEMSHELL_PLAN_IN_WIDTH: return give(p.in_width);
    case FEMSHELL_PLAN_IN_BASE: return give(p.in_base);
    case FEMSHELL_PLAN_IN_SLOTS: return give(p.in_slots);
    case FEMSHELL_PLAN_GAT_SLOTS: return give(p.gat_slots);
    case FEMSHELL_PLAN_LOC_LIST: return give(p.loc_list);
    case FEMSHELL_PLAN_LOC_INDEX: return give(p.loc_index);
    case FEMSHELL_PLAN_ITEM_PTR: return give(p.item_ptr);
    case FEMSHELL_PLAN_ITEMS:
        if (out && !p.items.empty()) std::memcpy(out, p.items.data(), p.items.size() * sizeof(Plan::Item));
        return (int64_t)p.items.size() * 4;
    case FEMSHELL_PLAN_PAIRS16: return give(p.pairs16);
    case FEMSHELL_PLAN_SLICE_ELEM_PTR: return give(p.slice_elem_ptr);
    case FEMSHELL_PLAN_PEER_RANKS:
        for (auto &h : p.peers) tmp.push_back(h.rank);
        return give(tmp);
    case FEMSHELL_PLAN_PEER_RECV_OFFSET:
        for (auto &h : p.peers) tmp.push_back(h.recv_offset);
        return give(tmp);
    case FEMSHELL_PLAN_PEER_RECV_COUNT:
        for (auto &h : p.peers) tmp.push_back(h.recv_count);
        return give(tmp);
    case FEMSHELL_PLAN_PEER_SEND_PTR:
        tmp.push_back(0);
        for (auto &h : p.peers) tmp.push_back(tmp.back() + (int32_t)h.send_nodes.size());
        return give(tmp);
    case FEMSHELL_PLAN_PEER_SEND_NODES:
        for (auto &h : p.peers) tmp.insert(tmp.end(), h.send_nodes.begin(), h.send_nodes.end());
        return give(tmp);
    default: return -1;
    }
}


// ---- host-only pieces of the multigrid setup ---------------------------------------------------------------

struct femshell_amg_coarsening {
    std::vector<int32_t> agg;
    Bsr P, Ac;
    std::vector<double> Bc;
};

namespace {
// false: the pattern is not what every routine behind these entry points assumes -- row pointers ascending from 0, the columns
// of a row strictly ascending and inside [0, n_cols) (merge intersections and binary searches walk them: graph_for_aggregation)
bool pattern_ok(int32_t n, const int32_t *rowptr, const int32_t *colidx, int32_t n_cols)
{
    if (rowptr[0] != 0) return false;
    for (int32_t i = 0; i < n; i++) {
        if (rowptr[i + 1] < rowptr[i]) return false;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; q++)
            if (colidx[q] < 0 || (n_cols != INT32_MAX && colidx[q] >= n_cols) || (q > rowptr[i] && colidx[q] <= colidx[q - 1])) return false;
    }
    return true;
}
bool to_bsr(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *vals, Bsr *A, int32_t n_cols = -1)
{
    if (!pattern_ok(n, rowptr, colidx, n_cols < 0 ? n : n_cols)) return false;
    A->nr = A->nc = n;
    A->ptr.assign(rowptr, rowptr + n + 1);
    A->col.assign(colidx, colidx + rowptr[n]);
    A->val.assign(vals, vals + 36ll * rowptr[n]);
    return true;
}
} // namespace

int femshell_amg_host_rbm(int32_t n_nodes, const double *xyz, const uint8_t *dmask, double *B_out)
{
    if (n_nodes <= 0 || !xyz || !B_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_rbm: invalid argument");
    std::vector<double> B;
    rigid_body_modes(n_nodes, xyz, dmask, &B);
    std::memcpy(B_out, B.data(), B.size() * sizeof(double));
    return FEMSHELL_OK;
}

int femshell_amg_host_coarsen(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                              const double *B, double lambda_max, femshell_amg_coarsening **out)
{
    if (n_nodes <= 0 || !rowptr || !colidx || !vals || !B || !out || !(lambda_max > 0.0))
        return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_coarsen: invalid argument");
    Bsr A;
    if (!to_bsr(n_nodes, rowptr, colidx, vals, &A))
        return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_coarsen: rows of the block CSR pattern must hold strictly ascending columns in [0, n_nodes)");
    femshell_amg_coarsening *h = new femshell_amg_coarsening();
    const int32_t na = aggregate_nodes(A, &h->agg);
    std::vector<double> Bv(B, B + 36ll * n_nodes), Q, Dinv;
    tentative_prolongator(h->agg, na, Bv, &Q, &h->Bc);
    block_diagonal_inverse(A, &Dinv);
    smoothed_prolongator(A, Dinv, h->agg, na, Q, (4.0 / 3.0) / lambda_max, &h->P);
    Bsr R;
    galerkin_product(A, h->P, &R, &h->Ac);
    *out = h;
    return FEMSHELL_OK;
}

int32_t femshell_amg_host_aggregate(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const int32_t *visit,
                                    int32_t *agg_out)
{
    if (n_nodes <= 0 || !rowptr || !colidx || !agg_out) {
        set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_aggregate: invalid argument");
        return -1;
    }
    if (!pattern_ok(n_nodes, rowptr, colidx, n_nodes)) {
        set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_aggregate: rows of the graph must hold strictly ascending columns in [0, n_nodes)");
        return -1;
    }
    Bsr A;
    A.nr = A.nc = n_nodes;
    A.ptr.assign(rowptr, rowptr + n_nodes + 1);
    A.col.assign(colidx, colidx + rowptr[n_nodes]);
    std::vector<int32_t> agg, order;
    if (visit) {
        order.assign(visit, visit + n_nodes);
        std::vector<char> seen((size_t)n_nodes, 0);
        for (int32_t v : order) {
            if (v < 0 || v >= n_nodes || seen[(size_t)v]) {
                set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_aggregate: visit is not a permutation of the nodes");
                return -1;
            }
            seen[(size_t)v] = 1;
        }
    }
    const int32_t na = aggregate_nodes(A, &agg, visit ? &order : nullptr);
    std::memcpy(agg_out, agg.data(), (size_t)n_nodes * sizeof(int32_t));
    return na;
}

int32_t femshell_amg_host_patch_clusters(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals, double tau,
                                         int32_t max_nodes, int32_t *labels_out, int64_t *edges_out)
{
    if (n_nodes <= 0 || !rowptr || !colidx || !vals || !labels_out || !(tau > 0.0)) {
        set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_patch_clusters: invalid argument");
        return -1;
    }
    Bsr A;
    if (!to_bsr(n_nodes, rowptr, colidx, vals, &A)) {
        set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_patch_clusters: rows of the block CSR pattern must hold strictly ascending columns in [0, n_nodes)");
        return -1;
    }
    std::vector<double> Dinv;
    block_diagonal_inverse(A, &Dinv);
    std::vector<PatchEdge> edges;
    patch_edges_host(A, Dinv, tau, &edges);
    if (edges_out) *edges_out = (int64_t)edges.size();
    std::vector<int32_t> label, ptr, nodes;
    const int32_t nc = patch_clusters(n_nodes, std::move(edges), max_nodes, &label, &ptr, &nodes);
    std::memcpy(labels_out, label.data(), (size_t)n_nodes * sizeof(int32_t));
    return nc;
}

int32_t femshell_amg_host_aggregate_glued(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const int32_t *labels,
                                          const int32_t *visit, int32_t *agg_out)
{
    if (n_nodes <= 0 || !rowptr || !colidx || !labels || !agg_out || !pattern_ok(n_nodes, rowptr, colidx, n_nodes)) {
        set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_aggregate_glued: invalid argument");
        return -1;
    }
    Bsr A;
    A.nr = A.nc = n_nodes;
    A.ptr.assign(rowptr, rowptr + n_nodes + 1);
    A.col.assign(colidx, colidx + rowptr[n_nodes]);
    std::vector<int32_t> lab(labels, labels + n_nodes), agg, order;
    if (visit) order.assign(visit, visit + n_nodes);
    const int32_t na = aggregate_nodes_glued(A, lab, &agg, visit ? &order : nullptr);
    std::memcpy(agg_out, agg.data(), (size_t)n_nodes * sizeof(int32_t));
    return na;
}

void femshell_amg_coarsening_destroy(femshell_amg_coarsening *h) { delete h; }

int femshell_plan_node_normals(const femshell_plan *plan, int32_t from_gather_lists, double *normals_out)
{
    if (!plan || !normals_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_plan_node_normals: invalid argument");
    const Plan &p = plan->p;
    if (from_gather_lists) {
        RawVec<double> N;
        node_normals_plan(p, &N);
        std::memcpy(normals_out, N.data(), N.size() * sizeof(double));
    } else {
        std::vector<double> N;
        node_normals(p.n_own, p.xyz_local.data(), p.n_ltri(), p.tri_local.data(), p.n_lquad(), p.quad_local.data(), &N);
        std::memcpy(normals_out, N.data(), N.size() * sizeof(double));
    }
    return FEMSHELL_OK;
}

int64_t femshell_amg_coarsening_array(const femshell_amg_coarsening *h, int which, void *out)
{
    if (!h) return -1;
    auto give = [&](const void *src, size_t count, size_t elem) -> int64_t {
        if (out && count) std::memcpy(out, src, count * elem);
        return (int64_t)count;
    };
    switch (which) {
    case FEMSHELL_COARSEN_AGG: return give(h->agg.data(), h->agg.size(), sizeof(int32_t));
    case FEMSHELL_COARSEN_P_ROWPTR: return give(h->P.ptr.data(), h->P.ptr.size(), sizeof(int64_t));
    case FEMSHELL_COARSEN_P_COLS: return give(h->P.col.data(), h->P.col.size(), sizeof(int32_t));
    case FEMSHELL_COARSEN_P_VALS: return give(h->P.val.data(), h->P.val.size(), sizeof(double));
    case FEMSHELL_COARSEN_AC_ROWPTR: return give(h->Ac.ptr.data(), h->Ac.ptr.size(), sizeof(int64_t));
    case FEMSHELL_COARSEN_AC_COLS: return give(h->Ac.col.data(), h->Ac.col.size(), sizeof(int32_t));
    case FEMSHELL_COARSEN_AC_VALS: return give(h->Ac.val.data(), h->Ac.val.size(), sizeof(double));
    case FEMSHELL_COARSEN_BC: return give(h->Bc.data(), h->Bc.size(), sizeof(double));
    default: return -1;
    }
}

int femshell_amg_host_dense_inverse(int32_t n_nodes, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                    double *inv_out)
{
    if (n_nodes <= 0 || !rowptr || !colidx || !vals || !inv_out)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_dense_inverse: invalid argument");
    Bsr A;
    if (!to_bsr(n_nodes, rowptr, colidx, vals, &A))
        return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_host_dense_inverse: rows of the block CSR pattern must hold strictly ascending columns in [0, n_nodes)");
    std::vector<double> inv;
    if (!dense_inverse(A, &inv)) return set_err(FEMSHELL_ERR_BREAKDOWN, "femshell_amg_host_dense_inverse: matrix is not positive definite");
    std::memcpy(inv_out, inv.data(), inv.size() * sizeof(double));
    return FEMSHELL_OK;
}

int64_t femshell_amg_host_pack(int32_t n_rows, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                               int32_t diag_first, int32_t *slice_width, int64_t *slice_base, int32_t *cols, double *ell_vals)
{
    if (n_rows <= 0 || !rowptr || !colidx || !vals) return -1;
    Bsr A;
    if (!to_bsr(n_rows, rowptr, colidx, vals, &A, INT32_MAX)) return -1; // (rectangular operators: any column >= 0, ascending per row)
    SlicedEll S;
    pack_sliced_ell(A, diag_first != 0, &S);
    if (slice_width) std::memcpy(slice_width, S.slice_width.data(), S.slice_width.size() * sizeof(int32_t));
    if (slice_base) std::memcpy(slice_base, S.slice_base.data(), S.slice_base.size() * sizeof(int64_t));
    if (cols) std::memcpy(cols, S.cols.data(), S.cols.size() * sizeof(int32_t));
    if (ell_vals) std::memcpy(ell_vals, S.vals.data(), S.vals.size() * sizeof(double));
    return S.slice_base.back();
}

int64_t femshell_amg_host_pack_sym(int32_t n_rows, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                   int32_t *slice_width, int64_t *slice_base, int32_t *cols, double *ell_vals,
                                   int32_t *in_width, int64_t *in_base, int32_t *in_slots, int32_t *in_rows, int64_t *in_total)
{
    if (n_rows <= 0 || !rowptr || !colidx || !vals) return -1;
    Bsr A;
    if (!to_bsr(n_rows, rowptr, colidx, vals, &A, INT32_MAX)) return -1; // (rectangular operators: any column >= 0, ascending per row)
    SlicedEllSym S;
    pack_sliced_ell_sym(A, &S);
    if (slice_width) std::memcpy(slice_width, S.slice_width.data(), S.slice_width.size() * sizeof(int32_t));
    if (slice_base) std::memcpy(slice_base, S.slice_base.data(), S.slice_base.size() * sizeof(int64_t));
    if (cols) std::memcpy(cols, S.cols.data(), S.cols.size() * sizeof(int32_t));
    if (ell_vals) std::memcpy(ell_vals, S.vals.data(), S.vals.size() * sizeof(double));
    if (in_width) std::memcpy(in_width, S.in_width.data(), S.in_width.size() * sizeof(int32_t));
    if (in_base) std::memcpy(in_base, S.in_base.data(), S.in_base.size() * sizeof(int64_t));
    if (in_slots) std::memcpy(in_slots, S.in_slots.data(), S.in_slots.size() * sizeof(int32_t));
    if (in_rows) std::memcpy(in_rows, S.in_rows.data(), S.in_rows.size() * sizeof(int32_t));
    if (in_total) *in_total = (int64_t)S.in_slots.size();
    return S.slice_base.back();
}

int femshell_reorder_host(int32_t kind, int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                          int32_t n_quad, const int32_t *quad, int32_t *perm_out)
{
    if (n_nodes <= 0 || !xyz || !perm_out || (n_tri > 0 && !tri) || (n_quad > 0 && !quad)) return -1;
    for (int64_t q = 0; q < 3ll * n_tri; q++)
        if (tri[q] < 0 || tri[q] >= n_nodes) return -1;
    for (int64_t q = 0; q < 4ll * n_quad; q++)
        if (quad[q] < 0 || quad[q] >= n_nodes) return -1;
    std::vector<int32_t> perm;
    if (kind == 1) rcm_order(n_nodes, n_tri, tri, n_quad, quad, &perm);
    else morton_order(n_nodes, xyz, &perm);
    std::memcpy(perm_out, perm.data(), perm.size() * sizeof(int32_t));
    return 0;
}

} // extern "C"
