// amg_device_setup.cpp -- coarsening steps of the multigrid setup with the numerics on the device (every level above
// FEMSHELL_AMG_DEVICE_MIN nodes, amg_solve.cpp).
//
// The finest level dominates the setup: K has 14M blocks on the 4M-triangle meshes, and the host path (amg_setup.cpp)
// first has to bring it over PCIe and mirror it.  Here the host only does integer work on the block graph the plan (or
// the previous step) already holds -- aggregation, the grouping of the nodes by aggregate, the patterns of P, A P,
// R = P^T and A_c = P^T A P, and the index lists that tell every block of a result which blocks feed it -- and the device
// computes the values from the operator where it lies (amg_kernels.hip: k_amg_tentative_qr, k_amg_prolongator, k_amg_ap,
// k_amg_restriction, k_amg_galerkin), directly in the sliced block ELL layout the cycle multiplies with.  The near-null
// space never exists as a host array: generated from the mesh on the finest level, the R factors of the previous step
// below.  A coarse operator travels back only when the next step runs on the host.  FEMSHELL_AMG_SETUP=host keeps
// everything on the host (the path tests compare this one with).
#include "amg_device.hpp"
#include "amg_pattern.hpp"
#include "amg_symbolic.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>

namespace femshell {

// rows as sorted lists -> sliced ELL pattern (32 rows per slice, `count` real entries per row, padding columns 0;
// diag_first: the entry equal to the row index is moved to slot 0)
bool pack_pattern(int32_t n_rows, const int64_t *ptr, const int32_t *col, bool diag_first, EllPattern *out, int32_t diag_key)
{
    EllPattern &E = *out;
    E = EllPattern();
    E.n_rows = n_rows;
    E.n_pad = (n_rows + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    E.n_slices = E.n_pad / kSliceNodes;
    E.slice_width.assign((size_t)E.n_slices, 1);
    E.slice_base.assign((size_t)E.n_slices + 1, 0);
    E.count.assign((size_t)E.n_pad, 0);
    std::atomic<int> too_wide{0};
    parallel_chunks(E.n_slices, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; s++) {
            int w = 1;
            for (int n = 0; n < kSliceNodes; n++) {
                const int32_t a = (int32_t)s * kSliceNodes + n;
                if (a >= n_rows) continue;
                const int64_t c = ptr[a + 1] - ptr[a];
                if (c > 255) {
                    too_wide.store(1);
                    return;
                }
                E.count[(size_t)a] = (uint8_t)c;
                w = std::max<int>(w, (int)c);
            }
            E.slice_width[(size_t)s] = w;
        }
    }, 64);
    if (too_wide.load()) return false;
    for (int32_t s = 0; s < E.n_slices; s++) {
        E.max_width = std::max(E.max_width, E.slice_width[(size_t)s]);
        E.slice_base[(size_t)s + 1] = E.slice_base[(size_t)s] + (int64_t)E.slice_width[(size_t)s] * kSliceNodes;
    }
    E.nnzb = ptr[n_rows];
    E.cols.resize((size_t)E.total()); // (padding columns 0: every slice is cleared by the thread that fills it)
    parallel_chunks(E.n_slices, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; s++) {
            std::fill(E.cols.begin() + E.slice_base[(size_t)s], E.cols.begin() + E.slice_base[(size_t)s + 1], 0);
            for (int n = 0; n < kSliceNodes; n++) {
                const int32_t a = (int32_t)s * kSliceNodes + n;
                if (a >= n_rows) continue;
                int k = 0;
                if (diag_first) {
                    E.cols[(size_t)(E.slice_base[s] + n)] = a + diag_key;
                    k = 1;
                }
                for (int64_t q = ptr[a]; q < ptr[a + 1]; q++) {
                    if (diag_first && col[q] == a + diag_key) continue;
                    E.cols[(size_t)(E.slice_base[s] + (int64_t)k * kSliceNodes + n)] = col[q];
                    k++;
                }
            }
        }
    }, 64);
    return true;
}

int upload_pattern(const EllPattern &E, DevPattern &D, double *vals, EllView *view, hipStream_t st)
{
    FS_HIP(D.slice_width.upload(E.slice_width, st));
    FS_HIP(D.slice_base.upload(E.slice_base, st));
    FS_HIP(D.cols.upload(E.cols, st));
    FS_HIP(D.count.upload(E.count, st));
    view->n_rows = E.n_rows;
    view->n_slices = E.n_slices;
    view->slice_width = D.slice_width.p;
    view->slice_base = D.slice_base.p;
    view->cols = D.cols.p;
    view->count = D.count.p;
    view->vals = vals;
    view->total = E.total();
    return FEMSHELL_OK;
}

// the pattern's device arrays become the operator's own (the cycle multiplies with them)
void adopt(AmgOperator &op, const EllPattern &E, DevPattern &D, DevBuf<double> &vals, int32_t n_cols_pad)
{
    std::swap(op.slice_width.p, D.slice_width.p);
    std::swap(op.slice_width.n, D.slice_width.n);
    std::swap(op.slice_base.p, D.slice_base.p);
    std::swap(op.slice_base.n, D.slice_base.n);
    std::swap(op.cols.p, D.cols.p);
    std::swap(op.cols.n, D.cols.n);
    std::swap(op.count.p, D.count.p); // (the row lengths: the next coarsening step's symbolic kernels read them)
    std::swap(op.count.n, D.count.n);
    std::swap(op.vals.p, vals.p);
    std::swap(op.vals.n, vals.n);
    op.nnzb = E.nnzb;
    op.n_cols_pad = n_cols_pad;
    op.dm = DeviceMatrix();
    op.dm.n_own = E.n_rows;
    op.dm.n_pad = E.n_pad;
    op.dm.n_slices = E.n_slices;
    op.dm.slice_width = op.slice_width.p;
    op.dm.slice_base = op.slice_base.p;
    op.dm.cols = op.cols.p;
    op.dm.vals = op.vals.p;
    op.dm.max_slice_width = E.max_width;
}

// host BSR (ascending columns) from a pattern and the ELL values brought back from the device
void ell_to_bsr(const EllPattern &E, const double *vals, int32_t n_cols, Bsr *out)
{
    Bsr &A = *out;
    A = Bsr();
    A.nr = E.n_rows;
    A.nc = n_cols;
    A.ptr.assign((size_t)E.n_rows + 1, 0);
    for (int32_t a = 0; a < E.n_rows; a++) A.ptr[a + 1] = A.ptr[a] + E.count[a];
    A.col.resize((size_t)A.ptr[E.n_rows]);
    A.val.resize((size_t)A.ptr[E.n_rows] * 36);
    parallel_chunks(E.n_rows, [&](int64_t a0, int64_t a1) {
        std::vector<std::pair<int32_t, int>> order;
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), n = (int)(a % kSliceNodes);
            order.clear();
            for (int k = 0; k < E.count[a]; k++) order.push_back({E.cols[(size_t)(E.slice_base[s] + (int64_t)k * kSliceNodes + n)], k});
            std::sort(order.begin(), order.end());
            int64_t nb = A.ptr[a];
            for (auto &ck : order) {
                A.col[(size_t)nb] = ck.first;
                const double *src = vals + (E.slice_base[s] + (int64_t)ck.second * kSliceNodes) * 36;
                double *blk = &A.val[(size_t)nb * 36];
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) blk[6 * i + j] = src[((int64_t)((j / 2) * 6 + i) * kSliceNodes + n) * 2 + (j & 1)];
                nb++;
            }
        }
    });
}

int download_vals(const DevBuf<double> &d, ValueArray *h, hipStream_t st)
{
    h->resize(d.n);
    FS_HIP(hipMemcpyAsync(h->data(), d.p, d.n * sizeof(double), hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    return FEMSHELL_OK;
}

int staged_download(femshell_ctx *c, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return FEMSHELL_OK;
    if (c == nullptr || c->stage_host == nullptr) {
        FS_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        return FEMSHELL_OK;
    }
    for (size_t off = 0; off < bytes; off += c->stage_bytes) {
        const size_t m = std::min(c->stage_bytes, bytes - off);
        FS_HIP(hipMemcpyAsync(c->stage_host, static_cast<const char *>(src_dev) + off, m, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        char *d = static_cast<char *>(dst_host) + off;
        const char *h = static_cast<const char *>(c->stage_host);
        parallel_chunks((int64_t)m, [&](int64_t b, int64_t e) { std::memcpy(d + b, h + b, (size_t)(e - b)); }, 1 << 20);
    }
    return FEMSHELL_OK;
}

int staged_upload(femshell_ctx *c, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st, int region)
{
    if (bytes == 0) return FEMSHELL_OK;
    if (c == nullptr || c->stage_host == nullptr) {
        FS_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        FS_HIP(hipStreamSynchronize(st));
        return FEMSHELL_OK;
    }
    for (size_t off = 0; off < bytes; off += c->stage_bytes) {
        const size_t m = std::min(c->stage_bytes, bytes - off);
        const char *s = static_cast<const char *>(src_host) + off;
        char *h = static_cast<char *>(c->stage_host) + (region ? c->stage_bytes : 0);
        parallel_chunks((int64_t)m, [&](int64_t b, int64_t e) { std::memcpy(h + b, s + b, (size_t)(e - b)); }, 1 << 20);
        FS_HIP(hipMemcpyAsync(static_cast<char *>(dst_dev) + off, h, m, hipMemcpyHostToDevice, st));
        FS_HIP(hipStreamSynchronize(st)); // (the buffer is free again)
    }
    return FEMSHELL_OK;
}

// the pattern of the context's K: the plan's slot arrays (padding slots of a row repeat the row's own index; multi-rank
// contexts never get here -- their hierarchy is built by the shadow context)
void pattern_of_plan(const Plan &p, HostEllPattern *out, bool light)
{
    HostEllPattern &H = *out;
    H = HostEllPattern();
    H.n = p.n_own;
    H.symmetric = p.symmetric;
    H.slice_width = p.slice_width;
    H.slice_base = p.slice_base;
    if (light) {
        H.borrowed = &p;
        if (p.symmetric) {
            H.in_width = p.in_width;
            H.in_base = p.in_base;
        }
        return;
    }
    auto copy_of = [](const RawVec<int32_t> &from, RawVec<int32_t> *to) { // (on the host threads: 32 + 2 x 24 MB at 4M triangles)
        to->resize(from.size());
        parallel_chunks((int64_t)from.size(), [&](int64_t b, int64_t e) { std::copy(from.begin() + b, from.begin() + e, to->begin() + b); }, 1 << 18);
    };
    copy_of(p.cols, &H.cols);
    H.count.assign((size_t)p.n_pad, 0);
    parallel_chunks(p.n_own, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), n = (int)(a % kSliceNodes);
            int cnt = 0;
            for (int k = 0; k < p.slice_width[s]; k++) {
                const int64_t slot = Plan::slot_index(p.slice_base[s], k, n);
                if (k == 0 || p.cols[slot] != a) cnt = k + 1; // real slots come first (ascending columns, padding behind)
            }
            H.count[(size_t)a] = (uint8_t)cnt;
        }
    });
    if (p.symmetric) {
        H.in_width = p.in_width;
        H.in_base = p.in_base;
        copy_of(p.in_slots, &H.in_slots);
        copy_of(p.in_rows, &H.in_rows);
    }
}

// the block graph of a level operator as a pattern-only BSR with ascending columns (both directions of a symmetric one)
void graph_of_pattern(const HostEllPattern &H, Bsr *G)
{
    Bsr &A = *G;
    A = Bsr();
    const int32_t n = H.n;
    A.nr = A.nc = n;
    A.ptr.assign((size_t)n + 1, 0);
    auto count_row = [&](int32_t a) {
        const int s = a / kSliceNodes, nn = a % kSliceNodes;
        int cnt = H.count[(size_t)a];
        if (H.symmetric)
            for (int k = 0; k < H.in_width[s]; k++)
                if (H.in_slots[(size_t)(H.in_base[s] + (int64_t)k * kSliceNodes + nn)] >= 0) cnt++;
        return cnt;
    };
    parallel_chunks(n, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) A.ptr[(size_t)a + 1] = count_row((int32_t)a);
    });
    for (int32_t a = 0; a < n; a++) A.ptr[(size_t)a + 1] += A.ptr[(size_t)a];
    A.col.resize((size_t)A.ptr[n]);
    parallel_chunks(n, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), nn = (int)(a % kSliceNodes);
            int64_t w = A.ptr[a];
            for (int k = 0; k < H.count[(size_t)a]; k++) A.col[(size_t)w++] = H.cols[(size_t)(H.slice_base[s] + (int64_t)k * kSliceNodes + nn)];
            if (H.symmetric)
                for (int k = 0; k < H.in_width[s]; k++) {
                    const size_t e = (size_t)(H.in_base[s] + (int64_t)k * kSliceNodes + nn);
                    if (H.in_slots[e] >= 0) A.col[(size_t)w++] = H.in_rows[e];
                }
            std::sort(A.col.begin() + A.ptr[a], A.col.begin() + A.ptr[a + 1]);
        }
    });
}

// The greedy passes of aggregate_nodes (amg_setup.cpp aggregate_piece) on the operator's ELL pattern itself -- own slots and
// in-list -- when they would run on the unfiltered graph in one piece and in index order (or the caller's visiting order): none of
// the three passes needs its neighbours sorted (pass 2 takes the LOWEST aggregated neighbour, a minimum), so the 23 ms the sorted
// graph of a 4M-triangle mesh costs the host are saved.  Returns -1 when the conditions do not hold (the caller builds the graph).
static int32_t aggregate_on_pattern(const HostEllPattern &H, const std::vector<int32_t> *visit, std::vector<int32_t> *aggout)
{
    const int32_t n = H.n;
    const bool have_visit = visit != nullptr && (int32_t)visit->size() == n;
    // (a borrowed pattern: the plan's arrays, whose padding slots -- behind the real ones -- repeat the row's own index)
    const Plan *pl = H.borrowed;
    const int32_t *cols = pl ? pl->cols.data() : H.cols.data();
    const int32_t *in_slots = pl ? pl->in_slots.data() : H.in_slots.data(), *in_rows = pl ? pl->in_rows.data() : H.in_rows.data();
    auto for_each_nb = [&](int32_t i, auto f) { // the node itself first
        const int s = i / kSliceNodes, nn = i % kSliceNodes;
        const int64_t base = H.slice_base[(size_t)s] + nn;
        if (pl) {
            if (!f(i)) return;
            const int W = H.slice_width[(size_t)s];
            for (int k = 1; k < W; k++) {
                const int32_t j = cols[(size_t)(base + (int64_t)k * kSliceNodes)];
                if (j == i) break; // padding from here on
                if (!f(j)) return;
            }
        } else {
            const int cnt = H.count[(size_t)i];
            for (int k = 0; k < cnt; k++)
                if (!f(cols[(size_t)(base + (int64_t)k * kSliceNodes)])) return;
        }
        if (H.symmetric) {
            const int64_t ib = H.in_base[(size_t)s] + nn;
            for (int k = 0; k < H.in_width[(size_t)s]; k++) {
                const size_t e = (size_t)(ib + (int64_t)k * kSliceNodes);
                if (in_slots[e] >= 0 && !f(in_rows[e])) return;
            }
        }
    };
    // row lengths, and what aggregation_order looks at
    std::atomic<int64_t> widest{0}, edges{0}, dist_sum{0};
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        int64_t w = 0, e = 0, d = 0;
        for (int64_t i = i0; i < i1; i++) {
            int64_t deg = 0;
            for_each_nb((int32_t)i, [&](int32_t j) {
                deg++;
                d += std::llabs((int64_t)j - i);
                return true;
            });
            w = std::max(w, deg);
            e += deg;
        }
        int64_t cur = widest.load();
        while (w > cur && !widest.compare_exchange_weak(cur, w)) {}
        edges.fetch_add(e);
        dist_sum.fetch_add(d);
    }, 1 << 16);
    if (!aggregation_is_plain(n, widest.load(), edges.load(), (double)dist_sum.load(), have_visit)) return -1;
    std::vector<int32_t> agg((size_t)n, -1);
    int32_t na = 0;
    // pass 1: a node whose whole neighbourhood is free becomes the root of a new aggregate
    for (int32_t v = 0; v < n; v++) {
        const int32_t i = have_visit ? (*visit)[(size_t)v] : v;
        if (agg[(size_t)i] >= 0) continue;
        int deg = 0;
        bool free_nb = true;
        for_each_nb(i, [&](int32_t j) {
            deg++;
            free_nb = agg[(size_t)j] < 0;
            return free_nb;
        });
        if (!free_nb || deg <= 1) continue;
        for_each_nb(i, [&](int32_t j) {
            agg[(size_t)j] = na;
            return true;
        });
        na++;
    }
    // pass 2: leftovers join the aggregate of their first aggregated neighbour in the visiting order (state of pass 1)
    std::vector<int32_t> rank;
    if (have_visit) {
        rank.resize((size_t)n);
        for (int32_t v = 0; v < n; v++) rank[(size_t)(*visit)[(size_t)v]] = v;
    }
    RawVec<int32_t> agg2((size_t)n);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            agg2[(size_t)i] = agg[(size_t)i];
            if (agg[(size_t)i] >= 0) continue;
            int32_t best = -1, best_rank = 0;
            for_each_nb((int32_t)i, [&](int32_t j) {
                if (agg[(size_t)j] < 0) return true;
                const int32_t rj = rank.empty() ? j : rank[(size_t)j];
                if (best < 0 || rj < best_rank) {
                    best = agg[(size_t)j];
                    best_rank = rj;
                }
                return true;
            });
            if (best >= 0) agg2[(size_t)i] = best;
        }
    }, 1 << 14);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) { std::copy(agg2.begin() + i0, agg2.begin() + i1, agg.begin() + i0); }, 1 << 16);
    // pass 3: what is still free forms aggregates of its own
    for (int32_t v = 0; v < n; v++) {
        const int32_t i = have_visit ? (*visit)[(size_t)v] : v;
        if (agg[(size_t)i] >= 0) continue;
        for_each_nb(i, [&](int32_t j) {
            if (agg[(size_t)j] < 0) agg[(size_t)j] = na;
            return true;
        });
        agg[(size_t)i] = na;
        na++;
    }
    aggout->swap(agg);
    return na;
}

// FEMSHELL_AMG_SYMBOLIC=host: the patterns of a coarsening step on the host threads as until round 6 (the path tests compare the
// device's with, and the fallback of rows too long for the lane sets); default: amg_symbolic.hip
static bool symbolic_on_device()
{
    const char *e = getenv("FEMSHELL_AMG_SYMBOLIC"); // (read per setup: the tests switch it inside one process)
    return !(e && std::strcmp(e, "host") == 0);
}

// The integer work of a coarsening step on the host threads: the patterns as lists, their sliced ELL images (kept: eP, eAP, eR,
// eAc), the uploads into S.  G: the level's graph, sorted rows.
static int symbolic_host(femshell_ctx *c, const HostEllPattern &pat, const Bsr &G, const std::vector<int32_t> &agg, int32_t na, bool sym_coarse,
                         const AmgPatches *patches, DevSymbolic &S, EllPattern &eP, EllPattern &eAP, EllPattern &eR, EllPattern &eAc,
                         const std::function<void(const char *)> &lap)
{
    hipStream_t st = c->stream;
    const int32_t n = pat.n;
    // the nodes grouped by aggregate (ascending inside an aggregate)
    {
        std::vector<int32_t> gptr((size_t)na + 1, 0), order((size_t)n);
        for (int32_t i = 0; i < n; i++) gptr[(size_t)agg[i] + 1]++;
        int32_t largest = 0;
        for (int32_t I = 0; I < na; I++) {
            largest = std::max(largest, gptr[(size_t)I + 1]);
            gptr[(size_t)I + 1] += gptr[I];
        }
        {
            std::vector<int32_t> fill(gptr.begin(), gptr.end() - 1);
            for (int32_t i = 0; i < n; i++) order[(size_t)fill[agg[i]]++] = i;
        }
        S.largest = largest;
        FS_HIP(S.gptr.upload(gptr, st));
        FS_HIP(S.order.upload(order, st));
        FS_HIP(hipStreamSynchronize(st)); // gptr / order go out of scope
    }
    // P: per fine row the sorted distinct aggregates of its neighbours (the row itself included)
    std::vector<int64_t> pptr((size_t)n + 1, 0);
    RawVec<int32_t> pcol;
    {
        std::vector<uint8_t> cnt((size_t)n, 0);
        RawVec<int32_t> tmp_all((size_t)G.ptr[n]); // upper bound storage: distinct aggregates per row, compacted below
        // rows of clustered nodes: the cluster blocks couple a row to what ALL the cluster's members see (P = P0 - omega B^-1 A P0)
        std::vector<std::vector<int32_t>> cluster_rows(patches ? (size_t)patches->n_clusters : 0);
        if (patches)
            parallel_chunks(patches->n_clusters, [&](int64_t k0, int64_t k1) {
                for (int64_t k = k0; k < k1; k++) {
                    std::vector<int32_t> &r = cluster_rows[(size_t)k];
                    for (int32_t t = patches->h_ptr[(size_t)k]; t < patches->h_ptr[(size_t)k + 1]; t++) {
                        const int32_t j = patches->h_nodes[(size_t)t];
                        for (int64_t q = G.ptr[j]; q < G.ptr[(size_t)j + 1]; q++) r.push_back(agg[G.col[q]]);
                    }
                    std::sort(r.begin(), r.end());
                    r.erase(std::unique(r.begin(), r.end()), r.end());
                }
            }, 64);
        parallel_chunks(n, [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                if (patches && patches->label_p[(size_t)a] >= 0) {
                    cnt[a] = (uint8_t)std::min<size_t>(cluster_rows[(size_t)patches->label_p[(size_t)a]].size(), 255);
                    continue;
                }
                int32_t *t = &tmp_all[(size_t)G.ptr[a]];
                int m = 0;
                for (int64_t q = G.ptr[a]; q < G.ptr[a + 1]; q++) t[m++] = agg[G.col[q]];
                std::sort(t, t + m);
                m = (int)(std::unique(t, t + m) - t);
                cnt[a] = (uint8_t)std::min(m, 255);
            }
        });
        for (int32_t a = 0; a < n; a++) pptr[a + 1] = pptr[a] + cnt[a];
        pcol.resize((size_t)pptr[n]);
        parallel_chunks(n, [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                if (patches && patches->label_p[(size_t)a] >= 0) std::copy_n(cluster_rows[(size_t)patches->label_p[(size_t)a]].data(), cnt[a], &pcol[(size_t)pptr[a]]);
                else std::copy_n(&tmp_all[(size_t)G.ptr[a]], cnt[a], &pcol[(size_t)pptr[a]]);
            }
        });
    }
    lap("  pattern of P");
    auto p_index = [&](int32_t row, int32_t J) -> int {
        const int32_t *b = &pcol[(size_t)pptr[row]], *e = &pcol[(size_t)pptr[row + 1]];
        return (int)(std::lower_bound(b, e, J) - b);
    };
    // which slot of P's row every block of K feeds (own slots and in-list entries, in the order the kernels walk them)
    // (entries of padding slots and of empty in-list places: 0, written by the row's thread like the real ones)
    RawVec<uint8_t> pmap_own((size_t)pat.slice_base.back()), pmap_in(pat.in_slots.size());
    const int32_t n_rows_padded = (int32_t)(pat.slice_width.size() * (size_t)kSliceNodes);
    parallel_chunks(n_rows_padded, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), nn = (int)(a % kSliceNodes);
            const int real = a < n ? pat.count[(size_t)a] : 0;
            for (int k = 0; k < pat.slice_width[(size_t)s]; k++) {
                const int64_t slot = pat.slice_base[s] + (int64_t)k * kSliceNodes + nn;
                pmap_own[(size_t)slot] = k < real ? (uint8_t)p_index((int32_t)a, agg[pat.cols[(size_t)slot]]) : (uint8_t)0;
            }
            if (pat.symmetric)
                for (int k = 0; k < pat.in_width[s]; k++) {
                    const size_t e = (size_t)(pat.in_base[s] + (int64_t)k * kSliceNodes + nn);
                    pmap_in[e] = a < n && pat.in_slots[e] >= 0 ? (uint8_t)p_index((int32_t)a, agg[pat.in_rows[e]]) : (uint8_t)0;
                }
        }
    });
    lap("  slots of P fed by the blocks of A");
    // A P: per fine row the union of the P rows of its neighbours
    std::vector<int64_t> aptr;
    RawVec<int32_t> acol;
    build_rows(n, [&](int32_t a, std::vector<int32_t> &out) {
        for (int64_t q = G.ptr[a]; q < G.ptr[a + 1]; q++) {
            const int32_t j = G.col[q];
            out.insert(out.end(), pcol.begin() + pptr[j], pcol.begin() + pptr[j + 1]);
        }
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
    }, &aptr, &acol);
    lap("  pattern of A P");
    // R = P^T as lists: per aggregate the fine rows (ascending) and the slot of the aggregate in their P row
    std::vector<int64_t> rptr((size_t)na + 1, 0);
    RawVec<int32_t> rrow((size_t)pptr[n]);
    RawVec<uint8_t> rk((size_t)pptr[n]);
    {
        // a counting sort by aggregate that keeps the rows ascending within an aggregate, on T ranges of rows at once: every
        // range counts into a histogram of its own, the histograms are stacked range after range per aggregate, every range
        // fills its entries -- the result is the serial one (5M random increments and 5M random writes otherwise: 30 ms)
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), std::min<int64_t>(32, (n + 65535) / 65536)));
        std::vector<std::vector<int32_t>> hist((size_t)T, std::vector<int32_t>((size_t)na, 0));
        auto range_of = [&](int t, int64_t *a0, int64_t *a1) {
            *a0 = (int64_t)n * t / T;
            *a1 = (int64_t)n * (t + 1) / T;
        };
        parallel_chunks(T, [&](int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; t++) {
                int64_t a0, a1;
                range_of((int)t, &a0, &a1);
                std::vector<int32_t> &h = hist[(size_t)t];
                for (int64_t q = pptr[(size_t)a0]; q < pptr[(size_t)a1]; q++) h[(size_t)pcol[(size_t)q]]++;
            }
        }, 1);
        // per aggregate: first position of every range's entries (hist becomes the running offset)
        parallel_chunks(na, [&](int64_t I0, int64_t I1) {
            for (int64_t I = I0; I < I1; I++) {
                int64_t total = 0;
                for (int t = 0; t < T; t++) total += hist[(size_t)t][(size_t)I];
                rptr[(size_t)I + 1] = total;
            }
        }, 4096);
        for (int32_t I = 0; I < na; I++) rptr[(size_t)I + 1] += rptr[(size_t)I];
        parallel_chunks(na, [&](int64_t I0, int64_t I1) {
            for (int64_t I = I0; I < I1; I++) {
                int64_t at = 0;
                for (int t = 0; t < T; t++) {
                    const int32_t cnt_t = hist[(size_t)t][(size_t)I];
                    hist[(size_t)t][(size_t)I] = (int32_t)at; // offset of range t inside the aggregate's list
                    at += cnt_t;
                }
            }
        }, 4096);
        parallel_chunks(T, [&](int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; t++) {
                int64_t a0, a1;
                range_of((int)t, &a0, &a1);
                std::vector<int32_t> &h = hist[(size_t)t];
                for (int64_t a = a0; a < a1; a++)
                    for (int64_t q = pptr[(size_t)a]; q < pptr[(size_t)a + 1]; q++) {
                        const int32_t I = pcol[(size_t)q];
                        const int64_t d = rptr[(size_t)I] + h[(size_t)I]++;
                        rrow[(size_t)d] = (int32_t)a;
                        rk[(size_t)d] = (uint8_t)(q - pptr[(size_t)a]);
                    }
            }
        }, 1);
    }
    lap("  lists of R");
    // A_c: per aggregate the union of the A P rows of its fine rows (symmetric storage: columns >= the row only; the
    // coarse operator is symmetric, the cycle applies the stored blocks to both rows)
    std::vector<int64_t> cptr;
    RawVec<int32_t> ccol;
    build_rows(na, [&](int32_t I, std::vector<int32_t> &out) {
        for (int64_t q = rptr[I]; q < rptr[I + 1]; q++) {
            const int32_t i = rrow[(size_t)q];
            out.insert(out.end(), acol.begin() + aptr[i], acol.begin() + aptr[i + 1]);
        }
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
        if (sym_coarse) out.erase(out.begin(), std::lower_bound(out.begin(), out.end(), I)); // diagonal and upper blocks
    }, &cptr, &ccol);
    lap("  pattern of Ac");
    if (!pack_pattern(n, pptr.data(), pcol.data(), false, &eP) || !pack_pattern(n, aptr.data(), acol.data(), false, &eAP) ||
        !pack_pattern(na, cptr.data(), ccol.data(), true, &eAc))
        return set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: a row of an intermediate operator has more than 255 blocks");
    // (R's columns are the fine rows, already ascending per aggregate)
    if (!pack_pattern(na, rptr.data(), rrow.data(), false, &eR))
        return set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: an aggregate is seen by more than 255 fine rows");
    lap("  sliced layouts of the four");
    // work of the Galerkin product: useful = one 6x6x6 product per (fine row i, aggregate I in P's row i, block of
    // (A P)'s row i); issued on the matrix cores = 16x16x4 tiles, two k-steps per fine row and panel tile
    {
        double useful = 0.0, issued = 0.0;
        for (int32_t a = 0; a < n; a++) useful += 432.0 * (double)(pptr[a + 1] - pptr[a]) * (double)(aptr[a + 1] - aptr[a]);
        for (int32_t I = 0; I < na; I++) {
            const int cnt = (int)(cptr[I + 1] - cptr[I]);
            double tiles = 0.0;
            for (int g0 = 0; g0 < cnt; g0 += 16) tiles += (double)((6 * std::min(16, cnt - g0) + 15) / 16);
            issued += 2048.0 * 2.0 * tiles * (double)(rptr[I + 1] - rptr[I]);
        }
        S.useful_flops = useful;
        S.mfma_flops = issued;
    }
    FS_HIP(S.agg.upload(agg, st));
    FS_HIP(S.pmap_own.upload(pmap_own, st));
    FS_HIP(S.pmap_in.upload(pmap_in, st));
    FS_HIP(S.rptr.upload(rptr, st));
    FS_HIP(S.rrow.upload(rrow, st));
    FS_HIP(S.rk.upload(rk, st));
    auto up = [&](const EllPattern &E, DevPattern &D, int64_t *total) -> int {
        EllView unused;
        *total = E.total();
        return upload_pattern(E, D, nullptr, &unused, st);
    };
    int rc = up(eP, S.P, &S.totP);
    if (!rc) rc = up(eAP, S.AP, &S.totAP);
    if (!rc) rc = up(eR, S.R, &S.totR);
    if (!rc) rc = up(eAc, S.Ac, &S.totAc);
    if (rc) return rc;
    FS_HIP(hipStreamSynchronize(st)); // the host arrays above go out of scope
    lap("uploads");
    return FEMSHELL_OK;
}

static EllView view_of(const EllPattern &E, const DevPattern &D, double *vals, int64_t total)
{
    EllView v;
    v.n_rows = E.n_rows;
    v.n_slices = E.n_slices;
    v.slice_width = D.slice_width.p;
    v.slice_base = D.slice_base.p;
    v.cols = D.cols.p;
    v.count = D.count.p;
    v.vals = vals;
    v.total = total;
    return v;
}

// One coarsening step on the device.  In: the level operator in HBM (Adev: the context's K on level 0; block-Jacobi inverse
// valid), the host copy of its pattern, the near-null space B of the fine nodes, the spectral bound lam.  Out: L.P, L.R (operators
// of the cycle), next.A (the coarse level matrix in HBM, diagonal slot first), Ac_host (its host copy for the remaining levels),
// Bc, and for small problems the host copies the inspection exports want.  The host runs the greedy passes of the aggregation; the
// patterns are built in HBM (amg_symbolic.hip) unless the level has clusters of rigidly coupled nodes, a row outgrows the lane
// sets or FEMSHELL_AMG_SYMBOLIC=host asks for the host's lists.
int amg_device_coarsen(femshell_ctx *c, const DeviceMatrix &Adev, const HostEllPattern &pat_in, AmgLevel &L, AmgLevel &next,
                       const NearNullSrc &B, const std::function<int(double *)> &lam_of, bool keep_host, const std::function<bool(int32_t)> &want_host,
                       Bsr *Ac_host, std::vector<double> *Bc_out, DevBuf<double> *Bc_dev,
                       const std::function<void(const char *)> &lap, const std::function<int()> &before_qr)
{
    hipStream_t st = c->stream;
    // (a light pattern of level 0 -- HostEllPattern::borrowed -- is filled when the host's lists are needed after all)
    HostEllPattern filled;
    const HostEllPattern *patp = &pat_in;
    auto need_lists = [&] {
        if (patp->borrowed != nullptr && patp->cols.empty()) {
            pattern_of_plan(*patp->borrowed, &filled, false);
            patp = &filled;
        }
    };
    const int32_t n = pat_in.n;
    if (c->cfg.world_size != 1) return set_err(FEMSHELL_ERR_UNSUPPORTED, "multigrid setup: single-rank contexts only");
    const bool finest = &Adev == &c->dm;
    // ---- aggregation
    // When the library renumbered the nodes itself (FEMSHELL_REORDER_*: Morton, Cuthill-McKee) the greedy passes of the
    // FINEST level visit the nodes in the caller's order: the index order of a space-filling curve fragments the aggregates
    // at its jumps (4M-triangle cylinder, Morton numbering: 443 iterations instead of 137; panel 160 instead of 147), and the
    // aggregates -- hence every coarser level, whose numbering is the order of creation -- are then those of the caller's
    // numbering, whatever the internal one is
    const bool finest_renumbered = finest && !c->iperm.empty() && (int32_t)c->iperm.size() == n;
    const std::vector<int32_t> *visit = finest_renumbered ? &c->iperm : nullptr;
    // (clusters of rigidly coupled nodes -- amg_patch.hpp -- are glued into one node each before the greedy passes)
    const AmgPatches *patches = L.patches.get();
    // the level's pattern in HBM as the symbolic kernels walk it (levels >= 1: the row lengths the previous step left beside it)
    const uint8_t *dev_count = finest ? nullptr : L.A.count.p;
    const bool try_device = symbolic_on_device() && !patches && (finest || dev_count != nullptr);
    std::vector<int32_t> agg;
    Bsr G;
    int32_t na = -1;
    if (!(patches && patches->glue) && try_device) na = aggregate_on_pattern(*patp, visit, &agg); // (-1: the sorted graph is needed)
    if (na < 0) {
        need_lists();
        graph_of_pattern(*patp, &G);
        lap("  graph of the level");
        na = (patches && patches->glue) ? aggregate_nodes_glued(G, patches->label_p, &agg, visit) : aggregate_nodes(G, &agg, visit);
    }
    lap("  aggregation");
    const bool sym_coarse = coarse_symmetric_storage(na);

    // ---- patterns
    std::unique_ptr<DevSymbolic> Sp(new DevSymbolic());
    EllPattern eP, eAP, eR, eAc; // host path: complete; device path: the scalars, arrays on demand
    bool on_device = false;
    if (try_device) {
        GraphView gv;
        gv.n = n;
        gv.n_slices = (int32_t)pat_in.slice_width.size();
        gv.slice_width = Adev.slice_width;
        gv.slice_base = Adev.slice_base;
        gv.cols = Adev.cols;
        gv.count = dev_count;
        gv.symmetric = pat_in.symmetric ? 1 : 0;
        gv.in_width = Adev.in_width;
        gv.in_base = Adev.in_base;
        gv.in_slots = Adev.in_slots;
        gv.in_rows = Adev.in_rows;
        const int rcd = amg_symbolic_device(c, st, gv, pat_in.slice_base.back(), pat_in.symmetric ? pat_in.in_base.back() : (int64_t)0, agg, na, sym_coarse,
                                            Sp.get());
        if (rcd == FEMSHELL_OK) {
            on_device = true;
            c->amg->stats.symbolic_device++;
            eP = Sp->iP;
            eAP = Sp->iAP;
            eR = Sp->iR;
            eAc = Sp->iAc;
            lap("  patterns in HBM");
        } else if (rcd != FEMSHELL_ERR_UNSUPPORTED) {
            return rcd;
        } else {
            FS_HIP(hipStreamSynchronize(st));
            Sp.reset(new DevSymbolic()); // a row too long for the lane sets: the host's lists
            c->amg->stats.symbolic_fallback++;
        }
    }
    if (!on_device) {
        if (!try_device) c->amg->stats.symbolic_host++;
        need_lists();
        if (G.nr != n) {
            graph_of_pattern(*patp, &G);
            lap("  graph of the level");
        }
        const int rch = symbolic_host(c, *patp, G, agg, na, sym_coarse, patches, *Sp, eP, eAP, eR, eAc, lap);
        if (rch) return rch;
    }
    DevSymbolic &S = *Sp;

    // ---- tentative prolongator on the device: QR of every aggregate's rows of B, one wave each (k_amg_tentative_qr)
    DevBuf<double> d_Q;
    {
        FS_HIP(d_Q.alloc((size_t)n * 36));
        FS_HIP(Bc_dev->alloc((size_t)na * 36));
        // FEMSHELL_AMG_QR=memory: every aggregate takes the path of the large ones (rows in Q instead of registers; tests)
        const bool in_memory = getenv("FEMSHELL_AMG_QR") && std::string(getenv("FEMSHELL_AMG_QR")) == "memory";
        if (before_qr) {
            const int rcq = before_qr();
            if (rcq) return rcq;
        }
        launch_amg_tentative_qr(B, S.gptr.p, S.order.p, na, S.largest, d_Q.p, Bc_dev->p, in_memory, st);
        FS_HIP(hipGetLastError());
        Bc_out->clear();
        if (want_host(na) || keep_host) {
            Bc_out->resize((size_t)na * 36);
            FS_HIP(hipMemcpyAsync(Bc_out->data(), Bc_dev->p, Bc_out->size() * sizeof(double), hipMemcpyDeviceToHost, st));
            FS_HIP(hipStreamSynchronize(st));
        }
    }
    lap("tentative P");

    // ---- values on the device
    DevBuf<double> vP, vAP, vR, vAc;
    FS_HIP(vP.alloc((size_t)S.totP * 36));
    FS_HIP(vAP.alloc((size_t)S.totAP * 36));
    FS_HIP(vR.alloc((size_t)S.totR * 36));
    FS_HIP(vAc.alloc((size_t)S.totAc * 36));
    lap("  allocation of the values");
    const EllView wP = view_of(eP, S.P, vP.p, S.totP), wAP = view_of(eAP, S.AP, vAP.p, S.totAP), wR = view_of(eR, S.R, vR.p, S.totR),
                  wAc = view_of(eAc, S.Ac, vAc.p, S.totAc);
    // FEMSHELL_AMG_GALERKIN=mfma: one wave per coarse row on the matrix cores instead of one lane per result block on the
    // vector ALUs.  Measured on the 4M-triangle panel with A P stored block-contiguously: 2.5 ms on the vector ALUs, 8.2 ms
    // on the matrix cores (6-row panels leave 10 of 16 tile rows idle and the operands arrive 8 bytes at a time) -- the
    // product is a gather of 288-byte blocks at 4 TFLOP/s, not a GEMM, so the vector-ALU kernel is the default and the
    // matrix-core kernel the measured alternative (tests run both; bench.py reports both).  Read per setup.
    const bool use_mfma = getenv("FEMSHELL_AMG_GALERKIN") && std::string(getenv("FEMSHELL_AMG_GALERKIN")) == "mfma";
    // the spectral bound of the level: its power iteration has been running beside the host work above
    double lam = 0.0;
    int rc = lam_of(&lam);
    if (rc) return rc;
    lap("  spectral bound (the power iteration's end)");
    hipEvent_t ev[5];
    for (auto &e : ev) FS_HIP(hipEventCreate(&e));
    FS_HIP(hipEventRecord(ev[0], st));
    launch_amg_prolongator(Adev, S.agg.p, d_Q.p, (4.0 / 3.0) / lam, S.pmap_own.p, S.pmap_in.p, wP, st);
    if (patches) { // the cluster blocks' share of the smoothing
        launch_patch_prolongator(Adev, S.agg.p, d_Q.p, (4.0 / 3.0) / lam, wP, patches->view(), eP.max_width, st);
    }
    FS_HIP(hipEventRecord(ev[1], st));
    launch_amg_ap(Adev, wP, wAP, st);
    FS_HIP(hipEventRecord(ev[2], st));
    launch_amg_restriction(wP, S.rptr.p, S.rrow.p, S.rk.p, wR, st);
    FS_HIP(hipEventRecord(ev[3], st));
    launch_amg_galerkin(wP, wAP, S.rptr.p, S.rrow.p, S.rk.p, wAc, st, use_mfma);
    FS_HIP(hipEventRecord(ev[4], st));
    FS_HIP(hipGetLastError());
    FS_HIP(hipStreamSynchronize(st));
    if (!c->amg->levels.empty() && &L == c->amg->levels[0].get()) { // the statistics describe the step that matters: level 0
        AmgSetupStats &T = c->amg->stats;
        float ms = 0.f;
        FS_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        T.prolongator_ms = ms;
        FS_HIP(hipEventElapsedTime(&ms, ev[1], ev[2]));
        T.ap_ms = ms;
        FS_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
        T.restriction_ms = ms;
        FS_HIP(hipEventElapsedTime(&ms, ev[3], ev[4]));
        T.galerkin_ms = ms;
        T.galerkin_mfma = use_mfma ? 1 : 0;
        T.galerkin_useful_flops = S.useful_flops;
        T.galerkin_mfma_flops_issued = use_mfma ? S.mfma_flops : 0.0;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    lap("P, AP, R, Ac on the device");

    // ---- the coarse operator goes back for the remaining levels; small problems keep P for the inspection exports
    if (on_device) { // the host copy of the coarse pattern: the next step's aggregation reads it (and the exports below)
        rc = download_pattern(c, S.Ac, S.totAc, &eAc, st);
        if (rc) return rc;
        if (keep_host) {
            rc = download_pattern(c, S.P, S.totP, &eP, st);
            if (rc) return rc;
        }
    }
    lap("  host copy of the coarse pattern");
    {
        ValueArray h;
        *Ac_host = Bsr();
        if (want_host(na) || keep_host) {
            rc = download_vals(vAc, &h, st);
            if (rc) return rc;
            ell_to_bsr(eAc, h.data(), na, Ac_host);
            if (sym_coarse) mirror_upper(Ac_host); // the host algorithms of the next levels take the full matrix
        }
        if (keep_host) {
            rc = download_vals(vP, &h, st);
            if (rc) return rc;
            ell_to_bsr(eP, h.data(), na, &L.hP);
        }
        L.agg.swap(agg); // (the inspection export; row-partitioned contexts look up which coarse rows their nodes reach)
    }
    const int32_t nc_pad = eAc.n_pad;
    const int64_t total_ac = S.totAc;
    adopt(L.P, eP, S.P, vP, nc_pad);
    adopt(L.R, eR, S.R, vR, eP.n_pad);
    adopt(next.A, eAc, S.Ac, vAc, nc_pad);
    next.pattern = HostEllPattern();
    next.pattern.n = na;
    next.pattern.symmetric = sym_coarse;
    if (sym_coarse) {
        if (on_device) {
            HostEllPattern &Np = next.pattern;
            Np.in_width.resize(S.in_width.n);
            Np.in_base.resize(S.in_base.n);
            Np.in_slots.resize((size_t)S.in_total);
            Np.in_rows.resize((size_t)S.in_total);
            rc = staged_download(c, Np.in_width.data(), S.in_width.p, S.in_width.n * sizeof(int32_t), st);
            if (!rc) rc = staged_download(c, Np.in_base.data(), S.in_base.p, S.in_base.n * sizeof(int64_t), st);
            if (!rc) rc = staged_download(c, Np.in_slots.data(), S.in_slots.p, (size_t)S.in_total * sizeof(int32_t), st);
            if (!rc) rc = staged_download(c, Np.in_rows.data(), S.in_rows.p, (size_t)S.in_total * sizeof(int32_t), st);
            if (rc) return rc;
            rc = attach_in_lists_device(next.A, S.in_width, S.in_base, S.in_slots, S.in_rows, S.max_in_width, total_ac, st);
            if (rc) return rc;

        } else {
            SlicedEllSym I;
            build_in_lists(na, eAc.slice_width, eAc.slice_base, eAc.cols.data(), eAc.count, &I);
            rc = attach_in_lists(next.A, I, total_ac, st);
            if (rc) return rc;
            next.pattern.in_width.swap(I.in_width);
            next.pattern.in_base.swap(I.in_base);
            next.pattern.in_slots.swap(I.in_slots);
            next.pattern.in_rows.swap(I.in_rows);
        }
    }
    next.pattern.slice_width.swap(eAc.slice_width);
    next.pattern.slice_base.swap(eAc.slice_base);
    next.pattern.cols.swap(eAc.cols);
    next.pattern.count.swap(eAc.count);
    lap("download of the coarse operator");
    return FEMSHELL_OK;
}

} // namespace femshell
