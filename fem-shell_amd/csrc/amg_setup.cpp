// amg_setup.cpp -- host side of the smoothed-aggregation setup (see amg.hpp): sparse block algebra on BSR
// matrices with 6x6 blocks, run once per matrix on the host's threads.  No device code here.
#include "amg.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "plan.hpp"

namespace femshell {

int host_threads() { return host_thread_count(); }

void parallel_chunks(int64_t n, const std::function<void(int64_t, int64_t)> &f, int64_t min_chunk)
{
    if (n <= 0) return;
    int64_t nt = std::min<int64_t>(host_threads(), (n + min_chunk - 1) / min_chunk);
    if (nt <= 1) {
        f(0, n);
        return;
    }
    run_on_host_threads((int)nt, [&f, n, nt](int t) { f(n * t / nt, n * (t + 1) / nt); });
}

namespace {

// c += a * b (6x6, row-major)
inline void blk_mac(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) {
            const double aik = a[6 * i + k];
            for (int j = 0; j < 6; j++) c[6 * i + j] += aik * b[6 * k + j];
        }
}

// 6x6 SPD inverse by Cholesky; false when the block is not positive definite
bool spd_inverse6(const double *A, double *inv)
{
    double L[6][6] = {}, Li[6][6] = {};
    for (int c = 0; c < 6; c++) {
        double d = A[6 * c + c];
        for (int k = 0; k < c; k++) d -= L[c][k] * L[c][k];
        if (!(d > 0.0)) return false;
        const double lcc = std::sqrt(d);
        L[c][c] = lcc;
        for (int r = c + 1; r < 6; r++) {
            double v = A[6 * r + c];
            for (int k = 0; k < c; k++) v -= L[r][k] * L[c][k];
            L[r][c] = v / lcc;
        }
    }
    for (int c = 0; c < 6; c++) {
        Li[c][c] = 1.0 / L[c][c];
        for (int r = c + 1; r < 6; r++) {
            double v = 0.0;
            for (int k = c; k < r; k++) v -= L[r][k] * Li[k][c];
            Li[r][c] = v / L[r][r];
        }
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double v = 0.0;
            for (int k = std::max(i, j); k < 6; k++) v += Li[k][i] * Li[k][j];
            inv[6 * i + j] = v;
        }
    return true;
}

} // namespace

void node_normals(int32_t n, const double *xyz, int64_t n_tri, const int32_t *tri, int64_t n_quad, const int32_t *quad,
                  std::vector<double> *out)
{
    std::vector<double> &N = *out;
    N.assign((size_t)n * 3, 0.0);
    auto cross_of = [&](int32_t a, int32_t b, int32_t c, double w[3]) { // (b - a) x (c - a): twice the area times the normal
        const double *A = xyz + 3ll * a, *B = xyz + 3ll * b, *C = xyz + 3ll * c;
        const double u[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, v[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        w[0] = u[1] * v[2] - u[2] * v[1];
        w[1] = u[2] * v[0] - u[0] * v[2];
        w[2] = u[0] * v[1] - u[1] * v[0];
    };
    // Ranges of nodes on the host threads: every thread walks all elements and adds to the nodes of its own range only, so
    // that a node's sum runs over its elements in ascending order (triangles, then quadrilaterals) whatever the number of
    // threads is -- the serial scatter loop's result, bit for bit.
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 32768));
    parallel_chunks(T, [&](int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++) {
            const int32_t a0 = (int32_t)((int64_t)n * t / T), a1 = (int32_t)((int64_t)n * (t + 1) / T);
            const uint32_t span = (uint32_t)(a1 - a0);
            auto mine = [&](int32_t a) { return (uint32_t)(a - a0) < span; };
            auto add = [&](int32_t a, const double w[3]) {
                if (mine(a))
                    for (int d = 0; d < 3; d++) N[3ull * a + d] += w[d];
            };
            for (int64_t e = 0; e < n_tri; e++) {
                const int32_t *c = tri + 3 * e;
                if (!(mine(c[0]) || mine(c[1]) || mine(c[2]))) continue;
                double w[3];
                cross_of(c[0], c[1], c[2], w);
                for (int i = 0; i < 3; i++) add(c[i], w);
            }
            for (int64_t e = 0; e < n_quad; e++) {
                const int32_t *c = quad + 4 * e;
                if (!(mine(c[0]) || mine(c[1]) || mine(c[2]) || mine(c[3]))) continue;
                double w[3], w2[3];
                cross_of(c[0], c[1], c[2], w);
                cross_of(c[0], c[2], c[3], w2);
                for (int d = 0; d < 3; d++) w[d] += w2[d];
                for (int i = 0; i < 4; i++) add(c[i], w);
            }
            for (int32_t a = a0; a < a1; a++) {
                double *v = &N[3ull * a];
                const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (l > 0.0)
                    for (int d = 0; d < 3; d++) v[d] /= l;
            }
        }
    }, 1);
}

// The same normals from the plan's gather lists: the diagonal slot of a node row lists every element the node belongs to, in
// ascending local element order -- triangles, then quadrilaterals: the order of the sums above -- so a node's normal is a loop over
// its own six elements instead of every thread's walk over all of them (11 ms at 4M triangles on sixteen threads, 0.27 s at 32M:
// threads x elements).  Bit for bit the array node_normals returns (tests/test_plan_cpu.py).
void node_normals_plan(const Plan &p, RawVec<double> *out)
{
    const int32_t n = p.n_own, nlt = p.n_ltri();
    const double *xyz = p.xyz_local.data();
    const int32_t *tri = p.tri_local.data(), *quad = p.quad_local.data();
    out->resize((size_t)n * 3);
    double *N = out->data();
    auto cross_of = [&](int32_t a, int32_t b, int32_t c, double w[3]) {
        const double *A = xyz + 3ll * a, *B = xyz + 3ll * b, *C = xyz + 3ll * c;
        const double u[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, v[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        w[0] = u[1] * v[2] - u[2] * v[1];
        w[1] = u[2] * v[0] - u[0] * v[2];
        w[2] = u[0] * v[1] - u[1] * v[0];
    };
    parallel_chunks(n, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            const int64_t slot = p.slice_base[(size_t)(a / kSliceNodes)] + (a % kSliceNodes); // slot 0 of the row: the diagonal block
            double v[3] = {0.0, 0.0, 0.0};
            for (int32_t q = p.pair_ptr[(size_t)slot]; q < p.pair_ptr[(size_t)slot + 1]; q++) {
                const int32_t le = (int32_t)(p.pairs[(size_t)q] >> 4);
                double w[3];
                if (le < nlt) {
                    const int32_t *c = tri + 3ll * le;
                    cross_of(c[0], c[1], c[2], w);
                } else {
                    const int32_t *c = quad + 4ll * (le - nlt);
                    double w2[3];
                    cross_of(c[0], c[1], c[2], w);
                    cross_of(c[0], c[2], c[3], w2);
                    for (int d = 0; d < 3; d++) w[d] += w2[d];
                }
                for (int d = 0; d < 3; d++) v[d] += w[d];
            }
            const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (l > 0.0)
                for (int d = 0; d < 3; d++) v[d] /= l;
            for (int d = 0; d < 3; d++) N[3 * a + d] = v[d];
        }
    }, 1 << 14);
}

// Near-null space: the six rigid-body modes of the mesh -- with one correction for this element.  The drilling stiffness
// (fem-shell.cpp:1035-1052) is a penalty on the rotation about the element normal that is NOT coupled to the in-plane
// displacements: a rigid rotation omega of the whole structure, nodal rotations included, costs the drilling energy of
// theta_n = omega . n, while the same displacement field with theta_n = 0 costs nothing.  The latter is what the operator's
// soft mode looks like, so the rotational part of the rotation modes is projected onto the tangent plane of each node:
// theta = omega - (omega . n) n.  With the plain modes the multigrid needed 715 iterations on a 10k-triangle flap loaded in
// its own plane (the coupled example), and the coarsest operators of the 250k-triangle flap lost definiteness.
void mesh_centre(int32_t n, const double *xyz, double c[3])
{
    c[0] = c[1] = c[2] = 0.0;
    for (int32_t a = 0; a < n; a++)
        for (int d = 0; d < 3; d++) c[d] += xyz[3ll * a + d];
    for (int d = 0; d < 3; d++) c[d] /= std::max(n, 1);
}

void rigid_body_modes(int32_t n, const double *xyz, const uint8_t *dmask, std::vector<double> *Bout, const double *normals)
{
    std::vector<double> &B = *Bout;
    B.assign((size_t)n * 36, 0.0);
    double c[3];
    mesh_centre(n, xyz, c);
    parallel_chunks(n, [&](int64_t b0, int64_t b1) {
        for (int64_t a = b0; a < b1; a++) {
            double *b = &B[(size_t)a * 36];
            const double x = xyz[3 * a] - c[0], y = xyz[3 * a + 1] - c[1], z = xyz[3 * a + 2] - c[2];
            for (int i = 0; i < 6; i++) b[6 * i + i] = 1.0;
            // u = omega x r: rotation about x -> (0,-z,y), about y -> (z,0,-x), about z -> (-y,x,0)
            b[6 * 1 + 3] = -z;
            b[6 * 2 + 3] = y;
            b[6 * 0 + 4] = z;
            b[6 * 2 + 4] = -x;
            b[6 * 0 + 5] = -y;
            b[6 * 1 + 5] = x;
            if (normals != nullptr) {
                const double *nv = normals + 3 * a;
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) b[6 * (3 + i) + 3 + j] -= nv[i] * nv[j];
            }
            const uint8_t m = dmask ? dmask[a] : 0;
            for (int v = 0; v < 6; v++)
                if ((m >> v) & 1u)
                    for (int j = 0; j < 6; j++) b[6 * v + j] = 0.0;
        }
    });
}

// Order in which the greedy aggregation visits the nodes.  The passes below sweep the nodes in sequence, which gives
// compact aggregates when the numbering itself sweeps the mesh (structured grids, Morton / Cuthill-McKee numberings) and
// ragged ones when it does not: on a 500k-triangle Delaunay mesh with shuffled numbering the solve took 287 iterations
// against 209 with a Cuthill-McKee numbering.  When the numbering is scattered (mean index distance of neighbours beyond
// 8 sqrt(n)) the nodes are therefore visited in breadth-first order of the graph (components from their lowest node,
// neighbours in column order); otherwise in index order.  Empty result = index order.
void aggregation_order(const Bsr &A, std::vector<int32_t> *order)
{
    const int32_t n = A.nr;
    order->clear();
    if (n < 64) return;
    // (an integer sum: exact whatever the number of host threads is)
    std::atomic<int64_t> dist_sum{0};
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        int64_t d = 0;
        for (int64_t i = i0; i < i1; i++)
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) d += std::llabs((int64_t)A.col[q] - i);
        dist_sum.fetch_add(d);
    }, 1 << 16);
    const double dist = (double)dist_sum.load();
    const int64_t edges = A.ptr[n];
    // FEMSHELL_AMG_AGG_ORDER=bfs|index forces one of the two (experiments); default: by the scatter of the numbering
    const char *force = getenv("FEMSHELL_AMG_AGG_ORDER");
    if (force && std::strcmp(force, "index") == 0) return;
    const bool bfs = force && std::strcmp(force, "bfs") == 0;
    if (!bfs && (edges == 0 || dist / (double)edges <= 8.0 * std::sqrt((double)n))) return;
    order->reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    for (int32_t s0 = 0; s0 < n; s0++) {
        if (seen[s0]) continue;
        seen[s0] = 1;
        size_t head = order->size();
        order->push_back(s0);
        while (head < order->size()) {
            const int32_t i = (*order)[head++];
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
                const int32_t j = A.col[q];
                if (!seen[j]) {
                    seen[j] = 1;
                    order->push_back(j);
                }
            }
        }
    }
}

// The graph the greedy passes see when rows have more than `keep` neighbours: per node the `keep` neighbours it shares most
// neighbours with (ties: the lower column), an edge stays when either end keeps it.  Why: the Galerkin operators of a structured
// mesh reach three fine couplings far, and depending on how the aggregates of level 1 happen to tile -- a matter of the row length
// modulo the tile -- the graph of level 2 has 13 or 17 neighbours per node; with 17 the aggregates of levels 2 and 3 hold 20
// nodes instead of 15 and the 10M-triangle pinched cylinder needs 164 iterations where the 9M-triangle one needs 89
// (profiles/r05_aggregation_lottery.txt).  The far couplings are the ones that share few neighbours with the node.  Rows are
// sorted (graph_of_pattern, the host operators' columns ascend); keep <= 0 or no row beyond it: the graph itself.
static const Bsr *graph_for_aggregation(const Bsr &A, int keep, Bsr *store)
{
    const int32_t n = A.nr;
    if (keep <= 0) return &A;
    int64_t widest = 0;
    for (int32_t i = 0; i < n; i++) widest = std::max<int64_t>(widest, A.ptr[(size_t)i + 1] - A.ptr[(size_t)i]);
    if (widest <= (int64_t)keep + 1) return &A; // (the node itself is one of its columns)
    RawVec<uint8_t> kept((size_t)A.ptr[n]);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        std::vector<std::pair<int32_t, int32_t>> cand; // (-common neighbours, column)
        for (int64_t i = i0; i < i1; i++) {
            const int64_t b = A.ptr[i], e = A.ptr[i + 1];
            if (e - b <= (int64_t)keep + 1) {
                for (int64_t q = b; q < e; q++) kept[(size_t)q] = 1;
                continue;
            }
            cand.clear();
            for (int64_t q = b; q < e; q++) {
                const int32_t j = A.col[q];
                kept[(size_t)q] = j == (int32_t)i ? 1 : 0;
                if (j == (int32_t)i) continue;
                int32_t common = 0;
                int64_t x = b, y = A.ptr[j];
                const int64_t ye = A.ptr[(size_t)j + 1];
                while (x < e && y < ye) {
                    const int32_t cx = A.col[x], cy = A.col[y];
                    if (cx == cy) {
                        common++;
                        x++;
                        y++;
                    } else if (cx < cy) {
                        x++;
                    } else {
                        y++;
                    }
                }
                cand.emplace_back(-common, j);
            }
            std::sort(cand.begin(), cand.end());
            for (int k = 0; k < keep && k < (int)cand.size(); k++) {
                const int32_t j = cand[(size_t)k].second;
                const int64_t q = std::lower_bound(A.col.begin() + b, A.col.begin() + e, j) - A.col.begin();
                if (q < e && A.col[q] == j) kept[(size_t)q] = 1; // (always, on rows of ascending columns -- what every caller hands over)
            }
        }
    }, 1 << 12);
    Bsr &G = *store;
    G = Bsr();
    G.nr = G.nc = n;
    G.ptr.assign((size_t)n + 1, 0);
    auto stays = [&](int64_t i, int64_t q) {
        if (kept[(size_t)q]) return true;
        const int32_t j = A.col[q]; // the other end's opinion
        const int64_t b = A.ptr[j], e = A.ptr[(size_t)j + 1];
        const int64_t r = std::lower_bound(A.col.begin() + b, A.col.begin() + e, (int32_t)i) - A.col.begin();
        return r < e && A.col[r] == (int32_t)i && kept[(size_t)r] != 0;
    };
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            int64_t cnt = 0;
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) cnt += stays(i, q) ? 1 : 0;
            G.ptr[(size_t)i + 1] = cnt;
        }
    }, 1 << 12);
    for (int32_t i = 0; i < n; i++) G.ptr[(size_t)i + 1] += G.ptr[(size_t)i];
    G.col.resize((size_t)G.ptr[n]);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            int64_t w = G.ptr[i];
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++)
                if (stays(i, q)) G.col[(size_t)w++] = A.col[q];
        }
    }, 1 << 12);
    return &G;
}

// FEMSHELL_AMG_AGG_KEEP: neighbours per node the aggregation looks at (default 12 -- the graphs of the 4M-triangle north-star
// meshes, 11.9 neighbours per node on their coarse levels, keep their aggregates bit for bit; 0: all of them, as in rounds 2-4)
static int aggregation_keep()
{
    const char *e = getenv("FEMSHELL_AMG_AGG_KEEP"); // (read per setup: the tests switch it inside one process)
    return e && *e ? atoi(e) : 12;
}

// FEMSHELL_AMG_AGG_CHUNK: rows per chunk of the chunked aggregation below (default 0: the whole graph in one piece)
static int64_t aggregation_chunk()
{
    const char *e = getenv("FEMSHELL_AMG_AGG_CHUNK"); // (read per setup: the tests switch it inside one process)
    return e && *e ? atoll(e) : 0;
}

static int32_t aggregate_piece(const Bsr &Afull, std::vector<int32_t> *aggout, const std::vector<int32_t> *visit);

// Would aggregate_nodes run its greedy passes on the graph itself (no row beyond the filter's width), in one piece, in index order
// (or in the caller's visiting order)?  Then the passes need no sorted lists and amg_device_setup.cpp runs them on the operator's
// ELL pattern directly (aggregate_on_pattern) instead of building the graph first.
bool aggregation_is_plain(int32_t n, int64_t widest_row, int64_t edges, double neighbour_distance_sum, bool have_visit)
{
    const int64_t chunk = aggregation_chunk();
    if (chunk > 0 && (int64_t)n > chunk + chunk / 2) return false;
    const int keep = aggregation_keep();
    if (keep > 0 && widest_row > (int64_t)keep + 1) return false;
    if (have_visit || n < 64) return true;
    const char *force = getenv("FEMSHELL_AMG_AGG_ORDER");
    if (force && std::strcmp(force, "index") == 0) return true;
    if (force && std::strcmp(force, "bfs") == 0) return false;
    return edges == 0 || neighbour_distance_sum / (double)edges <= 8.0 * std::sqrt((double)n);
}

// The greedy passes are sequential sweeps on one host thread.  MEASURED AND NOT ADOPTED (round 6, profiles/r06_chunked_aggregation.txt):
// cut into pieces of 131072 rows and aggregated on sixteen threads the passes of a 4M-triangle mesh take 3 ms instead of 7 -- the
// lap "graph + aggregation" of round 5 was mostly the graph -- and the seams, where the tiling of a structured mesh starts anew,
// cost iterations: panel 100 -> 106, pinched cylinder 83 -> 107, flap 37 -> 40.  Off by default (FEMSHELL_AMG_AGG_CHUNK=0); with
// a chunk set, graphs of more than one and a half chunks are cut into pieces of about `chunk` consecutive rows (whole
// slices of 32; the same boundaries as a row partition over that many ranks: partition_rows), every piece is aggregated on its
// own -- without the edges that leave it, exactly as the ranks of a row-partitioned level aggregate theirs (amg_dist.cpp) -- on a
// host thread of its own, and the aggregates are numbered piece by piece.  The pieces depend on the row count alone, never on the
// number of threads.  Restated by oracle/amg_oracle.py aggregate (aggregate_by_rank over chunk_bounds).
int32_t aggregate_nodes(const Bsr &Afull, std::vector<int32_t> *aggout, const std::vector<int32_t> *visit)
{
    const int64_t chunk = aggregation_chunk();
    const int32_t n = Afull.nr;
    if (chunk <= 0 || (int64_t)n <= chunk + chunk / 2) return aggregate_piece(Afull, aggout, visit);
    const int nc = (int)(((int64_t)n + chunk - 1) / chunk);
    std::vector<int32_t> bound((size_t)nc + 1);
    for (int k = 0; k <= nc; k++) {
        int32_t b, e;
        partition_rows(n, nc, k < nc ? k : nc - 1, &b, &e);
        bound[(size_t)k] = k < nc ? b : e;
    }
    // the visiting order of a piece: the caller's order restricted to it
    std::vector<std::vector<int32_t>> vis;
    const bool have_visit = visit != nullptr && (int32_t)visit->size() == n;
    if (have_visit) {
        vis.resize((size_t)nc);
        for (int k = 0; k < nc; k++) vis[(size_t)k].reserve((size_t)(bound[(size_t)k + 1] - bound[(size_t)k]));
        std::vector<int32_t> piece_of_slice(((size_t)n + kSliceNodes - 1) / kSliceNodes);
        for (int k = 0; k < nc; k++)
            for (int32_t sl = bound[(size_t)k] / kSliceNodes; sl < (bound[(size_t)k + 1] + kSliceNodes - 1) / kSliceNodes; sl++) piece_of_slice[(size_t)sl] = k;
        for (int32_t v : *visit) {
            const int k = piece_of_slice[(size_t)(v / kSliceNodes)];
            vis[(size_t)k].push_back(v - bound[(size_t)k]);
        }
    }
    std::vector<std::vector<int32_t>> part((size_t)nc);
    std::vector<int32_t> count((size_t)nc, 0);
    parallel_chunks(nc, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; k++) {
            const int32_t b = bound[(size_t)k], e = bound[(size_t)k + 1];
            Bsr S;
            S.nr = S.nc = e - b;
            S.ptr.assign((size_t)(e - b) + 1, 0);
            S.col.reserve((size_t)(Afull.ptr[e] - Afull.ptr[b]));
            for (int32_t i = b; i < e; i++) {
                for (int64_t q = Afull.ptr[i]; q < Afull.ptr[(size_t)i + 1]; q++) {
                    const int32_t j = Afull.col[q];
                    if (j >= b && j < e) S.col.push_back(j - b);
                }
                S.ptr[(size_t)(i - b) + 1] = (int64_t)S.col.size();
            }
            count[(size_t)k] = aggregate_piece(S, &part[(size_t)k], have_visit ? &vis[(size_t)k] : nullptr);
        }
    }, 1);
    std::vector<int32_t> first((size_t)nc + 1, 0);
    for (int k = 0; k < nc; k++) first[(size_t)k + 1] = first[(size_t)k] + count[(size_t)k];
    aggout->resize((size_t)n);
    parallel_chunks(nc, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; k++) {
            const int32_t b = bound[(size_t)k], off = first[(size_t)k];
            const std::vector<int32_t> &a = part[(size_t)k];
            for (size_t i = 0; i < a.size(); i++) (*aggout)[(size_t)b + i] = a[i] + off;
        }
    }, 1);
    return first[(size_t)nc];
}

static int32_t aggregate_piece(const Bsr &Afull, std::vector<int32_t> *aggout, const std::vector<int32_t> *visit)
{
    Bsr filtered;
    const Bsr &A = *graph_for_aggregation(Afull, aggregation_keep(), &filtered);
    const int32_t n = A.nr;
    std::vector<int32_t> agg((size_t)n, -1);
    int32_t na = 0;
    std::vector<int32_t> order;
    if (visit != nullptr && (int32_t)visit->size() == n) order = *visit;
    else aggregation_order(A, &order);
    // pass 1: a node whose whole neighbourhood is free becomes the root of a new aggregate
    for (int32_t v = 0; v < n; v++) {
        const int32_t i = order.empty() ? v : order[(size_t)v];
        if (agg[i] >= 0) continue;
        const int64_t b = A.ptr[i], e = A.ptr[i + 1];
        if (e - b <= 1) continue;
        bool free_nb = true;
        for (int64_t q = b; q < e && free_nb; q++) free_nb = agg[A.col[q]] < 0;
        if (!free_nb) continue;
        for (int64_t q = b; q < e; q++) agg[A.col[q]] = na;
        na++;
    }
    // pass 2: leftovers join the aggregate of their first aggregated neighbour (state of pass 1) -- "first" in the visiting
    // order: with index order that is the lowest column, as the columns ascend; with a visiting order of its own (the
    // caller's numbering under a library renumbering) the choice must not fall back on the internal column order, or the
    // leftovers scatter over their neighbours' aggregates (4M-triangle cylinder, Morton numbering: 316 iterations)
    std::vector<int32_t> rank;
    if (!order.empty()) {
        rank.resize((size_t)n);
        for (int32_t v = 0; v < n; v++) rank[(size_t)order[(size_t)v]] = v;
    }
    RawVec<int32_t> agg2((size_t)n);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) { // (reads the state of pass 1 only: the rows are independent)
        for (int64_t i = i0; i < i1; i++) {
            agg2[(size_t)i] = agg[(size_t)i];
            if (agg[(size_t)i] >= 0) continue;
            int32_t best = -1, best_rank = 0;
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
                const int32_t j = A.col[q];
                if (agg[j] < 0) continue;
                const int32_t rj = rank.empty() ? j : rank[(size_t)j];
                if (best < 0 || rj < best_rank) {
                    best = agg[j];
                    best_rank = rj;
                }
                if (rank.empty()) break; // ascending columns: the first hit is the lowest
            }
            if (best >= 0) agg2[(size_t)i] = best;
        }
    }, 1 << 14);
    parallel_chunks(n, [&](int64_t i0, int64_t i1) { std::copy(agg2.begin() + i0, agg2.begin() + i1, agg.begin() + i0); }, 1 << 16);
    // pass 3: what is still free forms aggregates of its own
    for (int32_t v = 0; v < n; v++) {
        const int32_t i = order.empty() ? v : order[(size_t)v];
        if (agg[i] >= 0) continue;
        for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++)
            if (agg[A.col[q]] < 0) agg[A.col[q]] = na;
        agg[i] = na;
        na++;
    }
    aggout->swap(agg);
    return na;
}

// ---- clusters of rigidly coupled nodes (amg_patch.hpp) ---------------------------------------------------------------

int32_t patch_clusters(int32_t n, std::vector<PatchEdge> edges, int max_nodes, std::vector<int32_t> *label, std::vector<int32_t> *ptr,
                       std::vector<int32_t> *nodes)
{
    if (max_nodes > kPatchMaxNodes) max_nodes = kPatchMaxNodes;
    for (auto &e : edges)
        if (e.a > e.c) std::swap(e.a, e.c);
    std::sort(edges.begin(), edges.end(), [](const PatchEdge &x, const PatchEdge &y) {
        if (x.sigma2 != y.sigma2) return x.sigma2 > y.sigma2;
        if (x.a != y.a) return x.a < y.a;
        return x.c < y.c;
    });
    std::vector<int32_t> parent((size_t)n), size((size_t)n, 1);
    for (int32_t i = 0; i < n; i++) parent[(size_t)i] = i;
    auto find = [&](int32_t a) {
        while (parent[(size_t)a] != a) {
            parent[(size_t)a] = parent[(size_t)parent[(size_t)a]];
            a = parent[(size_t)a];
        }
        return a;
    };
    for (const PatchEdge &e : edges) {
        if (e.a < 0 || e.c >= n || e.a == e.c) continue;
        int32_t ra = find(e.a), rc = find(e.c);
        if (ra == rc || size[(size_t)ra] + size[(size_t)rc] > max_nodes) continue;
        if (rc < ra) std::swap(ra, rc); // the root is the cluster's smallest node
        parent[(size_t)rc] = ra;
        size[(size_t)ra] += size[(size_t)rc];
    }
    label->assign((size_t)n, -1);
    ptr->assign(1, 0);
    nodes->clear();
    int32_t nc = 0;
    // roots ascending = clusters by their smallest node; members in ascending order
    std::vector<int32_t> id_of_root((size_t)n, -1);
    for (int32_t i = 0; i < n; i++) {
        const int32_t r = find(i);
        if (size[(size_t)r] < 2) continue;
        if (id_of_root[(size_t)r] < 0) id_of_root[(size_t)r] = nc++; // (r == its smallest member: met first)
        (*label)[(size_t)i] = id_of_root[(size_t)r];
    }
    std::vector<int32_t> cnt((size_t)nc + 1, 0);
    for (int32_t i = 0; i < n; i++)
        if ((*label)[(size_t)i] >= 0) cnt[(size_t)(*label)[(size_t)i] + 1]++;
    for (int32_t k = 0; k < nc; k++) cnt[(size_t)k + 1] += cnt[(size_t)k];
    *ptr = cnt;
    nodes->resize((size_t)cnt[(size_t)nc]);
    std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
    for (int32_t i = 0; i < n; i++)
        if ((*label)[(size_t)i] >= 0) (*nodes)[(size_t)fill[(size_t)(*label)[(size_t)i]]++] = i;
    return nc;
}

void patch_edges_host(const Bsr &A, const std::vector<double> &Dinv, double tau, std::vector<PatchEdge> *edges)
{
    edges->clear();
    const double tau2 = tau * tau;
    for (int32_t i = 0; i < A.nr; i++)
        for (int64_t q = A.ptr[(size_t)i]; q < A.ptr[(size_t)i + 1]; q++) {
            const int32_t j = A.col[(size_t)q];
            if (j <= i || j >= A.nr) continue;
            const double s2 = patch_sigma2(&Dinv[(size_t)i * 36], &A.val[(size_t)q * 36], &Dinv[(size_t)j * 36], tau2);
            if (s2 > tau2) edges->push_back(PatchEdge{i, j, s2});
        }
}

int32_t patch_matrices(const std::vector<int32_t> &ptr, const std::vector<int64_t> &moff, const double *dinv_of_member, double *Bc)
{
    const int32_t nc = (int32_t)ptr.size() - 1;
    std::atomic<int32_t> fell_back{0};
    parallel_chunks(nc, [&](int64_t c0, int64_t c1) {
        std::vector<double> L, Li, inv;
        for (int64_t c = c0; c < c1; c++) {
            const int m = ptr[(size_t)c + 1] - ptr[(size_t)c], N = 6 * m;
            double *B = Bc + moff[(size_t)c];
            // Cholesky B = L L^T, inverse by two triangular solves
            L.assign((size_t)N * N, 0.0);
            bool ok = true;
            for (int j = 0; j < N && ok; j++) {
                double d = B[(size_t)j * N + j];
                for (int k = 0; k < j; k++) d -= L[(size_t)j * N + k] * L[(size_t)j * N + k];
                if (!(d > 0.0)) {
                    ok = false;
                    break;
                }
                const double ljj = std::sqrt(d);
                L[(size_t)j * N + j] = ljj;
                for (int i = j + 1; i < N; i++) {
                    double v = 0.5 * (B[(size_t)i * N + j] + B[(size_t)j * N + i]);
                    for (int k = 0; k < j; k++) v -= L[(size_t)i * N + k] * L[(size_t)j * N + k];
                    L[(size_t)i * N + j] = v / ljj;
                }
            }
            if (!ok) {
                fell_back.fetch_add(1);
                std::fill(B, B + (size_t)N * N, 0.0);
                continue;
            }
            Li.assign((size_t)N * N, 0.0); // L^-1, lower triangular
            for (int j = 0; j < N; j++) {
                Li[(size_t)j * N + j] = 1.0 / L[(size_t)j * N + j];
                for (int i = j + 1; i < N; i++) {
                    double v = 0.0;
                    for (int k = j; k < i; k++) v -= L[(size_t)i * N + k] * Li[(size_t)k * N + j];
                    Li[(size_t)i * N + j] = v / L[(size_t)i * N + i];
                }
            }
            for (int i = 0; i < N; i++)
                for (int j = 0; j <= i; j++) {
                    double v = 0.0;
                    for (int k = i; k < N; k++) v += Li[(size_t)k * N + i] * Li[(size_t)k * N + j];
                    B[(size_t)i * N + j] = B[(size_t)j * N + i] = v; // B^-1 = L^-T L^-1
                }
            for (int t = 0; t < m; t++) { // minus the point blocks
                const double *d = dinv_of_member + (size_t)(ptr[(size_t)c] + t) * 36;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) B[(size_t)(6 * t + i) * N + 6 * t + j] -= d[6 * i + j];
            }
        }
    }, 16);
    return fell_back.load();
}

int32_t aggregate_nodes_glued(const Bsr &A, const std::vector<int32_t> &label, std::vector<int32_t> *aggout, const std::vector<int32_t> *visit)
{
    const int32_t n = A.nr;
    int32_t nclusters = 0;
    for (int32_t i = 0; i < n; i++) nclusters = std::max(nclusters, label[(size_t)i] + 1);
    if (nclusters == 0) return aggregate_nodes(A, aggout, visit);
    // quotient nodes in the order the visiting order meets them
    std::vector<int32_t> qid((size_t)n, -1), q_of_cluster((size_t)nclusters, -1);
    int32_t nq = 0;
    const bool have_visit = visit != nullptr && (int32_t)visit->size() == n;
    for (int32_t v = 0; v < n; v++) {
        const int32_t i = have_visit ? (*visit)[(size_t)v] : v;
        const int32_t c = label[(size_t)i];
        if (c < 0) qid[(size_t)i] = nq++;
        else {
            if (q_of_cluster[(size_t)c] < 0) q_of_cluster[(size_t)c] = nq++;
            qid[(size_t)i] = q_of_cluster[(size_t)c];
        }
    }
    // members of the quotient nodes, then the quotient graph (rows sorted, the node itself included)
    std::vector<int32_t> mptr((size_t)nq + 1, 0), mem((size_t)n);
    for (int32_t i = 0; i < n; i++) mptr[(size_t)qid[(size_t)i] + 1]++;
    for (int32_t q = 0; q < nq; q++) mptr[(size_t)q + 1] += mptr[(size_t)q];
    {
        std::vector<int32_t> fill(mptr.begin(), mptr.end() - 1);
        for (int32_t i = 0; i < n; i++) mem[(size_t)fill[(size_t)qid[(size_t)i]]++] = i;
    }
    Bsr Q;
    Q.nr = Q.nc = nq;
    Q.ptr.assign((size_t)nq + 1, 0);
    std::vector<std::vector<int32_t>> rows((size_t)nq);
    parallel_chunks(nq, [&](int64_t q0, int64_t q1) {
        for (int64_t q = q0; q < q1; q++) {
            std::vector<int32_t> &r = rows[(size_t)q];
            for (int32_t t = mptr[(size_t)q]; t < mptr[(size_t)q + 1]; t++) {
                const int32_t i = mem[(size_t)t];
                for (int64_t e = A.ptr[(size_t)i]; e < A.ptr[(size_t)i + 1]; e++) r.push_back(qid[(size_t)A.col[(size_t)e]]);
            }
            r.push_back((int32_t)q);
            std::sort(r.begin(), r.end());
            r.erase(std::unique(r.begin(), r.end()), r.end());
        }
    }, 1 << 10);
    for (int32_t q = 0; q < nq; q++) Q.ptr[(size_t)q + 1] = Q.ptr[(size_t)q] + (int64_t)rows[(size_t)q].size();
    Q.col.resize((size_t)Q.ptr[(size_t)nq]);
    for (int32_t q = 0; q < nq; q++) std::copy(rows[(size_t)q].begin(), rows[(size_t)q].end(), Q.col.begin() + Q.ptr[(size_t)q]);
    std::vector<int32_t> aq;
    const int32_t na = aggregate_nodes(Q, &aq);
    aggout->resize((size_t)n);
    for (int32_t i = 0; i < n; i++) (*aggout)[(size_t)i] = aq[(size_t)qid[(size_t)i]];
    return na;
}

void tentative_prolongator(const std::vector<int32_t> &agg, int32_t na, const std::vector<double> &B,
                           std::vector<double> *Qout, std::vector<double> *Bcout)
{
    const int32_t n = (int32_t)agg.size();
    std::vector<double> &Q = *Qout, &Bc = *Bcout;
    Q.assign((size_t)n * 36, 0.0);
    Bc.assign((size_t)na * 36, 0.0);
    // nodes grouped by aggregate, ascending node id inside
    std::vector<int32_t> ptr((size_t)na + 1, 0), order((size_t)n);
    for (int32_t i = 0; i < n; i++) ptr[agg[i] + 1]++;
    for (int32_t a = 0; a < na; a++) ptr[a + 1] += ptr[a];
    {
        std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
        for (int32_t i = 0; i < n; i++) order[fill[agg[i]]++] = i;
    }
    parallel_chunks(na, [&](int64_t a0, int64_t a1) {
        std::vector<double> M, v;
        for (int64_t a = a0; a < a1; a++) {
            const int32_t k = ptr[a + 1] - ptr[a], rows = 6 * k;
            M.assign((size_t)rows * 6, 0.0); // column-major: M[j*rows + r]
            for (int32_t t = 0; t < k; t++) {
                const double *b = &B[(size_t)order[ptr[a] + t] * 36];
                for (int d = 0; d < 6; d++)
                    for (int m = 0; m < 6; m++) M[(size_t)m * rows + 6 * t + d] = b[6 * d + m];
            }
            double R[36] = {0};
            v.resize(rows);
            for (int j = 0; j < 6; j++) {
                double *cj = &M[(size_t)j * rows];
                double n0 = 0.0;
                for (int r = 0; r < rows; r++) n0 += cj[r] * cj[r];
                n0 = std::sqrt(n0);
                for (int pass = 0; pass < 2; pass++)
                    for (int i = 0; i < j; i++) {
                        const double *qi = &M[(size_t)i * rows];
                        double c = 0.0;
                        for (int r = 0; r < rows; r++) c += qi[r] * cj[r];
                        for (int r = 0; r < rows; r++) cj[r] -= c * qi[r];
                        R[6 * i + j] += c;
                    }
                double nj = 0.0;
                for (int r = 0; r < rows; r++) nj += cj[r] * cj[r];
                nj = std::sqrt(nj);
                if (n0 > 0.0 && nj > 1e-8 * n0) {
                    R[6 * j + j] = nj;
                    for (int r = 0; r < rows; r++) cj[r] /= nj;
                } else {
                    R[6 * j + j] = 0.0; // dependent column: no coarse dof here
                    for (int r = 0; r < rows; r++) cj[r] = 0.0;
                }
            }
            for (int32_t t = 0; t < k; t++) {
                double *q = &Q[(size_t)order[ptr[a] + t] * 36];
                for (int d = 0; d < 6; d++)
                    for (int m = 0; m < 6; m++) q[6 * d + m] = M[(size_t)m * rows + 6 * t + d];
            }
            std::memcpy(&Bc[(size_t)a * 36], R, sizeof R);
        }
    }, 16);
}

void block_diagonal_inverse(const Bsr &A, std::vector<double> *Dout)
{
    std::vector<double> &D = *Dout;
    D.assign((size_t)A.nr * 36, 0.0);
    parallel_chunks(A.nr, [&](int64_t r0, int64_t r1) {
        for (int64_t i = r0; i < r1; i++) {
            double *inv = &D[(size_t)i * 36];
            bool ok = false;
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++)
                if (A.col[q] == i) {
                    ok = spd_inverse6(&A.val[(size_t)q * 36], inv);
                    break;
                }
            if (!ok) {
                std::fill(inv, inv + 36, 0.0);
                for (int d = 0; d < 6; d++) inv[7 * d] = 1.0;
            }
        }
    });
}

void bsr_multiply(const Bsr &A, const Bsr &B, Bsr *Cout)
{
    Bsr &C = *Cout;
    C = Bsr();
    C.nr = A.nr;
    C.nc = B.nc;
    const int64_t n = A.nr;
    // pass 1 (symbolic): distinct columns per row; pass 2 (numeric) writes every row straight into its place --
    // both parallel over row chunks, nothing gigabyte-sized is touched by one thread alone
    std::vector<int32_t> cnt((size_t)n, 0);
    parallel_chunks(n, [&](int64_t r0, int64_t r1) {
        std::vector<int32_t> marker((size_t)B.nc, -1);
        for (int64_t i = r0; i < r1; i++) {
            int32_t c = 0;
            for (int64_t qa = A.ptr[i]; qa < A.ptr[i + 1]; qa++) {
                const int32_t k = A.col[qa];
                for (int64_t qb = B.ptr[k]; qb < B.ptr[k + 1]; qb++)
                    if (marker[B.col[qb]] != (int32_t)i) {
                        marker[B.col[qb]] = (int32_t)i;
                        c++;
                    }
            }
            cnt[i] = c;
        }
    }, 1024);
    C.ptr.assign((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) C.ptr[i + 1] = C.ptr[i] + cnt[i];
    C.col.resize((size_t)C.ptr[n]);
    C.val.resize((size_t)C.ptr[n] * 36);
    parallel_chunks(n, [&](int64_t r0, int64_t r1) {
        std::vector<int32_t> marker((size_t)B.nc, -1), list;
        std::vector<int32_t> rank;
        for (int64_t i = r0; i < r1; i++) {
            list.clear();
            for (int64_t qa = A.ptr[i]; qa < A.ptr[i + 1]; qa++) {
                const int32_t k = A.col[qa];
                for (int64_t qb = B.ptr[k]; qb < B.ptr[k + 1]; qb++)
                    if (marker[B.col[qb]] < 0) {
                        marker[B.col[qb]] = 0;
                        list.push_back(B.col[qb]);
                    }
            }
            std::sort(list.begin(), list.end());
            const int64_t base = C.ptr[i];
            for (size_t s2 = 0; s2 < list.size(); s2++) {
                marker[list[s2]] = (int32_t)s2 + 1; // position + 1 in the row
                C.col[(size_t)base + s2] = list[s2];
            }
            double *out = &C.val[(size_t)base * 36];
            std::fill(out, out + list.size() * 36, 0.0);
            for (int64_t qa = A.ptr[i]; qa < A.ptr[i + 1]; qa++) {
                const int32_t k = A.col[qa];
                const double *a = &A.val[(size_t)qa * 36];
                for (int64_t qb = B.ptr[k]; qb < B.ptr[k + 1]; qb++)
                    blk_mac(a, &B.val[(size_t)qb * 36], out + (size_t)(marker[B.col[qb]] - 1) * 36);
            }
            for (int32_t j : list) marker[j] = -1;
        }
    }, 1024);
}

void bsr_transpose(const Bsr &A, Bsr *Tout)
{
    Bsr &T = *Tout;
    T = Bsr();
    T.nr = A.nc;
    T.nc = A.nr;
    T.ptr.assign((size_t)T.nr + 1, 0);
    for (int32_t c : A.col) T.ptr[c + 1]++;
    for (int32_t i = 0; i < T.nr; i++) T.ptr[i + 1] += T.ptr[i];
    T.col.resize(A.col.size());
    T.val.resize(A.val.size());
    std::vector<int64_t> src(A.col.size()); // block of A that lands in each block of T
    {
        std::vector<int64_t> fill(T.ptr.begin(), T.ptr.end() - 1);
        for (int32_t i = 0; i < A.nr; i++) // ascending rows -> ascending columns of the transpose
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
                const int64_t d = fill[A.col[q]]++;
                T.col[(size_t)d] = i;
                src[(size_t)d] = q;
            }
    }
    parallel_chunks((int64_t)src.size(), [&](int64_t d0, int64_t d1) {
        for (int64_t d = d0; d < d1; d++) {
            const double *s2 = &A.val[(size_t)src[(size_t)d] * 36];
            double *t = &T.val[(size_t)d * 36];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) t[6 * c + r] = s2[6 * r + c];
        }
    }, 4096);
}

void smoothed_prolongator(const Bsr &A, const std::vector<double> &Dinv, const std::vector<int32_t> &agg, int32_t na,
                          const std::vector<double> &Q, double omega, Bsr *Pout)
{
    Bsr P0;
    P0.nr = A.nr;
    P0.nc = na;
    P0.ptr.resize((size_t)A.nr + 1);
    for (int64_t i = 0; i <= A.nr; i++) P0.ptr[i] = i;
    P0.col.assign(agg.begin(), agg.end());
    P0.val.assign(Q.begin(), Q.end());
    Bsr &P = *Pout;
    bsr_multiply(A, P0, &P); // pattern of P = pattern of A P0 (it contains (i, agg[i]): A has its diagonal)
    parallel_chunks(A.nr, [&](int64_t r0, int64_t r1) {
        double t[36];
        for (int64_t i = r0; i < r1; i++) {
            const double *di = &Dinv[(size_t)i * 36];
            for (int64_t q = P.ptr[i]; q < P.ptr[i + 1]; q++) {
                double *v = &P.val[(size_t)q * 36];
                std::fill(t, t + 36, 0.0);
                blk_mac(di, v, t);
                const bool own = P.col[q] == agg[i];
                for (int e = 0; e < 36; e++) v[e] = (own ? Q[(size_t)i * 36 + e] : 0.0) - omega * t[e];
            }
        }
    });
}

void galerkin_product(const Bsr &A, const Bsr &P, Bsr *R, Bsr *Ac)
{
    Bsr AP;
    bsr_multiply(A, P, &AP);
    bsr_transpose(P, R);
    bsr_multiply(*R, AP, Ac);
    // coarse dofs without fine support (zero column of P): unit diagonal keeps the level matrix SPD
    for (int32_t i = 0; i < Ac->nr; i++)
        for (int64_t q = Ac->ptr[i]; q < Ac->ptr[i + 1]; q++)
            if (Ac->col[q] == i) {
                double *d = &Ac->val[(size_t)q * 36];
                for (int v = 0; v < 6; v++)
                    if (d[7 * v] == 0.0) d[7 * v] = 1.0;
            }
}

bool dense_inverse(const Bsr &A, std::vector<double> *invout)
{
    const int64_t n = 6ll * A.nr;
    std::vector<double> L((size_t)(n * n), 0.0);
    for (int32_t i = 0; i < A.nr; i++)
        for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
            const int32_t j = A.col[q];
            const double *v = &A.val[(size_t)q * 36];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) L[(size_t)(6 * i + r) * n + 6 * j + c] = v[6 * r + c];
        }
    // symmetrise (the Galerkin product is symmetric up to rounding), then Cholesky in the lower triangle.
    // Semi-definite operators are legitimate: the reference's Test A/B cantilevers fix u,v,w at three collinear nodes
    // only, so a rigid rotation about that line has no stiffness, and the reference's Krylov solve copes because the
    // loads do not excite it.  A pivot that has lost eleven digits against its diagonal entry marks such a direction:
    // it is dropped (zero row and column of the inverse), which makes the result the inverse on the complement -- what
    // a preconditioner needs.  A clearly negative pivot is a real failure.
    // How many digits the operator itself carries: it is a chain of Galerkin products, in which the rigid-body modes of
    // the stiff bending part (entries of order D / h^2) cancel down to the stiffness of the coarse modes.  On the
    // 250k-triangle flap of the coupled example (h = 4e-4) the softest global mode of the coarsest operator -- one of the
    // last pivots -- came out at -4e-7 of its diagonal entry, with the product symmetric to 2e-13 (the rounding errors of
    // mirrored entries are the same, so the asymmetry does not show them).  Such a pivot is noise: the direction is
    // dropped like a semi-definite one and the outer Krylov iteration takes care of the mode.  A pivot that is negative
    // beyond 1e-4 of its diagonal entry is a real failure.
    for (int64_t r = 0; r < n; r++)
        for (int64_t c = 0; c < r; c++) L[r * n + c] = 0.5 * (L[r * n + c] + L[c * n + r]);
    std::vector<char> dead((size_t)n, 0);
    for (int64_t c = 0; c < n; c++) {
        const double a_cc = L[c * n + c];
        double d = a_cc;
        for (int64_t k = 0; k < c; k++) d -= L[c * n + k] * L[c * n + k];
        if (d < -1e-4 * std::fabs(a_cc) || !(a_cc > 0.0)) {
            if (getenv("FEMSHELL_AMG_VERBOSE"))
                fprintf(stderr, "[femshell amg setup] dense inverse: pivot %lld of %lld: d = %.3e, diagonal %.3e (ratio %.2e)\n",
                        (long long)c, (long long)n, d, a_cc, d / a_cc);
            return false;
        }
        if (d <= 1e-11 * a_cc) {
            dead[(size_t)c] = 1;
            L[c * n + c] = 1.0;
            for (int64_t r = c + 1; r < n; r++) L[r * n + c] = 0.0;
            for (int64_t k = 0; k < c; k++) L[c * n + k] = 0.0;
            continue;
        }
        const double lcc = std::sqrt(d);
        L[c * n + c] = lcc;
        parallel_chunks(n - c - 1, [&](int64_t b0, int64_t e0) {
            for (int64_t r = c + 1 + b0; r < c + 1 + e0; r++) {
                double v = L[r * n + c];
                const double *lr = &L[r * n], *lc = &L[c * n];
                for (int64_t k = 0; k < c; k++) v -= lr[k] * lc[k];
                L[r * n + c] = v / lcc;
            }
        }, 64);
    }
    // Li = L^-1 column by column, then inv = Li^T Li
    std::vector<double> Li((size_t)(n * n), 0.0);
    parallel_chunks(n, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; c++) {
            if (dead[(size_t)c]) continue; // dropped direction: zero column
            Li[c * n + c] = 1.0 / L[c * n + c];
            for (int64_t r = c + 1; r < n; r++) {
                if (dead[(size_t)r]) continue; // ... and zero row
                double v = 0.0;
                const double *lr = &L[r * n];
                for (int64_t k = c; k < r; k++) v -= lr[k] * Li[k * n + c];
                Li[r * n + c] = v / lr[r];
            }
        }
    }, 8);
    std::vector<double> &inv = *invout;
    inv.assign((size_t)(n * n), 0.0);
    // rows of Li^T are columns of Li: inv(i,j) = sum_k Li(k,i) Li(k,j); use the transposed copy for locality
    std::vector<double> LiT((size_t)(n * n));
    for (int64_t r = 0; r < n; r++)
        for (int64_t c = 0; c <= r; c++) LiT[c * n + r] = Li[r * n + c];
    parallel_chunks(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++)
            for (int64_t j = 0; j <= i; j++) {
                double v = 0.0;
                const double *a = &LiT[i * n], *b2 = &LiT[j * n];
                for (int64_t k = i; k < n; k++) v += a[k] * b2[k];
                inv[i * n + j] = v;
                inv[j * n + i] = v;
            }
    }, 8);
    return true;
}

void pack_sliced_ell(const Bsr &A, bool diag_first, SlicedEll *out)
{
    SlicedEll &S = *out;
    S = SlicedEll();
    S.n_rows = A.nr;
    S.n_pad = (A.nr + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    S.n_slices = S.n_pad / kSliceNodes;
    S.slice_width.assign((size_t)S.n_slices, 1);
    S.slice_base.assign((size_t)S.n_slices + 1, 0);
    for (int32_t s = 0; s < S.n_slices; s++) {
        int w = 1;
        for (int n = 0; n < kSliceNodes; n++) {
            const int32_t a = s * kSliceNodes + n;
            if (a >= A.nr) continue;
            int cnt = (int)(A.ptr[a + 1] - A.ptr[a]);
            if (diag_first) { // slot 0 is reserved for the diagonal block even when the row has none
                bool have = false;
                for (int64_t q = A.ptr[a]; q < A.ptr[a + 1] && !have; q++) have = A.col[q] == a;
                if (!have) cnt++;
            }
            w = std::max(w, cnt);
        }
        S.slice_width[s] = w;
        S.max_width = std::max(S.max_width, w);
        S.slice_base[s + 1] = S.slice_base[s] + (int64_t)w * kSliceNodes;
    }
    const int64_t total = S.slice_base[S.n_slices];
    S.cols.assign((size_t)total, 0);
    S.vals.resize((size_t)total * 36);
    parallel_chunks(S.n_slices, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; s++) {
            const int w = S.slice_width[s];
            const int64_t base = S.slice_base[s];
            double *dst = &S.vals[(size_t)base * 36];
            std::fill(dst, dst + (size_t)w * kSliceNodes * 36, 0.0); // padding slots stay zero
            for (int n = 0; n < kSliceNodes; n++) {
                const int32_t a = (int32_t)s * kSliceNodes + n;
                const int32_t pad_col = diag_first ? std::min(a, S.n_pad - 1) : 0;
                int k = 0;
                auto put = [&](int64_t q) {
                    S.cols[(size_t)(base + (int64_t)k * kSliceNodes + n)] = A.col[q];
                    const double *v = &A.val[(size_t)q * 36];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++)
                            dst[((((int64_t)k * 3 + j / 2) * 6 + i) * kSliceNodes + n) * 2 + (j & 1)] = v[6 * i + j];
                    k++;
                };
                if (a < A.nr) {
                    if (diag_first) {
                        bool have = false;
                        for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++)
                            if (A.col[q] == a) {
                                put(q);
                                have = true;
                            }
                        if (!have) { // keep slot 0 for the (zero) diagonal block
                            S.cols[(size_t)(base + n)] = a;
                            k = 1;
                        }
                        for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++)
                            if (A.col[q] != a) put(q);
                    } else {
                        for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++) put(q);
                    }
                }
                for (; k < w; k++) S.cols[(size_t)(base + (int64_t)k * kSliceNodes + n)] = pad_col;
            }
        }
    }, 8);
}

void build_in_lists(int32_t n_rows, const std::vector<int32_t> &slice_width, const std::vector<int64_t> &slice_base,
                    const int32_t *cols, const std::vector<uint8_t> &count, SlicedEllSym *out)
{
    SlicedEllSym &S = *out;
    const int32_t n_slices = (int32_t)slice_width.size();
    const int32_t n_pad = n_slices * kSliceNodes;
    std::vector<int32_t> cnt((size_t)n_pad, 0);
    auto for_each_offdiag = [&](const std::function<void(int32_t a, int32_t c, int64_t slot)> &f) {
        for (int32_t s = 0; s < n_slices; s++)
            for (int k = 1; k < slice_width[s]; k++)
                for (int n = 0; n < kSliceNodes; n++) {
                    const int32_t a = s * kSliceNodes + n;
                    if (a >= n_rows || k >= count[a]) continue;
                    const int64_t slot = slice_base[s] + (int64_t)k * kSliceNodes + n;
                    if (cols[(size_t)slot] >= n_pad) continue; // ghost column of a row-partitioned level: no transpose on this rank
                    f(a, cols[(size_t)slot], slot);
                }
    };
    for_each_offdiag([&](int32_t, int32_t c, int64_t) { cnt[c]++; });
    S.in_width.assign((size_t)n_slices, 0);
    S.in_base.assign((size_t)n_slices + 1, 0);
    S.max_in_width = 0;
    for (int32_t s = 0; s < n_slices; s++) {
        int w = 0;
        for (int n = 0; n < kSliceNodes; n++) w = std::max(w, cnt[(size_t)s * kSliceNodes + n]);
        S.in_width[s] = w;
        S.max_in_width = std::max(S.max_in_width, w);
        S.in_base[s + 1] = S.in_base[s] + (int64_t)w * kSliceNodes;
    }
    S.in_slots.assign((size_t)S.in_base[n_slices], -1);
    S.in_rows.assign((size_t)S.in_base[n_slices], 0);
    std::fill(cnt.begin(), cnt.end(), 0);
    for_each_offdiag([&](int32_t a, int32_t c, int64_t slot) { // ascending slot index: a fixed order of the sums
        const size_t e = (size_t)(S.in_base[c / kSliceNodes] + (int64_t)cnt[c] * kSliceNodes + c % kSliceNodes);
        S.in_slots[e] = (int32_t)slot;
        S.in_rows[e] = a;
        cnt[c]++;
    });
}

void pack_sliced_ell_sym(const Bsr &A, SlicedEllSym *out)
{
    // the upper part as a matrix of its own, packed diagonal first
    Bsr U;
    U.nr = A.nr;
    U.nc = A.nc;
    U.ptr.assign((size_t)A.nr + 1, 0);
    for (int32_t a = 0; a < A.nr; a++) {
        int64_t c = 0;
        for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++) c += A.col[q] >= a;
        U.ptr[a + 1] = U.ptr[a] + c;
    }
    U.col.resize((size_t)U.ptr[A.nr]);
    U.val.resize((size_t)U.ptr[A.nr] * 36);
    parallel_chunks(A.nr, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            int64_t w = U.ptr[a];
            for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++)
                if (A.col[q] >= a) {
                    U.col[(size_t)w] = A.col[q];
                    std::memcpy(&U.val[(size_t)w * 36], &A.val[(size_t)q * 36], 36 * sizeof(double));
                    w++;
                }
        }
    });
    SlicedEll base;
    pack_sliced_ell(U, true, &base);
    static_cast<SlicedEll &>(*out) = std::move(base);
    std::vector<uint8_t> count((size_t)out->n_pad, 0);
    for (int32_t a = 0; a < A.nr; a++) count[a] = (uint8_t)std::min<int64_t>(255, U.ptr[a + 1] - U.ptr[a]);
    build_in_lists(A.nr, out->slice_width, out->slice_base, out->cols.data(), count, out);
}

void mirror_upper(Bsr *Aio)
{
    Bsr T;
    bsr_transpose(*Aio, &T);
    Bsr &U = *Aio;
    Bsr F;
    F.nr = U.nr;
    F.nc = U.nc;
    F.ptr.assign((size_t)U.nr + 1, 0);
    for (int32_t a = 0; a < U.nr; a++) {
        int64_t lower = 0;
        for (int64_t q = T.ptr[a]; q < T.ptr[a + 1]; q++) lower += T.col[q] < a; // (c, a) stored with c < a
        F.ptr[a + 1] = F.ptr[a] + lower + (U.ptr[a + 1] - U.ptr[a]);
    }
    F.col.resize((size_t)F.ptr[U.nr]);
    F.val.resize((size_t)F.ptr[U.nr] * 36);
    parallel_chunks(U.nr, [&](int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            int64_t w = F.ptr[a];
            for (int64_t q = T.ptr[a]; q < T.ptr[a + 1]; q++) // ascending lower columns (transpose rows are ascending)
                if (T.col[q] < a) {
                    F.col[(size_t)w] = T.col[q];
                    std::memcpy(&F.val[(size_t)w * 36], &T.val[(size_t)q * 36], 36 * sizeof(double));
                    w++;
                }
            for (int64_t q = U.ptr[a]; q < U.ptr[a + 1]; q++) {
                F.col[(size_t)w] = U.col[q];
                std::memcpy(&F.val[(size_t)w * 36], &U.val[(size_t)q * 36], 36 * sizeof(double));
                w++;
            }
        }
    });
    *Aio = std::move(F);
}

} // namespace femshell
