"""Rigidly coupled node pairs glued into one aggregate before the greedy pass -- numpy experiment on the restatement
(oracle/amg_oracle.py), CPU only:  python tools/lab/glued_aggregation_experiment.py <points> <seed> [jittered]
Outcome (round 5): no gain on the random-point Delaunay shells, DESIGN section 0 row 8."""
import sys, time, numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import connected_components
sys.path.insert(0, ".")
from oracle import amg_oracle as ao
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell

def sigma_graph(A):
    A = A.tobsr((6, 6)); A.sort_indices()
    n = A.shape[0] // 6
    D = ao.block_diag(A)
    Li = np.zeros_like(D)
    for i in range(n):
        try: Li[i] = np.linalg.inv(np.linalg.cholesky(D[i]))
        except np.linalg.LinAlgError: Li[i] = np.eye(6)
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    S = np.einsum("eab,ebc,edc->ead", Li[rows], A.data, Li[A.indices])
    sv = np.linalg.svd(S, compute_uv=False)
    return A, rows, A.indices, sv[:, 0]

def glued_aggregate(Al, tau, max_cluster=6):
    A, rows, cols, smax = sigma_graph(Al)
    n = A.shape[0] // 6
    rigid = (rows != cols) & (smax > tau)
    # clusters: union of rigid edges, strongest first, bounded size
    order = np.argsort(-smax * rigid)
    parent = np.arange(n); size = np.ones(n, dtype=int)
    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for e in order:
        if not rigid[e]: break
        a, b = find(rows[e]), find(cols[e])
        if a != b and size[a] + size[b] <= max_cluster:
            parent[b] = a; size[a] += size[b]
    lab = np.array([find(i) for i in range(n)])
    uniq, cl = np.unique(lab, return_inverse=True)
    nc = len(uniq)
    # quotient graph
    C = sp.csr_matrix((np.ones(n), (np.arange(n), cl)), shape=(n, nc))
    G = sp.csr_matrix((np.ones(len(A.indices)), A.indices, A.indptr), shape=(n, n))
    Q = (C.T @ G @ C).tocsr(); Q.sort_indices()
    aq, na = ao.aggregate(Q.indptr, Q.indices)
    return aq[cl], na, int((size[uniq] > 1).sum()), int(size[uniq][size[uniq] > 1].sum())

def run(A, F0, xyz, tri, dmask, tau, levels_glued, label):
    t0 = time.time()
    lvl = [0]
    def coarsen(Al, B, lam, bounds=None):
        Al = Al.tobsr((6, 6)); Al.sort_indices()
        n = Al.shape[0] // 6
        if tau > 0 and lvl[0] < levels_glued:
            agg, na, ncl, nn = glued_aggregate(Al, tau)
            print("    level %d: %d clusters of %d nodes, %d aggregates" % (lvl[0], ncl, nn, na))
        else:
            agg, na = ao.aggregate(Al.indptr, Al.indices)
        lvl[0] += 1
        Q, Bc = ao.tentative(agg, na, B)
        P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
        Dm = ao.bd_matrix(ao.block_diag_inverse(Al))
        P = (P0 - ((4.0 / 3.0) / lam) * (Dm @ (Al @ P0))).tobsr((6, 6))
        Ac = (P.T @ (Al @ P)).tobsr((6, 6))
        d = Ac.diagonal()
        if np.any(d == 0.0): Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
        return agg, P, Ac, Bc
    oc = ao.coarsen
    ao.coarsen = coarsen
    try:
        levels = ao.setup(A, xyz, dmask, tri=tri, coarsest_nodes=1400)
        x, hist = ao.solve(A, F0, levels, rtol=1e-10, max_it=1000, refine_passes=1)
    finally:
        ao.coarsen = oc
    print("%-34s levels %-22s its %4d final %.1e (%.0f s)" % (label, [L.n for L in levels], len(hist), hist[-1], time.time() - t0), flush=True)

if __name__ == "__main__":
    n_pts = int(sys.argv[1]); seed = int(sys.argv[2])
    jit = len(sys.argv) > 3 and sys.argv[3] == "jittered"
    xyz, tri = delaunay_shell(n_pts, seed, jittered=jit)
    n = len(xyz)
    dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
    loads = np.zeros((n, 6)); loads[:, 2] = 1.0
    r0, c0, v0, F0 = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), oracle.material(0.3, 7.0e4, 0.03), dmask, loads)
    A = oracle.to_scipy(r0, c0, v0).tobsr((6, 6)); A.sort_indices()
    _, rows, cols, smax = sigma_graph(A)
    off = rows != cols
    print("sigma_max percentiles 50 90 99 99.9:", np.percentile(smax[off], [50, 90, 99, 99.9]))
    run(A, F0, xyz, tri, dmask, 0.0, 0, "plain")
    for tau in (0.9, 0.98, 0.995):
        run(A, F0, xyz, tri, dmask, tau, 1, "glued, tau %.3f, level 0" % tau)
    run(A, F0, xyz, tri, dmask, 0.98, 9, "glued, tau 0.98, all levels")
