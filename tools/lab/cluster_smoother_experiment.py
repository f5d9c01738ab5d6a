"""numpy: block-Jacobi smoother with one block per cluster of rigidly coupled nodes (12 x 12, 18 x 18) instead of per node, with and
without gluing the clusters into one aggregate -- random-point Delaunay shells (CPU only)."""
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
from oracle import amg_oracle as ao
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell
exec(open("tools/lab/glued_aggregation_experiment.py").read().split("if __name__")[0])

def clusters_of(Al, tau, max_cluster):
    A, rows, cols, smax = sigma_graph(Al)
    n = A.shape[0] // 6
    rigid = (rows != cols) & (smax > tau)
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
    return lab

def cluster_block_inverse(A, lab):
    A = A.tocsr()
    n = A.shape[0] // 6
    rows, cols, vals = [], [], []
    for c in np.unique(lab):
        nodes = np.flatnonzero(lab == c)
        idx = (6 * nodes[:, None] + np.arange(6)[None, :]).ravel()
        B = A[idx][:, idx].toarray()
        try:
            np.linalg.cholesky(B); Bi = np.linalg.inv(B)
        except np.linalg.LinAlgError:
            Bi = np.eye(len(idx))
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        rows.append(rr.ravel()); cols.append(cc.ravel()); vals.append(Bi.ravel())
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=A.shape)

def run(A, F0, xyz, tri, dmask, tau, glue, cluster_smoother, label, max_cluster=3):
    t0 = time.time()
    lvl = [0]
    orig_bd = ao.block_diag_inverse
    state = {}
    def coarsen(Al, B, lam, bounds=None):
        Al = Al.tobsr((6, 6)); Al.sort_indices()
        n = Al.shape[0] // 6
        if glue and lvl[0] == 0:
            agg, na, ncl, nn = glued_aggregate(Al, tau, max_cluster)
        else:
            agg, na = ao.aggregate(Al.indptr, Al.indices)
        lvl[0] += 1
        Q, Bc = ao.tentative(agg, na, B)
        P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
        Dm = state.get(id(Al)) if False else (state["Dm0"] if (cluster_smoother and lvl[0] == 1) else ao.bd_matrix(orig_bd(Al)))
        P = (P0 - ((4.0 / 3.0) / lam) * (Dm @ (Al @ P0))).tobsr((6, 6))
        Ac = (P.T @ (Al @ P)).tobsr((6, 6))
        d = Ac.diagonal()
        if np.any(d == 0.0): Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
        return agg, P, Ac, Bc
    oc, obd = ao.coarsen, ao.bd_matrix
    if cluster_smoother:
        lab = clusters_of(A, tau, max_cluster)
        state["Dm0"] = cluster_block_inverse(A, lab)
        ncl = len(np.unique(lab[np.bincount(lab, minlength=len(lab))[lab] > 1]))
        first = [True]
        def bd_matrix(Dinv):  # level 0 first: the cluster blocks
            if first[0]:
                first[0] = False
                return state["Dm0"]
            return obd(Dinv)
        ao.bd_matrix = bd_matrix
    ao.coarsen = coarsen
    try:
        levels = ao.setup(A, xyz, dmask, tri=tri, coarsest_nodes=1400)
        x, hist = ao.solve(A, F0, levels, rtol=1e-10, max_it=1000, refine_passes=1)
    finally:
        ao.coarsen, ao.bd_matrix = oc, obd
    print("%-58s levels %-14s lam0 %.2f its %4d final %.1e (%.0f s)" % (label, [L.n for L in levels], levels[0].lam, len(hist), hist[-1], time.time() - t0), flush=True)

n_pts = int(sys.argv[1]); seed = int(sys.argv[2])
xyz, tri = delaunay_shell(n_pts, seed)
n = len(xyz)
dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
loads = np.zeros((n, 6)); loads[:, 2] = 1.0
r0, c0, v0, F0 = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), oracle.material(0.3, 7.0e4, 0.03), dmask, loads)
A = oracle.to_scipy(r0, c0, v0).tobsr((6, 6)); A.sort_indices()
run(A, F0, xyz, tri, dmask, 0.0, False, False, "point blocks, plain aggregation")
run(A, F0, xyz, tri, dmask, 0.9, False, True, "cluster blocks (sigma > 0.90, <= 3 nodes), plain aggregation")
for tau, mc in ((0.9, 3), (0.8, 6), (0.6, 8)):
    run(A, F0, xyz, tri, dmask, tau, True, True, "cluster blocks (sigma > %.2f, <= %d nodes), glued aggregation" % (tau, mc), max_cluster=mc)
