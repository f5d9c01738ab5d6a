"""numpy (CPU only), round 6: which parts of the cluster-block method of round 5's experiment does the poor-quality Delaunay shell need?
  full    cluster blocks in the level-0 smoother, in lambda_max and in the smoothing of P; clusters glued into aggregates
  smooth  cluster blocks in the level-0 smoother (and its lambda_max) only: P smoothed with the point blocks and their lambda; glued
  smooth- the same without gluing
usage: r06_patch_smoother_experiment.py <points> <seed> [tau=0.8] [max_cluster=6] [variants]"""
import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
from oracle import amg_oracle as ao
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell
src = open("tools/lab/cluster_smoother_experiment.py").read()
exec(src.split("def run(")[0].split('exec(open("tools/lab/glued_aggregation_experiment.py")')[0])
exec(open("tools/lab/glued_aggregation_experiment.py").read().split("def run(")[0])
exec("def clusters_of" + src.split("def clusters_of")[1].split("def run(")[0])


def run(A, F0, xyz, tri, dmask, tau, mc, variant, max_it=1000):
    t0 = time.time()
    lvl = [0]
    lab = clusters_of(A, tau, mc)
    Dm0 = cluster_block_inverse(A, lab)
    obd, oc = ao.bd_matrix, ao.coarsen
    first = [True]

    def bd_matrix(Dinv):
        if first[0]:
            first[0] = False
            return Dm0
        return obd(Dinv)

    def coarsen(Al, B, lam, bounds=None):
        Al = Al.tobsr((6, 6)); Al.sort_indices()
        n = Al.shape[0] // 6
        l = lvl[0]
        lvl[0] += 1
        if l == 0 and variant in ("full", "smooth"):
            agg, na, ncl, nn = glued_aggregate(Al, tau, mc)
        else:
            agg, na = ao.aggregate(Al.indptr, Al.indices)
        Q, Bc = ao.tentative(agg, na, B)
        P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
        if l == 0 and variant == "full":
            Dm, lam_p = Dm0, lam
        else:
            Dm = obd(ao.block_diag_inverse(Al))
            lam_p = lam if l > 0 else 1.1 * ao.lambda_max(Al, Dm)
        P = (P0 - ((4.0 / 3.0) / lam_p) * (Dm @ (Al @ P0))).tobsr((6, 6))
        Ac = (P.T @ (Al @ P)).tobsr((6, 6))
        d = Ac.diagonal()
        if np.any(d == 0.0): Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
        return agg, P, Ac, Bc
    ao.bd_matrix, ao.coarsen = bd_matrix, coarsen
    try:
        levels = ao.setup(A, xyz, dmask, tri=tri, coarsest_nodes=1400)
        x, hist = ao.solve(A, F0, levels, rtol=1e-10, max_it=max_it, refine_passes=1)
    finally:
        ao.bd_matrix, ao.coarsen = obd, oc
    sizes = np.bincount(np.unique(lab, return_inverse=True)[1])
    print("%-8s tau %.2f <= %d: clusters %d (nodes in them %d)  levels %-18s lam0 %.2f  its %4d  final %.1e  (%.0f s)" % (
        variant, tau, mc, int((sizes > 1).sum()), int(sizes[sizes > 1].sum()), [L.n for L in levels], levels[0].lam, len(hist), hist[-1], time.time() - t0), flush=True)


n_pts, seed = int(sys.argv[1]), int(sys.argv[2])
tau = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
mc = int(sys.argv[4]) if len(sys.argv) > 4 else 6
variants = sys.argv[5].split(",") if len(sys.argv) > 5 else ["full", "smooth", "smooth-"]
xyz, tri = delaunay_shell(n_pts, seed)
n = len(xyz)
dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
loads = np.zeros((n, 6)); loads[:, 2] = 1.0
r0, c0, v0, F0 = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), oracle.material(0.3, 7.0e4, 0.03), dmask, loads)
A = oracle.to_scipy(r0, c0, v0).tobsr((6, 6)); A.sort_indices()
for v in variants:
    run(A, F0, xyz, tri, dmask, tau, mc, v)
