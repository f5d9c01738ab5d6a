import sys, time
import numpy as np, scipy.sparse as sp
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell
import amg_oracle as ao

def problem(npts, seed, qmin=0.0):
    xyz, tri = delaunay_shell(npts, seed)
    if qmin > 0:
        p, q, r = xyz[tri[:, 0]], xyz[tri[:, 1]], xyz[tri[:, 2]]
        area = 0.5 * np.linalg.norm(np.cross(q - p, r - p), axis=1)
        l2 = ((q - p) ** 2).sum(1) + ((r - q) ** 2).sum(1) + ((p - r) ** 2).sum(1)
        qual = 4.0 * np.sqrt(3.0) * area / l2
        tri = tri[qual >= qmin]
        used = np.unique(tri); remap = -np.ones(len(xyz), dtype=np.int64); remap[used] = np.arange(len(used))
        tri = remap[tri].astype(np.int32); xyz = xyz[used]
    n = len(xyz)
    dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
    loads = np.zeros((n, 6)); loads[:, 2] = 1.0
    mat = oracle.material(0.3, 7.0e4, 0.03)
    quad = np.zeros((0, 4), np.int32)
    r, c, v, F = oracle.assemble(xyz, tri, quad, mat, dirichlet=dmask, loads=loads)
    A = oracle.to_scipy(r, c, v).tobsr((6, 6))
    return xyz, tri, dmask, A, F

def strength_graph(A, theta):
    """block strength: ||A_ij||_F >= theta sqrt(||A_ii||_F ||A_jj||_F)"""
    A = A.tobsr((6, 6)); A.sort_indices()
    n = A.shape[0] // 6
    nrm = np.sqrt((A.data ** 2).sum(axis=(1, 2)))
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    dn = np.zeros(n); isd = A.indices == rows; dn[rows[isd]] = nrm[isd]
    keep = isd | (nrm >= theta * np.sqrt(dn[rows] * dn[A.indices]))
    cnt = np.bincount(rows[keep], minlength=n)
    return np.concatenate([[0], np.cumsum(cnt)]), A.indices[keep], keep.mean()

if __name__ == "__main__":
    npts, seed, qmin = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    xyz, tri, dmask, A, F = problem(npts, seed, qmin)
    print("nodes", len(xyz), "tri", len(tri), "qmin", qmin, flush=True)
    for theta in [0.0] + [float(x) for x in sys.argv[4:]]:
        orig = ao.aggregate
        if theta > 0:
            def agg_f(rowptr, colidx, _A=[None]):
                return orig(rowptr, colidx)
            # patch coarsen to aggregate on the filtered graph
            def coarsen_f(Am, B, lam, theta=theta):
                Am = Am.tobsr((6, 6)); Am.sort_indices()
                n = Am.shape[0] // 6
                rp, ci, frac = strength_graph(Am, theta)
                agg, na = orig(rp, ci)
                Q, Bc = ao.tentative(agg, na, B)
                P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
                Dm = ao.bd_matrix(ao.block_diag_inverse(Am))
                P = (P0 - ((4.0 / 3.0) / lam) * (Dm @ (Am @ P0))).tobsr((6, 6))
                Ac = (P.T @ (Am @ P)).tobsr((6, 6))
                d = Ac.diagonal()
                if np.any(d == 0.0):
                    Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
                return agg, P, Ac, Bc
            ao_coarsen = ao.coarsen; ao.coarsen = coarsen_f
        t0 = time.time()
        levels = ao.setup(A, xyz, dmask, coarsest_nodes=200, tri=tri)
        u, hist = ao.solve(A, F, levels, kcycle=True, rtol=1e-10, max_it=1500, refine_passes=0)
        print("theta %.3f levels %s iterations %d  (%.0f s)" % (theta, [L.n for L in levels], len(hist), time.time() - t0), flush=True)
        if theta > 0: ao.coarsen = ao_coarsen
