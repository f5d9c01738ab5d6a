"""Strength-of-connection filtering of the aggregation graph on level 0, tried in the numpy restatement only (CPU, no GPU):
   python tools/lab/strength_experiment.py panel|cyl|roof NX  |  quads | strip | jit | sliver QMIN
Block (i, j) is kept when ||A_ij||_F >= theta sqrt(||A_ii||_F ||A_jj||_F); P is still smoothed with the full operator.
Findings: DESIGN.md section 5."""
import sys, time
import numpy as np, scipy.sparse as sp
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tests.helpers import oracle, meshes
import amg_oracle as ao
from strength_experiment_common import strength_graph
from tests.test_gpu_parity import delaunay_shell

def run(name, xyz, tri, quad, dmask, loads, mat, thetas):
    r, c, v, F = oracle.assemble(xyz, tri, quad, oracle.material(*mat), dirichlet=dmask, loads=loads)
    A = oracle.to_scipy(r, c, v).tobsr((6, 6))
    orig_aggregate = ao.aggregate
    for theta in thetas:
        cur = {"A": None}
        def coarsen_f(Am, B, lam, theta=theta):
            Am = Am.tobsr((6, 6)); Am.sort_indices()
            n = Am.shape[0] // 6
            first = (n == A.shape[0] // 6)
            rp, ci, frac = strength_graph(Am, theta) if (theta > 0 and first) else (Am.indptr, Am.indices, 1.0)
            agg, na = orig_aggregate(rp, ci)
            Q, Bc = ao.tentative(agg, na, B)
            P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
            Dm = ao.bd_matrix(ao.block_diag_inverse(Am))
            P = (P0 - ((4.0 / 3.0) / lam) * (Dm @ (Am @ P0))).tobsr((6, 6))
            Ac = (P.T @ (Am @ P)).tobsr((6, 6))
            d = Ac.diagonal()
            if np.any(d == 0.0):
                Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
            return agg, P, Ac, Bc
        save = ao.coarsen; ao.coarsen = coarsen_f
        t0 = time.time()
        levels = ao.setup(A, xyz, dmask, coarsest_nodes=200, tri=tri if len(tri) else None, quad=quad if len(quad) else None)
        u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=1500, refine_passes=0)
        cx = sum(L.A.nnz for L in levels) / levels[0].A.nnz
        print("%-28s theta %.2f levels %-26s complexity %.2f iterations %4d  (%.0f s)" % (name, theta, [L.n for L in levels], cx, len(hist), time.time() - t0), flush=True)
        ao.coarsen = save

thetas = [0.0, 0.15]
which = sys.argv[1]
NX = int(sys.argv[2]) if len(sys.argv) > 2 and which in ('panel','cyl','roof') else 64
if which == "panel":
    m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
    run("panel %d" % NX, m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, (0.3, 1e7, 0.5), thetas)
elif which == "roof":
    m = meshes.scordelis_lo(NX)
    run("roof %d" % NX, m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, m.material, thetas)
elif which == "cyl":
    m = meshes.pinched_cylinder(NX, NX)
    run("cylinder %d" % NX, m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, m.material, thetas)
elif which == "quads":
    m = meshes.structured(48, 48, 0, 0, 10, 10, kind="q", bcids=(1, 1, 1, 1), factor=300.0, loading=2)
    run("quads 48 clamped", m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, (0.3, 1e7, 0.1), thetas)
elif which == "strip":
    m = meshes.structured(128, 8, 0, 0, 16, 1, kind="t", ul_lr=True, bcids=(-1, -1, -1, 1), factor=1.0, loading=2)
    run("strip 128x8 t=0.01", m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, (0.3, 1e7, 0.01), thetas)
elif which == "sliver":
    from strength_experiment_common import problem
    xyz, tri, dmask, A0, F0 = problem(3000, 2, float(sys.argv[2]))
    n = len(xyz); loads = np.zeros((n, 6)); loads[:, 2] = 1.0
    run("delaunay 3000 qmin %s" % sys.argv[2], xyz, tri, np.zeros((0, 4), np.int32), dmask, loads, (0.3, 7.0e4, 0.03), thetas)
elif which == "jit":
    xyz, tri = delaunay_shell(3000, 3, jittered=True)
    n = len(xyz); dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
    loads = np.zeros((n, 6)); loads[:, 2] = 1.0
    run("jittered delaunay 3000", xyz, tri, np.zeros((0, 4), np.int32), dmask, loads, (0.3, 7.0e4, 0.03), thetas)
