"""Rank-local aggregation for the row-partitioned multigrid, tried in the numpy restatement before it is built on the device:
what do aggregates that never span ranks, and a prolongator smoother without cross-rank couplings, cost in iterations?
python tools/lab/dist_amg_experiment.py panel|roof|cyl NX [dist_min]

variants of the prolongator smoother on a level whose rows are split over ranks (aggregates are rank-local in all of them):
  full      P = (I - w D^-1 A) P0 with the whole A (P then has entries in other ranks' aggregates: transfers need communication)
  drop      the blocks of A that couple different ranks are left out of the smoothing product
  boundary  rows that have a cross-rank coupling keep the tentative prolongator, all others are smoothed with the whole A
  lump      dropped blocks are added to the diagonal block of their row (filtered operator), D of the filtered operator
"""
import sys, os, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle, meshes
import amg_oracle as ao


def coarsen_part(A, B, lam, part, variant):
    A = A.tobsr((6, 6)); A.sort_indices()
    n = A.shape[0] // 6
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    same = part[rows] == part[A.indices]
    # rank-local aggregation: the graph without cross-rank edges, ranks one after the other (aggregates numbered rank by rank)
    cnt = np.bincount(rows[same], minlength=n)
    rp = np.concatenate([[0], np.cumsum(cnt)]); ci = A.indices[same]
    agg, na = ao.aggregate(rp, ci, visit=range(n))
    # renumber the aggregates rank by rank (stable): the coarse row partition is contiguous again
    arank = np.zeros(na, dtype=np.int64); arank[agg] = part
    order = np.argsort(arank, kind="stable"); ren = np.empty(na, dtype=np.int64); ren[order] = np.arange(na)
    agg = ren[agg]; cpart = arank[order]
    Q, Bc = ao.tentative(agg, na, B)
    P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
    w = (4.0 / 3.0) / lam
    if variant == "full":
        Dm = ao.bd_matrix(ao.block_diag_inverse(A))
        P = P0 - w * (Dm @ (A @ P0))
    else:
        data = A.data.copy(); data[~same] = 0.0
        Aloc = sp.bsr_matrix((data, A.indices, A.indptr), shape=A.shape)
        if variant == "drop":
            Dm = ao.bd_matrix(ao.block_diag_inverse(A))
            P = P0 - w * (Dm @ (Aloc @ P0))
        elif variant == "boundary":
            Dm = ao.bd_matrix(ao.block_diag_inverse(A))
            has_cross = np.bincount(rows[~same], minlength=n) > 0
            S = sp.diags(np.repeat((~has_cross).astype(np.float64), 6))
            P = P0 - w * (S @ (Dm @ (A @ P0)))
        elif variant == "lump":
            lump = np.zeros((n, 6, 6)); np.add.at(lump, rows[~same], A.data[~same])
            isd = A.indices == rows
            data[isd] += lump[rows[isd]]
            AF = sp.bsr_matrix((data, A.indices, A.indptr), shape=A.shape)
            Dm = ao.bd_matrix(ao.block_diag_inverse(AF))
            P = P0 - w * (Dm @ (AF @ P0))
        elif variant == "trunc":
            # smooth with the whole A, then cut the blocks of P that sit in other ranks' aggregates and put what they do to the
            # near-null space into the row's own aggregate: P_i,own += (sum_J P_iJ Bc_J) Bc_own^-1  (P Bc = B stays exact)
            Dm = ao.bd_matrix(ao.block_diag_inverse(A))
            Pf = (P0 - w * (Dm @ (A @ P0))).tobsr((6, 6)); Pf.sort_indices()
            prow = np.repeat(np.arange(n), np.diff(Pf.indptr))
            arank = np.zeros(na, dtype=np.int64); arank[agg] = part
            cross = arank[Pf.indices] != part[prow]
            pd = Pf.data.copy()
            delta = np.zeros((n, 6, 6)); np.add.at(delta, prow[cross], np.einsum("kab,kbc->kac", pd[cross], Bc[Pf.indices[cross]]))
            pd[cross] = 0.0
            own = Pf.indices == agg[prow]
            Rinv = np.array([np.linalg.pinv(Bc[I], rcond=1e-12) for I in range(na)])
            pd[own] += np.einsum("kab,kbc->kac", delta[prow[own]], Rinv[agg[prow[own]]])
            P = sp.bsr_matrix((pd, Pf.indices, Pf.indptr), shape=Pf.shape)
            P.eliminate_zeros()
        elif variant in ("gmod", "gmodD"):
            # filtered operator that keeps the near-null space: the dropped blocks of row i act on B_j; put G_i = (sum_j A_ij B_j) B_i^+
            # on the diagonal, so that A_F B = A B (= 0 on free rows) for all six modes, not for the translations only
            g = np.zeros((n, 6, 6)); np.add.at(g, rows[~same], np.einsum("kab,kbc->kac", A.data[~same], B[A.indices[~same]]))
            G = np.zeros((n, 6, 6))
            for i in np.nonzero(np.bincount(rows[~same], minlength=n) > 0)[0]:
                G[i] = g[i] @ np.linalg.pinv(B[i], rcond=1e-10)
            isd = A.indices == rows
            data[isd] += G[rows[isd]]
            AF = sp.bsr_matrix((data, A.indices, A.indptr), shape=A.shape)
            Dm = ao.bd_matrix(ao.block_diag_inverse(A)) if variant == "gmodD" else ao.bd_matrix(np.linalg.inv(ao.block_diag(AF)))
            P = P0 - w * (Dm @ (AF @ P0))
        else:
            raise ValueError(variant)
    P = P.tobsr((6, 6))
    Ac = (P.T @ (A @ P)).tobsr((6, 6))
    d = Ac.diagonal()
    if np.any(d == 0.0):
        Ac = (Ac + sp.diags((d == 0.0).astype(np.float64))).tobsr((6, 6))
    return agg, P, Ac, Bc, cpart


def setup_part(A, xyz, dmask, tri, part, variant, dist_min, coarsest_nodes):
    levels = []
    B = ao.rigid_body_modes(xyz, dmask, ao.node_normals(xyz, tri))
    A = A.tobsr((6, 6))
    while True:
        L = ao.Level(); L.A = A; L.n = A.shape[0] // 6; L.Dm = ao.bd_matrix(ao.block_diag_inverse(A)); levels.append(L)
        li = len(levels) - 1
        if L.n <= coarsest_nodes:
            L.dense_inv = np.linalg.inv(A.toarray()); break
        L.lam = 1.1 * ao.lambda_max(A, L.Dm)
        if part is not None and (li == 0 or L.n > dist_min):
            vs = variant.split("/")
            L.agg, L.P, Ac, B, part = coarsen_part(A, B, L.lam, part, vs[min(li, len(vs) - 1)])
            L.dist = True
        else:
            L.agg, L.P, Ac, B = ao.coarsen(A, B, L.lam); part = None
            L.dist = False
        L.R = L.P.T.tobsr((6, 6))
        deg = 3 if li == 0 else 4
        lmax, lmin = L.lam, L.lam / 30.0
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin); sigma = theta / delta
        L.inv_theta = 1.0 / theta; L.cheb = []; rho = 1.0 / sigma
        for _ in range(1, deg):
            rho_new = 1.0 / (2.0 * sigma - rho); L.cheb.append((rho_new * rho, 2.0 * rho_new / delta)); rho = rho_new
        A = Ac
    return levels


VARIANTS = os.environ.get("VARIANTS", "full,boundary,gmod,gmodD").split(",")

if __name__ == "__main__":
    which, NX = sys.argv[1], int(sys.argv[2])
    dist_min = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
    if which == "panel":
        m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2); mat = (0.3, 1e7, 0.5)
    elif which == "roof":
        m = meshes.scordelis_lo(NX); mat = m.material
    else:
        m = meshes.pinched_cylinder(NX, NX); mat = m.material
    r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), dirichlet=m.dirichlet_mask(), loads=m.loads)
    A = oracle.to_scipy(r, c, v).tobsr((6, 6))
    n = len(m.xyz)
    print(which, NX, "nodes", n, "dist_min", dist_min, flush=True)
    t0 = time.time()
    levels = setup_part(A, m.xyz, m.dirichlet_mask(), m.tri, None, None, dist_min, 60)
    u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=1)
    print("single rank: levels %s, %d iterations (%.0f s)" % ([L.n for L in levels], len(hist), time.time() - t0), flush=True)
    for world in (2, 4, 8):
        # the library's partition: whole slices of 32 nodes, equal shares
        slices = (n + 31) // 32
        bounds = [min(n, 32 * (slices * k // world)) for k in range(world + 1)]
        part = np.zeros(n, dtype=np.int64)
        for k in range(world):
            part[bounds[k]:bounds[k + 1]] = k
        for variant in VARIANTS:
            t0 = time.time()
            levels = setup_part(A, m.xyz, m.dirichlet_mask(), m.tri, part, variant, dist_min, 60)
            u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=1)
            print("world %d %-8s: levels %s dist %s, %d iterations (%.0f s)" % (world, variant, [L.n for L in levels],
                  [int(getattr(L, "dist", False)) for L in levels[:-1]], len(hist), time.time() - t0), flush=True)
