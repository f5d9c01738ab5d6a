"""CPU restatement (numpy/scipy) of the smoothed-aggregation multigrid preconditioner of libfemshell
(fem-shell_amd/csrc/amg.hpp): TEST INFRASTRUCTURE ONLY -- imported by tests/ and tools/, never by the product.

The reference delegates the solve to PETSc's KSP with user-chosen -ksp_type/-pc_type
(/root/reference/src/fem-shell/fem-shell.cpp:130-138, doc/implementation.tex:68-72); there is no reference
implementation of a multigrid method to restate.  What pins this path is therefore (a) the converged
displacements, which are solver independent and are compared with the direct solve of the oracle-assembled K,
and (b) this independent restatement of the published algorithm (Vanek, Mandel, Brezina, Computing 56, 1996;
K cycle: Notay & Vassilevski, NLAA 15, 2008), against which the hierarchy the library builds is compared
operator by operator.

Conventions: matrices are scipy BSR with 6x6 blocks; B[n, dof, mode] is the near-null space.
"""
import os

import numpy as np
import scipy.sparse as sp


def node_normals(xyz, tri=None, quad=None):
    """Area-weighted unit normals of the nodes (csrc/amg_setup.cpp: node_normals)."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    N = np.zeros_like(xyz)
    if tri is not None and len(tri):
        t = np.asarray(tri).reshape(-1, 3)
        w = np.cross(xyz[t[:, 1]] - xyz[t[:, 0]], xyz[t[:, 2]] - xyz[t[:, 0]])
        for i in range(3):
            np.add.at(N, t[:, i], w)
    if quad is not None and len(quad):
        q = np.asarray(quad).reshape(-1, 4)
        w = np.cross(xyz[q[:, 1]] - xyz[q[:, 0]], xyz[q[:, 2]] - xyz[q[:, 0]]) + np.cross(xyz[q[:, 2]] - xyz[q[:, 0]], xyz[q[:, 3]] - xyz[q[:, 0]])
        for i in range(4):
            np.add.at(N, q[:, i], w)
    ln = np.linalg.norm(N, axis=1)
    ok = ln > 0
    N[ok] /= ln[ok, None]
    return N


def rigid_body_modes(xyz, dmask, normals=None):
    """normals: the rotational part of the rotation modes is projected onto the tangent plane of each node (the drilling
    stiffness penalises the rotation about the normal without coupling it to the in-plane displacements)."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    n = len(xyz)
    c = xyz - xyz.sum(axis=0) / n
    B = np.zeros((n, 6, 6))
    for i in range(6):
        B[:, i, i] = 1.0
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    B[:, 1, 3], B[:, 2, 3] = -z, y   # rotation about x: u = (0,-z,y)
    B[:, 0, 4], B[:, 2, 4] = z, -x   # about y: (z,0,-x)
    B[:, 0, 5], B[:, 1, 5] = -y, x   # about z: (-y,x,0)
    if normals is not None:
        B[:, 3:6, 3:6] -= normals[:, :, None] * normals[:, None, :]
    if dmask is not None:
        dmask = np.asarray(dmask)
        for v in range(6):
            B[((dmask >> v) & 1) == 1, v, :] = 0.0
    return B


def aggregation_order(rowptr, colidx):
    """Index order, or breadth-first order of the graph when the numbering is scattered (csrc/amg_setup.cpp)."""
    n = len(rowptr) - 1
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    if n < 64 or rowptr[n] == 0:
        return range(n)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    if np.abs(colidx - rows).sum() / rowptr[n] <= 8.0 * np.sqrt(n):
        return range(n)
    seen = np.zeros(n, dtype=bool)
    order = []
    for s0 in range(n):
        if seen[s0]:
            continue
        seen[s0] = True
        order.append(s0)
        head = len(order) - 1
        while head < len(order):
            i = order[head]
            head += 1
            for j in colidx[rowptr[i]:rowptr[i + 1]]:
                if not seen[j]:
                    seen[j] = True
                    order.append(int(j))
    return order


def aggregation_keep():
    """Neighbours per node the aggregation looks at (csrc/amg_setup.cpp aggregation_keep: FEMSHELL_AMG_AGG_KEEP, default 12)."""
    e = os.environ.get("FEMSHELL_AMG_AGG_KEEP")
    return int(e) if e not in (None, "") else 12


def graph_for_aggregation(rowptr, colidx, keep=None):
    """The graph the greedy passes see (csrc/amg_setup.cpp graph_for_aggregation): when a row has more than `keep` neighbours
    besides the node itself, every such node keeps the `keep` neighbours it shares most neighbours with (ties: the lower
    column), and an edge stays when either end keeps it.  Rows must be sorted.  Returns (rowptr, colidx)."""
    keep = aggregation_keep() if keep is None else keep
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    if keep <= 0 or n == 0 or deg.max() <= keep + 1:
        return rowptr, colidx
    kept = np.ones(len(colidx), dtype=bool)
    for i in np.nonzero(deg > keep + 1)[0]:
        b, e = rowptr[i], rowptr[i + 1]
        nb = colidx[b:e]
        kept[b:e] = nb == i
        cand = []
        for j in nb:
            if j == i:
                continue
            common = len(np.intersect1d(nb, colidx[rowptr[j]:rowptr[j + 1]], assume_unique=True))
            cand.append((-common, int(j)))
        cand.sort()
        for _, j in cand[:keep]:
            kept[b + np.searchsorted(nb, j)] = True
    rows = np.repeat(np.arange(n), deg)
    pattern = sp.csr_matrix((np.ones(len(colidx), dtype=np.int64), (rows, colidx)), shape=(n, n))
    mine = sp.csr_matrix((kept.astype(np.int64), (rows, colidx)), shape=(n, n))
    either = ((mine + mine.T).multiply(pattern)).tocsr()  # the other end's opinion counts where the reverse edge exists
    either.eliminate_zeros()
    either.sort_indices()
    return either.indptr.astype(np.int64), either.indices.astype(np.int64)


def aggregation_chunk():
    """csrc/amg_setup.cpp aggregation_chunk: rows per piece of the chunked aggregation (FEMSHELL_AMG_AGG_CHUNK; default 0 = off)."""
    e = os.environ.get("FEMSHELL_AMG_AGG_CHUNK")
    return int(e) if e else 0


def aggregate(rowptr, colidx, visit=None):
    """csrc/amg_setup.cpp aggregate_nodes: graphs of more than one and a half chunks are cut into pieces of consecutive rows (the
    boundaries of a row partition over ceil(n / chunk) ranks), every piece is aggregated on its own without the edges that leave it
    (aggregate_piece), the aggregates are numbered piece by piece.  visit: the visiting order, restricted to each piece."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    n = len(rowptr) - 1
    chunk = aggregation_chunk()
    if chunk <= 0 or n <= chunk + chunk // 2:
        return aggregate_piece(rowptr, colidx, visit)
    nc = (n + chunk - 1) // chunk
    bounds = partition_bounds_equal(n, nc)
    agg = np.empty(n, dtype=np.int64)
    total = 0
    vis = None if visit is None else np.asarray(visit, dtype=np.int64)
    for k in range(nc):
        b0, b1 = int(bounds[k]), int(bounds[k + 1])
        ci = colidx[rowptr[b0]:rowptr[b1]]
        rows = np.repeat(np.arange(b0, b1), np.diff(rowptr[b0:b1 + 1]))
        keep = (ci >= b0) & (ci < b1)
        lp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep] - b0, minlength=b1 - b0))])
        v = None if vis is None else vis[(vis >= b0) & (vis < b1)] - b0
        a, na = aggregate_piece(lp, ci[keep] - b0, v)
        agg[b0:b1] = a + total
        total += na
    return agg, total


def aggregate_piece(rowptr, colidx, visit=None):
    """Greedy distance-1 aggregation, three passes, on graph_for_aggregation; returns (agg, n_aggregates).  visit: the order in
    which the passes meet the nodes (default: aggregation_order); a leftover of pass 1 joins the neighbour that comes first in it."""
    rowptr, colidx = graph_for_aggregation(rowptr, colidx)
    n = len(rowptr) - 1
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    order = aggregation_order(rowptr, colidx) if visit is None else [int(i) for i in visit]
    rank = np.empty(n, dtype=np.int64)
    rank[np.fromiter(order, dtype=np.int64, count=n)] = np.arange(n)
    for i in order:
        if agg[i] >= 0:
            continue
        nb = colidx[rowptr[i]:rowptr[i + 1]]
        if len(nb) <= 1:
            continue
        if np.all(agg[nb] < 0):
            agg[nb] = na
            na += 1
    agg2 = agg.copy()
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = colidx[rowptr[i]:rowptr[i + 1]]
        nb = nb[agg[nb] >= 0]
        if len(nb):
            agg2[i] = agg[nb[np.argmin(rank[nb])]]
    agg = agg2
    for i in order:
        if agg[i] < 0:
            nb = colidx[rowptr[i]:rowptr[i + 1]]
            agg[nb[agg[nb] < 0]] = na
            agg[i] = na
            na += 1
    return agg, na


def tentative(agg, na, B):
    """Per aggregate B_agg = Q R by modified Gram-Schmidt (two passes); dependent columns -> zero column of Q."""
    n = len(agg)
    order = np.argsort(agg, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(agg, minlength=na))])
    Q = np.zeros((n, 6, 6))
    Bc = np.zeros((na, 6, 6))
    for a in range(na):
        idx = order[ptr[a]:ptr[a + 1]]
        M = B[idx].reshape(-1, 6).copy()
        R = np.zeros((6, 6))
        for j in range(6):
            n0 = np.linalg.norm(M[:, j])
            for _ in range(2):
                for i in range(j):
                    c = M[:, i] @ M[:, j]
                    M[:, j] -= c * M[:, i]
                    R[i, j] += c
            nj = np.linalg.norm(M[:, j])
            if n0 > 0.0 and nj > 1e-8 * n0:
                R[j, j] = nj
                M[:, j] /= nj
            else:
                R[j, j] = 0.0
                M[:, j] = 0.0
        Q[idx] = M.reshape(-1, 6, 6)
        Bc[a] = R
    return Q, Bc


def block_diag(A):
    A = A.tobsr((6, 6))
    A.sort_indices()
    n = A.shape[0] // 6
    D = np.zeros((n, 6, 6))
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    sel = A.indices == rows
    D[rows[sel]] = A.data[sel]
    return D


def block_diag_inverse(A):
    D = block_diag(A)
    Dinv = np.zeros_like(D)
    for i in range(len(D)):
        try:
            np.linalg.cholesky(D[i])
            Dinv[i] = np.linalg.inv(D[i])
        except np.linalg.LinAlgError:
            Dinv[i] = np.eye(6)
    return Dinv


def bd_matrix(Dinv):
    n = len(Dinv)
    return sp.bsr_matrix((Dinv, np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * n))


# ---- patch smoother: clusters of rigidly coupled nodes (csrc/amg_patch.hpp) ---------------------------------------------

PATCH_POWER_STEPS = 16


def patch_sigma2(Di, A, Dj, tau2):
    """csrc/amg_patch.hpp patch_sigma2, over arrays of blocks: trace(Di A Dj A^T) where it is below tau2, else the norm ratio of the
    last of 16 power steps on T = Di A Dj A^T from the vector of ones."""
    Y = np.einsum("eab,ebc,ecd->ead", Di, A, Dj)
    tr = np.einsum("eab,eab->e", Y, A)
    out = tr.copy()
    go = np.flatnonzero(tr >= tau2)
    if len(go):
        Yg, Ag = Y[go], A[go]
        v = np.ones((len(go), 6))
        lam = np.zeros(len(go))
        for _ in range(PATCH_POWER_STEPS):
            w = np.einsum("eij,ei->ej", Ag, v)
            u = np.einsum("eij,ej->ei", Yg, w)
            nv, nu = np.einsum("ei,ei->e", v, v), np.einsum("ei,ei->e", u, u)
            ok = (nu > 0.0) & (nv > 0.0)
            lam = np.where(ok, np.sqrt(np.where(ok, nu / np.where(nv > 0, nv, 1.0), 0.0)), 0.0)
            v = np.where(ok[:, None], u / np.sqrt(np.where(nu > 0, nu, 1.0))[:, None], v)
        out[go] = lam
    return out


def patch_edges(A, Dinv, tau):
    """Rigid edges (a < c, sigma2) of a level: csrc/amg_kernels.hip k_patch_sigma / amg_setup.cpp patch_edges_host."""
    A = A.tobsr((6, 6))
    A.sort_indices()
    n = A.shape[0] // 6
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    up = np.flatnonzero(A.indices > rows)
    s2 = patch_sigma2(Dinv[rows[up]], A.data[up], Dinv[A.indices[up]], tau * tau)
    keep = s2 > tau * tau
    return rows[up][keep], A.indices[up][keep], s2[keep]


def patch_clusters(n, ea, ec, s2, max_nodes=6):
    """csrc/amg_setup.cpp patch_clusters: edges strongest first (ties: lower a, then lower c), two clusters are united while the
    union stays within max_nodes; clusters numbered by their smallest node.  Returns (label per node or -1, ptr, nodes)."""
    order = np.lexsort((ec, ea, -s2))
    parent = np.arange(n)
    size = np.ones(n, dtype=np.int64)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for e in order:
        ra, rc = find(int(ea[e])), find(int(ec[e]))
        if ra == rc or size[ra] + size[rc] > max_nodes:
            continue
        if rc < ra:
            ra, rc = rc, ra
        parent[rc] = ra
        size[ra] += size[rc]
    roots = np.array([find(i) for i in range(n)])
    label = -np.ones(n, dtype=np.int64)
    ids = {}
    for i in range(n):
        r = roots[i]
        if size[r] < 2:
            continue
        label[i] = ids.setdefault(int(r), len(ids))
    nc = len(ids)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(label[label >= 0], minlength=nc))]).astype(np.int64)
    nodes = np.argsort(label[label >= 0], kind="stable")
    nodes = np.flatnonzero(label >= 0)[nodes]
    return label, ptr, nodes


def patch_block_inverse(A, Dinv, label, ptr, nodes):
    """The smoother's block inverse with one block per cluster: the exact inverse of the cluster's diagonal block of A (point
    blocks where that block is not positive definite, and for every node outside a cluster).  Sparse (6 n) x (6 n)."""
    A = A.tocsr()
    n = len(Dinv)
    rows, cols, vals = [], [], []
    single = np.flatnonzero(label < 0)
    fell_back = 0
    for c in range(len(ptr) - 1):
        mem = nodes[ptr[c]:ptr[c + 1]]
        idx = (6 * mem[:, None] + np.arange(6)[None, :]).ravel()
        B = A[idx][:, idx].toarray()
        B = 0.5 * (B + B.T)
        try:
            np.linalg.cholesky(B)
            Bi = np.linalg.inv(B)
        except np.linalg.LinAlgError:
            fell_back += 1
            single = np.concatenate([single, mem])
            continue
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        rows.append(rr.ravel())
        cols.append(cc.ravel())
        vals.append(Bi.ravel())
    for i in single:
        idx = 6 * i + np.arange(6)
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        rows.append(rr.ravel())
        cols.append(cc.ravel())
        vals.append(Dinv[i].ravel())
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * n, 6 * n))


def aggregate_glued(rowptr, colidx, label, visit=None):
    """csrc/amg_setup.cpp aggregate_nodes_glued: every cluster is one node of the quotient graph (quotient nodes numbered in the
    order the visiting order meets them), the greedy passes run there, a cluster's nodes share their quotient node's aggregate."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    n = len(rowptr) - 1
    if not np.any(label >= 0):
        return aggregate(rowptr, colidx, visit)
    qid = -np.ones(n, dtype=np.int64)
    q_of_cluster = {}
    nq = 0
    for i in (range(n) if visit is None else [int(v) for v in visit]):
        c = int(label[i])
        if c < 0:
            qid[i] = nq
            nq += 1
        else:
            if c not in q_of_cluster:
                q_of_cluster[c] = nq
                nq += 1
            qid[i] = q_of_cluster[c]
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    G = sp.csr_matrix((np.ones(len(colidx)), (qid[rows], qid[colidx])), shape=(nq, nq))
    G = (G + sp.identity(nq, format="csr")).tocsr()
    G.sum_duplicates()
    G.sort_indices()
    aq, na = aggregate(G.indptr, G.indices)
    return aq[qid], na


def aggregate_by_rank(rowptr, colidx, bounds):
    """Aggregation of a level whose rows are split over ranks (csrc/amg_dist.cpp): every rank aggregates the graph of its
    own rows without the edges that leave it -- aggregates never span ranks -- and the aggregates are numbered rank by
    rank.  bounds: the world + 1 row boundaries.  Returns (agg, n_aggregates, coarse bounds)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    n = len(rowptr) - 1
    agg = np.empty(n, dtype=np.int64)
    cb = [0]
    for r in range(len(bounds) - 1):
        b0, b1 = int(bounds[r]), int(bounds[r + 1])
        ci = colidx[rowptr[b0]:rowptr[b1]]
        rows = np.repeat(np.arange(b0, b1), np.diff(rowptr[b0:b1 + 1]))
        keep = (ci >= b0) & (ci < b1)
        lp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep] - b0, minlength=b1 - b0))])
        a, na = aggregate(lp, ci[keep] - b0) if b1 > b0 else (np.zeros(0, dtype=np.int64), 0)
        agg[b0:b1] = a + cb[-1]
        cb.append(cb[-1] + na)
    return agg, cb[-1], cb


def coarsen(A, B, lam, bounds=None, patch=None):
    """One coarsening step with the upper spectral bound lam of D^-1 A: returns (agg, P, Ac, Bc), and the coarse row
    boundaries as a fifth item when the level is split over ranks (bounds: its row boundaries).  Only the aggregation knows
    about the ranks: tentative prolongator, smoothing with the whole A and the Galerkin product are the single-rank ones (the
    library exchanges the rows of Q, P and A P of the nodes along the cuts to that end)."""
    A = A.tobsr((6, 6))
    A.sort_indices()
    n = A.shape[0] // 6
    # (patch = (label, Dm): the level's clusters of rigidly coupled nodes -- glued into one node each before the greedy passes -- and
    #  its block inverse with one block per cluster, which smooths the prolongator too: csrc/amg_patch.hpp)
    if bounds is not None:
        agg, na, cbounds = aggregate_by_rank(A.indptr, A.indices, bounds)
    elif patch is not None:
        agg, na = aggregate_glued(A.indptr, A.indices, patch[0])
    else:
        agg, na = aggregate(A.indptr, A.indices)
    Q, Bc = tentative(agg, na, B)
    P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
    Dm = patch[1] if patch is not None else bd_matrix(block_diag_inverse(A))
    P = (P0 - ((4.0 / 3.0) / lam) * (Dm @ (A @ P0))).tobsr((6, 6))
    Ac = (P.T @ (A @ P)).tobsr((6, 6))
    # coarse dofs without fine support: unit diagonal
    d = Ac.diagonal()
    if np.any(d == 0.0):
        fix = sp.diags((d == 0.0).astype(np.float64))
        Ac = (Ac + fix).tobsr((6, 6))
    if bounds is not None:
        return agg, P, Ac, Bc, cbounds
    return agg, P, Ac, Bc


def lambda_max(A, Dm, iterations=30):
    """Power iteration for D^-1 A (the library does the same on the device with its own start vector)."""
    x = np.random.default_rng(0).uniform(-1.0, 1.0, A.shape[0])
    prev, lam = 0.0, 1.0
    for it in range(iterations):
        x = Dm @ (A @ x)
        nrm = np.linalg.norm(x)
        if it > 0:
            lam = nrm / prev
        prev = nrm
    return lam


class Level:
    pass


def partition_bounds_equal(n, world):
    """Row boundaries of `world` ranks in whole slices of 32 nodes (csrc/plan.cpp partition_rows: the split of a regular grid)."""
    slices = (n + 31) // 32
    return [min(n, 32 * (slices * k // world)) for k in range(world + 1)]


def setup(A, xyz, dmask, lams=None, coarsest_nodes=200, max_levels=12, eig_ratio=30.0, degree=3, coarse_degree=4,
          tri=None, quad=None, bounds=None, dist_min=60000, patch_tau=None, patch_max=6, patch_labels=None):
    """lams: upper bounds of the spectrum per level as the library reports them (femshell_amg_level); computed
    here (1.1 x power iteration) when None.  tri / quad: connectivity for the node normals of rigid_body_modes.
    bounds: row boundaries of a row-partitioned context (world + 1 entries): level 0, and every coarser level of more than
    dist_min nodes, is aggregated rank by rank; level 0 is then coarsened whatever its size."""
    levels = []
    normals = node_normals(xyz, tri, quad) if (tri is not None or quad is not None) else None
    B = rigid_body_modes(xyz, dmask, normals)
    A = A.tobsr((6, 6))
    while True:
        L = Level()
        L.A = A
        L.n = A.shape[0] // 6
        L.Dm = bd_matrix(block_diag_inverse(A))
        levels.append(L)
        li = len(levels) - 1
        L.patch = None
        if li == 0 and (patch_tau or patch_labels is not None) and bounds is None:
            # the patch smoother of level 0 (csrc/amg_patch.hpp): clusters from the rigid edges -- or the labels the library reports --
            # and the block inverse with one block per cluster in the smoother, the spectral bound and the smoothing of P
            Dinv = block_diag_inverse(A)
            if patch_labels is not None:
                label = np.asarray(patch_labels, dtype=np.int64)
                nc = int(label.max()) + 1 if len(label) else 0
                ptr = np.concatenate([[0], np.cumsum(np.bincount(label[label >= 0], minlength=nc))]).astype(np.int64)
                nodes = np.flatnonzero(label >= 0)[np.argsort(label[label >= 0], kind="stable")]
            else:
                ea, ec, s2 = patch_edges(A, Dinv, patch_tau)
                label, ptr, nodes = patch_clusters(L.n, ea, ec, s2, patch_max)
            if len(ptr) > 1:
                L.Dm = patch_block_inverse(A, Dinv, label, ptr, nodes)
                L.patch = (label, L.Dm)
                L.patch_label = label
        split = bounds is not None and (li == 0 or L.n > dist_min)
        L.bounds = list(bounds) if split else None
        # (K itself is the coarsest level only up to 200 nodes: csrc/amg_device.hpp kDirectNodes)
        limit = min(coarsest_nodes, 200) if li == 0 else coarsest_nodes
        if (L.n <= limit and not (split and li == 0)) or len(levels) >= max_levels:
            L.dense_inv = np.linalg.inv(A.toarray())
            break
        L.lam = lams[li] if lams is not None else 1.1 * lambda_max(A, L.Dm)
        if split:
            L.agg, L.P, Ac, B, bounds = coarsen(A, B, L.lam, bounds)
        else:
            L.agg, L.P, Ac, B = coarsen(A, B, L.lam, patch=L.patch)
            bounds = None
        L.R = L.P.T.tobsr((6, 6))
        deg = degree if li == 0 else coarse_degree
        lmax, lmin = L.lam, L.lam / eig_ratio
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        L.inv_theta = 1.0 / theta
        L.cheb = []
        rho = 1.0 / sigma
        for _ in range(1, deg):
            rho_new = 1.0 / (2.0 * sigma - rho)
            L.cheb.append((rho_new * rho, 2.0 * rho_new / delta))
            rho = rho_new
        A = Ac
    return levels


def smooth(L, b, x):
    """Chebyshev smoothing in D^-1 A; x None = zero initial guess."""
    if x is None:
        r = b
        x = np.zeros_like(b)
    else:
        r = b - L.A @ x
    d = L.inv_theta * (L.Dm @ r)
    x = x + d
    for a, c in L.cheb:
        r = r - L.A @ d
        d = a * d + c * (L.Dm @ r)
        x = x + d
    return x


def cycle(levels, li, b, kcycle):
    L = levels[li]
    if li == len(levels) - 1:
        return L.dense_inv @ b
    x = smooth(L, b, None)
    bc = L.R @ (b - L.A @ x)
    if kcycle and li + 2 < len(levels):
        xc = kcycle_solve(levels, li + 1, bc)
    else:
        xc = cycle(levels, li + 1, bc, kcycle)
    x = x + L.P @ xc
    return smooth(L, b, x)


def kcycle_solve(levels, li, rc):
    """Two steps of flexible CG on A_li x = rc preconditioned by the cycle of level li."""
    A = levels[li].A
    c1 = cycle(levels, li, rc, True)
    v1 = A @ c1
    rho1, a1 = c1 @ v1, c1 @ rc
    t = a1 / rho1 if rho1 > 0.0 else 0.0
    r2 = rc - t * v1
    c2 = cycle(levels, li, r2, True)
    v2 = A @ c2
    g, b2, a2 = c2 @ v1, c2 @ v2, c2 @ r2
    w1, w2 = t, 0.0
    if rho1 > 0.0:
        rho2 = b2 - g * g / rho1
        if rho2 > 0.0:
            w1 = a1 / rho1 - g * a2 / (rho1 * rho2)
            w2 = a2 / rho2
    return w1 * c1 + w2 * c2


def flexible_pcg(A, b, M, rtol=1e-10, max_it=1000, pass_of=None):
    """beta = z.(r - r_old) / r_old.z_old = -alpha z.q / rz_old; stops at ||r|| <= rtol ||b||.
    pass_of = (||x||^2 of the iterate this solve corrects, tolerance of the whole solve): the stopping rule of a refinement pass
    (csrc/kernels.hip kRefineTarget) -- the drop asked of the residual is 0.2 tol ||x|| / ||e_k||, within [1e-6, 1e-2]."""
    x = np.zeros_like(b)
    r = b.copy()
    bb = b @ b
    hist = []
    if bb == 0.0:
        return x, hist
    z = M(r)
    rz = r @ z
    p = z.copy()
    for _ in range(max_it):
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        rr = r @ r
        hist.append(np.sqrt(rr / bb))
        if pass_of is not None and x @ x > 0.0:
            rtol = min(max(0.2 * pass_of[1] * np.sqrt(pass_of[0] / (x @ x)), 1.0e-6), 1.0e-2)
        if rr <= rtol * rtol * bb:
            break
        z = M(r)
        rzn, zq = r @ z, z @ q
        beta = -alpha * zq / rz
        rz = rzn
        p = z + beta * p
    return x, hist


def residual_extended(A, b, x):
    """b - A x with products and row sums in numpy longdouble (the library uses double-double on the device)."""
    A = A.tocsr()
    A.sort_indices()
    ld = np.longdouble
    return (b.astype(ld) - np.add.reduceat(A.data.astype(ld) * x.astype(ld)[A.indices], A.indptr[:-1])).astype(np.float64)


def solve(A, b, levels, kcycle=True, rtol=1e-10, max_it=1000, refine_passes=0, adaptive=True, x0=None):
    """Flexible PCG around the cycle, then iterative refinement: residual of the iterate in extended precision, correction
    equation solved by the same method until its residual has dropped by 1e-4 -- adaptive: by 0.2 rtol ||x|| / ||e_k||, the drop
    that puts the estimate below at a fifth of the tolerance -- and x += e.  With a pass to follow the first
    phase stops a factor 100 above the tolerance.  ||e|| / ||x|| times the drop estimates the error the pass leaves; passes
    after the first -- at most `refine_passes`, and one more -- run while the estimate exceeds rtol
    x0: an initial guess (femshell_set_initial_guess) -- the first phase solves K e = b - K x0, the right-hand side in extended
    precision, down to the threshold it runs to from zero, and x0 + e takes the place of its iterate
    (csrc/amg_solve.cpp cg_amg)."""
    M = lambda r: cycle(levels, 0, r, kcycle)  # noqa: E731
    loosened = refine_passes >= 1 and rtol > 0
    rtol_first = min(100.0 * rtol, 1e-2) if loosened else rtol
    nb = np.linalg.norm(b)
    if x0 is None:
        x, hist = flexible_pcg(A, b, M, rtol_first, max_it)
    else:
        r = residual_extended(A, b, x0)
        nr = np.linalg.norm(r)
        if nr > rtol_first * nb:
            e, h = flexible_pcg(A, r, M, rtol_first * nb / nr, max_it)
            x, hist = x0 + e, [v * nr / nb for v in h]
        else:
            x, hist = x0.copy(), []
    est = None
    for k in range(refine_passes + (1 if loosened else 0)):
        r = residual_extended(A, b, x)
        nr = np.linalg.norm(r)
        if nr == 0.0 or len(hist) >= max_it or (k >= 1 and est <= rtol):
            break
        # the correction needs about four digits, not the full tolerance again: as many as put the estimate of what it leaves,
        # ||e|| / ||x|| x its drop, at a fifth of the tolerance (csrc/kernels.hip: kRefineDrop, kRefineTarget)
        e, h = flexible_pcg(A, r, M, 1.0e-4, max_it - len(hist), pass_of=(x @ x, rtol) if adaptive else None)
        hist = hist + [v * nr / nb for v in h]
        est = np.linalg.norm(e) / np.linalg.norm(x) * (h[-1] if h else 1.0)
        x = x + e
    return x, hist
