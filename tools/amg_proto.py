"""Prototype of a smoothed-aggregation multigrid preconditioner for the shell systems (scipy, CPU).

Throwaway exploration tool: decides the design of the device preconditioner.
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from tests.helpers import meshes, oracle  # noqa: E402


def rbm(xyz, dmask):
    """6 rigid body modes per node, (n,6,6): columns = modes; rows = dofs (u,v,w,tx,ty,tz)."""
    n = len(xyz)
    c = xyz - xyz.mean(axis=0)
    B = np.zeros((n, 6, 6))
    for i in range(3):
        B[:, i, i] = 1.0
        B[:, 3 + i, 3 + i] = 1.0
    # u = w x r : rotation about x: (0,-z,y); about y: (z,0,-x); about z: (-y,x,0)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    B[:, 1, 3], B[:, 2, 3] = -z, y
    B[:, 0, 4], B[:, 2, 4] = z, -x
    B[:, 0, 5], B[:, 1, 5] = -y, x
    for v in range(6):
        fixed = (dmask >> v) & 1
        B[fixed == 1, v, :] = 0.0
    return B


def aggregate(rowptr, colidx, strong=None):
    """Greedy distance-1 aggregation (Vanek): returns agg id per node, n_agg."""
    n = len(rowptr) - 1
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    # pass 1: nodes whose whole neighbourhood is free become roots
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = colidx[rowptr[i]:rowptr[i + 1]]
        if strong is not None:
            nb = nb[strong[rowptr[i]:rowptr[i + 1]]]
        if len(nb) <= 1:
            continue
        if np.all(agg[nb] < 0):
            agg[nb] = na
            na += 1
    # pass 2: attach leftovers to a neighbouring aggregate
    agg2 = agg.copy()
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = colidx[rowptr[i]:rowptr[i + 1]]
        if strong is not None:
            nb = nb[strong[rowptr[i]:rowptr[i + 1]]]
        cand = agg[nb]
        cand = cand[cand >= 0]
        if len(cand):
            agg2[i] = cand[0]
    agg = agg2
    # pass 3: remaining isolated nodes form their own aggregates
    for i in range(n):
        if agg[i] < 0:
            nb = colidx[rowptr[i]:rowptr[i + 1]]
            if strong is not None:
                nb = nb[strong[rowptr[i]:rowptr[i + 1]]]
            free = nb[agg[nb] < 0]
            agg[free] = na
            agg[i] = na
            na += 1
    return agg, na


def tentative(agg, na, B):
    """P0 (BSR n x na blocks 6x6) and coarse B (na,6,6) by per-aggregate QR."""
    n = len(agg)
    order = np.argsort(agg, kind="stable")
    counts = np.bincount(agg, minlength=na)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    Q = np.zeros((n, 6, 6))
    Bc = np.zeros((na, 6, 6))
    for a in range(na):
        idx = order[ptr[a]:ptr[a + 1]]
        M = B[idx].reshape(-1, 6)
        q, r = np.linalg.qr(M)  # M (6k x 6)
        if q.shape[1] < 6:  # fewer than 6 rows (cannot happen with k>=1: 6 rows)
            raise RuntimeError
        # rank deficiency (e.g. all dofs fixed or single node) -> zero columns handled by r small
        Q[idx] = q.reshape(-1, 6, 6)
        Bc[a] = r
    P0 = sp.bsr_matrix((Q, agg.astype(np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * na))
    return P0, Bc


def block_diag_inv(A):
    """inverse of the 6x6 diagonal blocks of BSR A, (n,6,6)."""
    A = A.tobsr((6, 6))
    A.sort_indices()
    n = A.shape[0] // 6
    D = np.zeros((n, 6, 6))
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    sel = A.indices == rows
    D[rows[sel]] = A.data[sel]
    # guard singular blocks
    Dinv = np.zeros_like(D)
    for i in range(n):
        try:
            Dinv[i] = np.linalg.inv(D[i])
        except np.linalg.LinAlgError:
            Dinv[i] = np.linalg.pinv(D[i])
    return Dinv


def bd_matrix(Dinv):
    n = len(Dinv)
    return sp.bsr_matrix((Dinv, np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32)), shape=(6 * n, 6 * n))


def lam_max(A, Dinv_m, its=15):
    """largest eigenvalue of Dinv A by power iteration / Lanczos via eigsh on the symmetrised operator"""
    n = A.shape[0]
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    lam = 1.0
    for _ in range(its):
        y = Dinv_m @ (A @ x)
        lam = np.linalg.norm(y) / np.linalg.norm(x)
        x = y / np.linalg.norm(y)
    return lam


class Level:
    pass


def setup(A, xyz, dmask, rowptr, colidx, max_levels=10, coarse_nodes=200, omega_scale=4.0 / 3.0, theta=0.0):
    levels = []
    B = rbm(xyz, dmask)
    A = A.tobsr((6, 6))
    while True:
        L = Level()
        L.A = A
        L.Dinv = block_diag_inv(A)
        L.Dm = bd_matrix(L.Dinv)
        L.lam = lam_max(A, L.Dm) * 1.05
        levels.append(L)
        n = A.shape[0] // 6
        if n <= coarse_nodes or len(levels) >= max_levels:
            L.dense = np.linalg.pinv(A.toarray(), hermitian=True)
            break
        Ab = A.tobsr((6, 6))
        Ab.sort_indices()
        rp, ci = Ab.indptr, Ab.indices
        strong = None
        if theta > 0:
            nrm = np.sqrt((Ab.data ** 2).sum(axis=(1, 2)))
            rows = np.repeat(np.arange(n), np.diff(rp))
            dn = np.zeros(n)
            dn[rows[ci == rows]] = nrm[ci == rows]
            strong = nrm >= theta * np.sqrt(dn[rows] * dn[ci])
        agg, na = aggregate(rp, ci, strong)
        P0, Bc = tentative(agg, na, B)
        omega = omega_scale / L.lam
        P = P0 - omega * (L.Dm @ (A @ P0))
        P = P.tobsr((6, 6))
        L.P = P
        L.R = P.T.tobsr((6, 6))
        A = (L.R @ (A @ P)).tobsr((6, 6))
        B = Bc
        print("  level %d: %d nodes -> %d aggregates, coarse nnzb/row %.1f, lam %.3f" % (len(levels) - 1, n, na, A.nnz / 36 / na, L.lam))
    return levels


def cheby(L, b, x, degree, ratio=30.0):
    """Chebyshev smoothing for D^-1 A with eigenvalues in [lam/ratio, lam]; x may be None (zero initial guess)."""
    lmax, lmin = L.lam, L.lam / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma
    if x is None:
        r = b.copy()
        x = np.zeros_like(b)
    else:
        r = b - L.A @ x
    d = (L.Dm @ r) / theta
    for k in range(degree):
        x = x + d
        if k == degree - 1:
            break
        r = r - L.A @ d
        rho_new = 1.0 / (2.0 * sigma - rho)
        d = rho_new * rho * d + (2.0 * rho_new / delta) * (L.Dm @ r)
        rho = rho_new
    return x


def vcycle(levels, li, b, degree):
    L = levels[li]
    if li == len(levels) - 1:
        return L.dense @ b
    x = cheby(L, b, None, degree)
    r = b - L.A @ x
    xc = vcycle(levels, li + 1, L.R @ r, degree)
    x = x + L.P @ xc
    x = cheby(L, b, x, degree)
    return x


def pcg(A, b, M, rtol, max_it):
    x = np.zeros_like(b)
    r = b.copy()
    z = M(r)
    p = z.copy()
    rz = r @ z
    bb = np.sqrt(b @ b)
    hist = []
    for it in range(max_it):
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        rr = np.sqrt(r @ r) / bb
        hist.append(rr)
        if rr <= rtol:
            break
        z = M(r)
        rzn = r @ z
        p = z + (rzn / rz) * p
        rz = rzn
    return x, hist


def problem(kind, n):
    if kind == "panel":
        m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
        mat = oracle.material(0.3, 1e7, 0.5)
    elif kind == "roof":
        m = meshes.scordelis_lo(n)
        mat = oracle.material(*m.material)
    elif kind == "cyl":
        m = meshes.pinched_cylinder(n, n)
        mat = oracle.material(*m.material)
    elif kind == "flap":
        m = meshes.structured(n // 2, n, 0, 0, 0.1, 1.0, kind="t", ul_lr=True, dead_axis="y", bcids=(2, 20, 2, 2))
        m.loads[:, 0] = 1.0 / m.n_nodes
        mat = oracle.material(0.3, 1e6, 0.1)
    else:
        raise ValueError(kind)
    dm = m.dirichlet_mask()
    rp, ci, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, dm, m.loads)
    A = sp.bsr_matrix((vals, ci, rp), shape=(6 * m.n_nodes, 6 * m.n_nodes))
    return m, dm, rp, ci, A, F


if __name__ == "__main__":
    kind = sys.argv[1]
    n = int(sys.argv[2])
    degree = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    theta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    t0 = time.time()
    m, dm, rp, ci, A, F = problem(kind, n)
    print("%s n=%d: %d nodes, %d tri; assembled in %.1fs" % (kind, n, m.n_nodes, len(m.tri), time.time() - t0))
    t0 = time.time()
    levels = setup(A, m.xyz, dm, rp, ci, theta=theta)
    print("setup %.1fs, %d levels" % (time.time() - t0, len(levels)))
    nnz = [L.A.nnz for L in levels]
    print("operator complexity %.3f" % (sum(nnz) / nnz[0]))
    t0 = time.time()
    x, hist = pcg(A, F, lambda r: vcycle(levels, 0, r, degree), 1e-12, 2000)
    print("AMG-PCG(deg %d): %d iterations to %.1e in %.1fs" % (degree, len(hist), hist[-1], time.time() - t0))
    for k in (1e-6, 1e-8, 1e-10, 1e-12):
        its = next((i + 1 for i, h in enumerate(hist) if h <= k), None)
        print("   rtol %.0e: %s its" % (k, its))
    res = np.linalg.norm(F - A @ x) / np.linalg.norm(F)
    print("true residual %.2e" % res)
