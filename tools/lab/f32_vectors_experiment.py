"""What would single-precision VECTORS inside the multigrid cycle cost in iterations?  The library already reads
single-precision copies of the level operators, D^-1, P and R where the cycle only smooths or transfers (levels of at
least 4096 nodes); this experiment rounds the cycle's own vectors (corrections, residuals, products, transposed products)
to float32 after every operation on those levels as well -- storage in single precision, arithmetic in double, the way a
kernel would do it -- in the numpy restatement.   python tools/lab/f32_vectors_experiment.py panel|roof|cyl|flap NX [min_nodes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import meshes, oracle  # noqa: E402
import amg_oracle as ao  # noqa: E402

MODE = {"mats": False, "vecs": False, "min_nodes": 4096}


def f32(v):
    return v.astype(np.float32).astype(np.float64)


def low(L):
    return L.n >= MODE["min_nodes"]


def mat32(M):
    M = M.copy()
    M.data = f32(M.data)
    return M


def prepare(levels):
    for li, L in enumerate(levels[:-1]):
        L.A32, L.Dm32 = mat32(L.A), mat32(L.Dm)
        big_coarse = levels[li + 1].n >= MODE["min_nodes"]
        L.P32, L.R32 = (mat32(L.P), mat32(L.R)) if big_coarse else (L.P, L.R)


def smooth(L, b, x, state):
    """Chebyshev smoothing as the library runs it: residuals by increments; returns x and the last (r, d)."""
    mats = MODE["mats"] and low(L)
    vecs = MODE["vecs"] and low(L)
    A, Dm = (L.A32, L.Dm32) if mats else (L.A, L.Dm)
    rd = f32 if vecs else (lambda v: v)
    if x is None:
        r = rd(b)
        x = np.zeros_like(b)
    else:
        r = rd(state["r"] - A @ state["d"])  # increment: the residual the pre-smoothing left, minus A (last step)
        r = r if state.get("r_is_current") else r
    d = rd(L.inv_theta * (Dm @ r))
    x = rd(x + d)
    for a, c in L.cheb:
        r = rd(r - A @ d)
        d = rd(a * d + c * (Dm @ r))
        x = rd(x + d)
    state["r"], state["d"] = r, d
    return x


def cycle(levels, li, b, kcycle):
    L = levels[li]
    if li == len(levels) - 1:
        return L.dense_inv @ b
    mats = MODE["mats"] and low(L)
    vecs = MODE["vecs"] and low(L)
    rd = f32 if vecs else (lambda v: v)
    A = L.A32 if mats else L.A
    P, R = (L.P32, L.R32) if mats else (L.P, L.R)
    st = {}
    x = smooth(L, b, None, st)
    r = rd(st["r"] - A @ st["d"])  # residual after pre-smoothing, by increment
    bc = R @ r
    nxt = levels[li + 1]
    if MODE["vecs"] and nxt.n >= MODE["min_nodes"] and li + 2 < len(levels):
        bc = f32(bc)
    if kcycle and li + 2 < len(levels):
        xc = kcycle_solve(levels, li + 1, bc)
    else:
        xc = cycle(levels, li + 1, bc, kcycle)
    e = rd(P @ xc)
    x = rd(x + e)
    # post-smoothing starts from the residual of x: r - A e (increment again)
    st2 = {"r": r, "d": e}
    return smooth(L, b, x, st2)


def kcycle_solve(levels, li, rc):
    L = levels[li]
    vecs = MODE["vecs"] and low(L) and li + 1 < len(levels)
    rd = f32 if vecs else (lambda v: v)
    A = L.A  # the K cycle's own products stay on the double-precision operator
    c1 = cycle(levels, li, rc, True)
    v1 = rd(A @ c1)
    rho1, a1 = c1 @ v1, c1 @ rc
    t = a1 / rho1 if rho1 > 0.0 else 0.0
    r2 = rd(rc - t * v1)
    c2 = cycle(levels, li, r2, True)
    v2 = rd(A @ c2)
    g, b2, a2 = c2 @ v1, c2 @ v2, c2 @ r2
    w1, w2 = t, 0.0
    if rho1 > 0.0:
        rho2 = b2 - g * g / rho1
        if rho2 > 0.0:
            w1 = a1 / rho1 - g * a2 / (rho1 * rho2)
            w2 = a2 / rho2
    return rd(w1 * c1 + w2 * c2)


ao.cycle = cycle
which, NX = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3:
    MODE["min_nodes"] = int(sys.argv[3])
if which == "panel":
    m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
    mat = (0.3, 1e7, 0.5)
elif which == "roof":
    m = meshes.scordelis_lo(NX)
    mat = m.material
elif which == "flap":
    m = meshes.structured(NX // 2, NX, 0, 0, 0.1, 1.0, kind="t", ul_lr=True, bcids=(2, 20, 2, 2), factor=1.0, loading=0)
    m.xyz = m.xyz[:, [0, 2, 1]].copy()
    mat = (0.3, 1e6, 0.1)
    m.loads[:] = 0.0
    m.loads[np.abs(m.xyz[:, 0]) < 1e-12, 0] = 1.0
else:
    m = meshes.pinched_cylinder(NX, NX)
    mat = m.material
r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), dirichlet=m.dirichlet_mask(), loads=m.loads)
A = oracle.to_scipy(r, c, v).tobsr((6, 6))
levels = ao.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=200, tri=m.tri)
prepare(levels)
print(which, NX, "levels", [L.n for L in levels], "single precision on levels of >=", MODE["min_nodes"], "nodes", flush=True)
for mats, vecs in ((False, False), (True, False), (True, True)):
    MODE["mats"], MODE["vecs"] = mats, vecs
    u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=1)
    print("  operators %s, vectors %s: %d iterations, final %.2e" % ("f32" if mats else "f64", "f32" if vecs else "f64", len(hist), hist[-1]), flush=True)
