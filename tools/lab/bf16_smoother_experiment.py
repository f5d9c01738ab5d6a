"""What would a 16-bit copy of the level operators cost the smoother?  The library's Chebyshev products (and the two residual
increments of a cycle) read a single-precision copy of the operator's values (146 B per block instead of 292).  This experiment
rounds that copy further -- to bfloat16 (8 significant bits) or to an 11-bit significand (IEEE half without its range limit) --
in the numpy restatement, everything else as the library has it (D^-1, P, R and the cycle's vectors in single precision on
levels of at least 4096 nodes, FP64 Krylov and K-cycle products).   python tools/lab/bf16_smoother_experiment.py panel|roof|cyl|flap NX [min_nodes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import meshes, oracle  # noqa: E402
import amg_oracle as ao  # noqa: E402

MODE = {"mats": False, "vecs": False, "min_nodes": 4096, "sig": 24, "diag_sig": 24}


def f32(v):
    return v.astype(np.float32).astype(np.float64)


def low(L):
    return L.n >= MODE["min_nodes"]


def rbits(v, sig):
    """round to `sig` significant bits (24 = float32)"""
    if sig >= 24:
        return f32(v)
    m, e = np.frexp(v)
    s = 2.0 ** sig
    return np.ldexp(np.round(m * s) / s, e)


def mat32(M, sig=24):
    M = M.copy()
    M.data = rbits(M.data, sig)
    return M


def prepare(levels):
    for li, L in enumerate(levels[:-1]):
        L.A32, L.Dm32 = mat32(L.A, MODE["sig"]), mat32(L.Dm)
        if MODE["diag_sig"] != MODE["sig"]:  # the diagonal blocks of the copy at another width
            Ab = L.A32.tobsr((6, 6))
            Af = mat32(L.A, MODE["diag_sig"]).tobsr((6, 6))
            for i in range(Ab.shape[0] // 6):
                for q in range(Ab.indptr[i], Ab.indptr[i + 1]):
                    if Ab.indices[q] == i:
                        Ab.data[q] = Af.data[q]
            L.A32 = Ab.tocsr()
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
print(which, NX, "levels", [L.n for L in levels], "reduced precision on levels of >=", MODE["min_nodes"], "nodes", flush=True)
for sig, dsig, name in ((53, 53, "FP64 everywhere"), (24, 24, "float32 copy (the library today)"), (11, 11, "11-bit significand (half)"), (8, 8, "bfloat16"),
                        (8, 24, "bfloat16, diagonal blocks float32"), (11, 24, "half, diagonal blocks float32")):
    MODE["sig"], MODE["diag_sig"] = sig, dsig
    MODE["mats"] = MODE["vecs"] = sig < 53
    if sig < 53:
        prepare(levels)
    u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=1)
    print("  smoother's copy of A: %-36s %d iterations, final %.2e" % (name, len(hist), hist[-1]), flush=True)
