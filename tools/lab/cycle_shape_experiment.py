"""Smoothing shape of the multigrid cycle, tried in the numpy restatement: pre/post Chebyshev degrees per level class and what
they cost in operator products.  python tools/lab/cycle_shape_experiment.py panel|roof|cyl NX"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle, meshes
import amg_oracle as ao

EIG = 30.0
def cheb_coeffs(lam, deg, ratio=EIG):
    lmax, lmin = lam, lam / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    out, rho = [], 1.0 / sigma
    for _ in range(1, deg):
        rho_new = 1.0 / (2.0 * sigma - rho)
        out.append((rho_new * rho, 2.0 * rho_new / delta))
        rho = rho_new
    return 1.0 / theta, out

COUNT = {}
def smooth(L, li, b, x, deg):
    if deg == 0:
        return np.zeros_like(b) if x is None else x
    inv_theta, cheb = cheb_coeffs(L.lam, deg)
    if x is None:
        r = b; x = np.zeros_like(b)
    else:
        r = b - L.A @ x; COUNT[li] = COUNT.get(li, 0) + 1
    d = inv_theta * (L.Dm @ r); x = x + d
    for a, c in cheb:
        r = r - L.A @ d; COUNT[li] = COUNT.get(li, 0) + 1
        d = a * d + c * (L.Dm @ r); x = x + d
    return x

SHAPE = {}
def cycle(levels, li, b, kcycle):
    L = levels[li]
    if li == len(levels) - 1:
        return L.dense_inv @ b
    pre, post = SHAPE["fine"] if li == 0 else SHAPE["coarse"]
    if pre > 0:
        x = smooth(L, li, b, None, pre)
        r = b - L.A @ x; COUNT[li] = COUNT.get(li, 0) + 1
    else:
        x = np.zeros_like(b); r = b
    bc = L.R @ r
    if kcycle and li + 2 < len(levels):
        xc = ao.kcycle_solve(levels, li + 1, bc); COUNT[li + 1] = COUNT.get(li + 1, 0) + 2
    else:
        xc = cycle(levels, li + 1, bc, kcycle)
    x = x + L.P @ xc
    return smooth(L, li, b, x, post)
ao.cycle = cycle

which, NX = sys.argv[1], int(sys.argv[2])
if which == "panel":
    m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2); mat = (0.3, 1e7, 0.5)
elif which == "roof":
    m = meshes.scordelis_lo(NX); mat = m.material
else:
    m = meshes.pinched_cylinder(NX, NX); mat = m.material
r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), dirichlet=m.dirichlet_mask(), loads=m.loads)
A = oracle.to_scipy(r, c, v).tobsr((6, 6))
levels = ao.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=60, tri=m.tri)
nnz = [L.A.nnz for L in levels]
print(which, NX, "levels", [L.n for L in levels], flush=True)
shapes = [((2, 2), (4, 4)), ((0, 3), (0, 6)), ((0, 4), (0, 8)), ((0, 3), (4, 4)), ((0, 4), (4, 4)), ((2, 2), (0, 6)), ((2, 2), (0, 8)),
          ((1, 2), (4, 4)), ((2, 2), (3, 3)), ((0, 3), (3, 3)), ((3, 0), (4, 4)), ((0, 2), (4, 4)), ((0,2),(0,4))]
for fine, coarse in shapes:
    SHAPE["fine"], SHAPE["coarse"] = fine, coarse
    COUNT.clear()
    u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=0)
    its = len(hist)
    work = sum(COUNT.get(l, 0) * nnz[l] for l in COUNT) / nnz[0] + its  # level-0 product equivalents incl. the Krylov product
    print("fine %s coarse %s: %3d iterations, products per iteration by level %s, work %.0f fine products (%.2f per iteration)"
          % (fine, coarse, its, [round(COUNT.get(l, 0) / its, 1) for l in range(len(levels) - 1)], work, work / its), flush=True)
