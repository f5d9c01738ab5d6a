"""Chebyshev smoothers of the 4th kind (Lottes 2023, "Optimal polynomial smoothers for multigrid V-cycles") against the
1st-kind polynomial on [lam/30, lam] the library uses, in the numpy restatement.  python tools/lab/cheb4_experiment.py panel|roof|cyl NX"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle, meshes
import amg_oracle as ao

# optimized weights beta_k of the 4th-kind smoother (Lottes 2023, table 1) for degrees 1..6
OPT = {1: [1.12500000000000], 2: [1.02387287570313, 1.26408905371085], 3: [1.00842544782028, 1.08867839208730, 1.33753125909618],
       4: [1.00391310427285, 1.04035811188593, 1.14863498546254, 1.38268869241000],
       5: [1.00212930146164, 1.02173711549260, 1.07872433192603, 1.19810065292663, 1.41322542791682],
       6: [1.00128517255940, 1.01304293035233, 1.04678215124113, 1.11616489419675, 1.23829020218444, 1.43524297106744]}
MODE = {"kind": "first"}

def smooth(L, b, x):
    deg = len(L.cheb) + 1
    if MODE["kind"] == "first":
        return ao_smooth(L, b, x)
    rho = L.lam
    if x is None:
        r = b; x = np.zeros_like(b)
    else:
        r = b - L.A @ x
    beta = OPT[deg] if MODE["kind"] == "fourth_opt" else [1.0] * deg
    d = (4.0 / (3.0 * rho)) * (L.Dm @ r)
    for k in range(1, deg):
        x = x + beta[k - 1] * d
        r = r - L.A @ d
        d = ((2 * k - 1) / (2 * k + 3)) * d + ((8 * k + 4) / ((2 * k + 3) * rho)) * (L.Dm @ r)
    return x + beta[deg - 1] * d

ao_smooth = ao.smooth
ao.smooth = smooth
which, NX = sys.argv[1], int(sys.argv[2])
if which == "panel":
    m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2); mat = (0.3, 1e7, 0.5)
elif which == "roof":
    m = meshes.scordelis_lo(NX); mat = m.material
else:
    m = meshes.pinched_cylinder(NX, NX); mat = m.material
r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), dirichlet=m.dirichlet_mask(), loads=m.loads)
A = oracle.to_scipy(r, c, v).tobsr((6, 6))
for deg, cdeg in ((3, 4), (2, 3), (2, 4), (3, 3)):
    levels = ao.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=60, tri=m.tri, degree=deg, coarse_degree=cdeg)
    out = []
    for kind in ("first", "fourth", "fourth_opt"):
        MODE["kind"] = kind
        u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=1)
        out.append("%s %d" % (kind, len(hist)))
    print(which, NX, "levels", [L.n for L in levels], "degrees %d/%d:" % (deg, cdeg), ", ".join(out), flush=True)
