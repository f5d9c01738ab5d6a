"""Notay's rule for the K cycle (skip the second inner step when the first one brought the residual below tol x ||b||), tried in
the numpy restatement:  python tools/lab/kcycle_skip_experiment.py panel|roof|cyl NX [tol]   """
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle, meshes
import amg_oracle as ao

stats = {"calls": 0, "skipped": 0, "ratios": []}
TOL = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25

def kcycle_skip(levels, li, rc):
    A = levels[li].A
    c1 = ao.cycle(levels, li, rc, True)
    v1 = A @ c1
    rho1, a1 = c1 @ v1, c1 @ rc
    t = a1 / rho1 if rho1 > 0.0 else 0.0
    r2 = rc - t * v1
    stats["calls"] += 1
    ratio = np.linalg.norm(r2) / max(np.linalg.norm(rc), 1e-300)
    stats["ratios"].append(ratio)
    if TOL > 0 and ratio <= TOL:
        stats["skipped"] += 1
        return t * c1
    c2 = ao.cycle(levels, li, r2, True)
    v2 = A @ c2
    g, b2, a2 = c2 @ v1, c2 @ v2, c2 @ r2
    w1, w2 = t, 0.0
    if rho1 > 0.0:
        rho2 = b2 - g * g / rho1
        if rho2 > 0.0:
            w1 = a1 / rho1 - g * a2 / (rho1 * rho2)
            w2 = a2 / rho2
    return w1 * c1 + w2 * c2

ao.kcycle_solve = kcycle_skip
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
u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=0)
ra = np.array(stats["ratios"])
print("%s %d tol %.2f levels %s iterations %d  kcycle calls %d skipped %d  ratio ||r2||/||b||: min %.2f median %.2f max %.2f" % (which, NX, TOL, [L.n for L in levels], len(hist), stats["calls"], stats["skipped"], ra.min(), np.median(ra), ra.max()), flush=True)
