import sys, importlib; sys.path.insert(0,'.')
import numpy as np
from tests.helpers import meshes, oracle
pkg = importlib.import_module("fem-shell_amd")
m = meshes.structured(32, 16, 0, 0, 48, 12, kind="t", ul_lr=True, bcids=(-1, -1, 1, -1))
tip = 8*33+32; m.loads[tip, 2] = 1.0; m.loads[tip, 1] = 40.0
nu,E,t = 0.25, 30000.0, 1.0
mat = oracle.material(nu,E,t)
r0,c0,v0,F0 = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
u0 = oracle.refined_solve(r0,c0,v0,F0)
n = np.linalg.norm
for rtol in (1e-10, 1e-12, 1e-13, 1e-14):
    fs = pkg.FemShell(nu,E,t); fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    u, info = fs.solve(rtol=rtol, max_it=200000)
    h = fs.residual_history()
    x, oi = oracle.pcg(r0,c0,v0,F0, rtol=rtol, max_it=200000, history=True)
    print("rtol %g: gpu its %d conv %d err %.2e trueres %.2e | cpu its %d err %.2e trueres %.2e | hist ratio @2000 %.3f" % (
        rtol, info["iterations"], info["converged"], n(u.ravel()-u0)/n(u0), n(F0-oracle.spmv(r0,c0,v0,u.ravel()))/n(F0),
        oi["iterations"], n(x-u0)/n(u0), n(F0-oracle.spmv(r0,c0,v0,x))/n(F0), h[min(2000,len(h)-1)]/oi["history"][min(2000,len(oi["history"])-1)]))
