import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_parity import delaunay_shell
from tests.helpers import oracle, meshes
from oracle import amg_oracle as ao

def run(tag, xyz, tri, dm, t=0.03, **kw):
    n = len(xyz)
    rng = np.random.default_rng(3)
    loads = np.zeros((n, 6)); loads[:, 2] = 1.0
    r, c, v, F = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), oracle.material(0.3, 7e4, t), dm, loads)
    A = oracle.to_scipy(r, c, v).tobsr((6, 6))
    levels = ao.setup(A, xyz, dm, **kw)
    x, hist = ao.solve(A, F, levels, kcycle=True, rtol=1e-10, max_it=800)
    print(tag, "n", n, "levels", [L.n for L in levels], "its", len(hist), "lam", [round(L.lam, 2) for L in levels[:-1]], flush=True)

mode = sys.argv[1]
if mode == "jit":
    xyz, tri = delaunay_shell(2500, 5, jittered=True)
    dm = np.zeros(len(xyz), np.uint8); dm[xyz[:, 0] < 0.2] = 0x3F
    run("jittered curved", xyz, tri, dm)
    flat = xyz.copy(); flat[:, 2] = 0
    run("jittered flat", flat, tri, dm)
if mode == "struct":
    m = meshes.structured(49, 49, 0.0, 0.0, 3.0, 2.0, "t")
    dm = np.zeros(m.n_nodes, np.uint8); dm[m.xyz[:, 0] < 0.2] = 0x3F
    run("structured flat", m.xyz, m.tri, dm)
    xyz = m.xyz.copy(); xyz[:, 2] = 0.3 * np.sin(xyz[:, 0]) * np.cos(xyz[:, 1])
    run("structured curved", xyz, m.tri, dm)
