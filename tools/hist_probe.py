import importlib, sys
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell
pkg = importlib.import_module("fem-shell_amd")
for n_pts, seed in [(700, 1), (3000, 2)]:
    xyz, tri = delaunay_shell(n_pts, seed)
    n = len(xyz)
    rng = np.random.default_rng(seed)
    dmask = np.zeros(n, dtype=np.uint8)
    dmask[xyz[:, 0] < 0.15] = 0x3F
    dmask[rng.integers(0, n, 5)] |= 0x07
    loads = rng.normal(size=(n, 6))
    fs = pkg.FemShell(0.3, 7.0e4, 0.03)
    fs.set_mesh(xyz, tri); fs.set_dirichlet(dmask); fs.set_loads(loads); fs.assemble()
    mat = oracle.material(0.3, 7.0e4, 0.03)
    r0, c0, v0, F0 = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), mat, dmask, loads)
    fs.solve(rtol=0.0, max_it=200, fetch=False)
    _, info0 = oracle.pcg(r0, c0, v0, F0, rtol=0.0, max_it=200, history=True)
    h = fs.residual_history(); h0 = np.array(info0["history"])
    d = np.abs(h[:60] - h0[:60]) / h0[:60]
    print(n_pts, " ".join("%.1e" % x for x in d[::4]))
