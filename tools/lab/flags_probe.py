import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("fem-shell_amd")
from tests.helpers import meshes
m = meshes.structured(1414, 1414, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri)
dm = m.dirichlet_mask()
for k in range(4):
    fs.sync(); t0 = time.perf_counter(); fs.set_dirichlet(dm); fs.sync(); t1 = time.perf_counter()
    fs.assemble(); fs.sync(); t2 = time.perf_counter()
    print("set_dirichlet %.2f ms, first assemble after it %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
