"""Assembly and CG rates on a structured QUAD4 mesh (profiling aid; no BASELINE config uses quads)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.time()
m = meshes.structured(nx, nx, 0, 0, 10, 10, kind="q", bcids=(0, 0, 0, 0), factor=300.0, loading=2)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri, m.quad); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
print("setup %.1f s, %d quads, %d nodes" % (time.time() - t0, len(m.quad), m.n_nodes))
fs.assemble()
ms, by = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 5)
print("%s (QUAD4): %.3f ms  %.1f Melem/s  %.0f GB/s" % (fs.assembly_kernel(), ms, len(m.quad) / ms / 1e3, by / ms / 1e6))
_, info = fs.solve(rtol=0.0, max_it=200, fetch=False)
print("cg: %.4f ms/iter" % (1e3 * info["solve_seconds"] / info["iterations"]))
