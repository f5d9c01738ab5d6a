import importlib, sys, os
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
m = panel_mesh(1414)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
fs.assemble()
for rep in range(6):
    ms, by = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 10)
    print("round %d: %.3f ms" % (rep, ms))
import time
t0=time.time()
for i in range(20): fs.assemble()
print("20 femshell_assemble calls: %.3f ms each (wall)" % ((time.time()-t0)/20*1e3))
