"""Host-side symbolic phase (femshell_set_mesh) of the 4M-triangle panel, three repetitions:  python tools/set_mesh_time.py [nx]"""
import importlib, sys, time
sys.path.insert(0, ".")
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1414
m = panel_mesh(nx)
for _ in range(3):
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    t0 = time.perf_counter()
    fs.set_mesh(m.xyz, m.tri)
    print("set_mesh %.3f s" % (time.perf_counter() - t0), flush=True)
    fs.close()
