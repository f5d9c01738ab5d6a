"""Wall time of the host phases of a one-shot run at the north-star size (4M triangles): FEMSHELL_PLAN_VERBOSE /
FEMSHELL_AMG_VERBOSE laps of set_mesh and the multigrid setup on the GPU box's cores.  Usage: python setup_phases_probe.py [nx]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["FEMSHELL_PLAN_VERBOSE"] = "1"
os.environ["FEMSHELL_AMG_VERBOSE"] = "1"
from tests.helpers import meshes  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1414
m = meshes.structured(nx, nx, 0, 0, 10.0, 10.0, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
pkg = importlib.import_module("fem-shell_amd")
print("cores", len(os.sched_getaffinity(0)), "triangles", len(m.tri), flush=True)
for rep in range(3):
    fs = pkg.FemShell(0.3, 1.0e7, 0.5, device=0)
    t = time.time()
    fs.set_mesh(m.xyz, m.tri, m.quad)
    t1 = time.time()
    print("== set_mesh %.3f s" % (t1 - t), flush=True)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.set_preconditioner("amg")
    fs.assemble()
    t2 = time.time()
    print("== dirichlet, loads, first assembly %.3f s" % (t2 - t1), flush=True)
    u, info = fs.solve(rtol=1e-8, max_it=2000)
    print("== solve %.3f s, pc setup %.3f s, %d iterations" % (time.time() - t2, info.get("pc_setup_seconds", -1), info["iterations"]), flush=True)
    fs.close()
