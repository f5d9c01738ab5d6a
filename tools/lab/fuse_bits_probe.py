"""Which of the fused passes (FEMSHELL_AMG_FUSE bits) changes bits of the solution?  roof 64, all FP64."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
m = meshes.scordelis_lo(64)
os.environ["FEMSHELL_AMG_SMOOTH_F32"] = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["FEMSHELL_AMG_VEC_F32"] = sys.argv[2] if len(sys.argv) > 2 else "0"
refine = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cycle = sys.argv[4] if len(sys.argv) > 4 else "K"
print("SMOOTH_F32 %s VEC_F32 %s refine_passes %d cycle %s" % (os.environ["FEMSHELL_AMG_SMOOTH_F32"], os.environ["FEMSHELL_AMG_VEC_F32"], refine, cycle))
ref = None
for fuse in ("0", "0", "1", "2", "4", "7"):
    os.environ["FEMSHELL_AMG_FUSE"] = fuse
    fs = pkg.FemShell(*m.material, device=0)
    fs.set_mesh(m.xyz, m.tri, m.quad)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.set_preconditioner("amg", coarsest_nodes=60, refine_passes=refine, cycle=cycle)
    u, info = fs.solve(rtol=1e-12, max_it=500)
    if ref is None:
        ref = u
    print("FUSE=%s: %d iterations, levels %d, differing entries %d, max diff %.2e" % (fuse, info["iterations"], info["amg_levels"],
          int((u != ref).sum()), np.abs(u - ref).max()), flush=True)
    fs.close()
