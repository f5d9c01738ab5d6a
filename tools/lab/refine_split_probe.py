"""How far does the first pass of a multigrid solve have to go?  Manufactured solutions at full size: true error,
estimated error and iteration counts for main-pass tolerances 1e-10 ... 1e-5 with one (or up to three) refinement passes
(lab probe behind the refinement rule of csrc/amg_solve.cpp).   python tools/lab/refine_split_probe.py [panel|cylinder] [n]"""
import importlib, sys
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import fullsize, manufactured
pkg = importlib.import_module("fem-shell_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "panel"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1414
m, mat = fullsize.workload(kind, n)
fs = pkg.FemShell(*mat)
fs.set_mesh(m.xyz, m.tri)
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(np.zeros((m.n_nodes, 6)))
fs.assemble()
u_star = manufactured.smooth_field(m, kind)
b = manufactured.rhs_of(fs, u_star)
fs.set_loads(b)
ref = u_star  # (the rounding correction delta is ~1e-13 relative: enough for this probe)
for passes in (1, 3):
    for rtol in (1e-10, 1e-8, 1e-7, 1e-6, 1e-5):
        fs.set_preconditioner("amg", refine_passes=passes)
        u, info = fs.solve(rtol=rtol, max_it=3000)
        err = np.linalg.norm(u - ref) / np.linalg.norm(ref)
        print("%s passes<=%d rtol %.0e: %3d iterations %.3f s, passes done %d, true error %.2e, estimate %.2e (correction %.2e x reduction %.2e), true residual %.2e"
              % (kind, passes, rtol, info["iterations"], info["solve_seconds"], info["refine_passes_done"], err, info["error_estimate"],
                 info["refine_correction_rel"], info["refine_residual_reduction"], info["true_rel_residual"]), flush=True)
fs.close()
