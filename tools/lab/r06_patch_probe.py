"""Round 6: the multigrid solve of random-point Delaunay shells (poor element quality) with and without the patch smoother
(csrc/amg_patch.hpp):  python tools/lab/r06_patch_probe.py [points=20000] [seed=3] [max_it=1500]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import delaunay_shell
pkg = importlib.import_module("fem-shell_amd")
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
jit = len(sys.argv) > 4 and sys.argv[4] == "jittered"
xyz, tri = delaunay_shell(n_pts, seed, jittered=jit)
if os.environ.get("NUMBERING") == "x":  # numbered along x, as tests/helpers/multirank_worker.py numbers the shell for several ranks
    order = np.argsort(xyz[:, 0], kind="stable")
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    xyz, tri = np.ascontiguousarray(xyz[order]), inv[tri].astype(np.int32)
n = len(xyz)
dmask = np.zeros(n, dtype=np.uint8); dmask[xyz[:, 0] < 0.15] = 0x3F
loads = np.zeros((n, 6)); loads[:, 2] = 1.0
for tau in (os.environ.get("TAUS", "0.8,0").split(",")):
    os.environ["FEMSHELL_AMG_PATCH_TAU"] = tau
    fs = pkg.FemShell(0.3, 7.0e4, 0.03)
    fs.set_mesh(xyz, tri); fs.set_dirichlet(dmask); fs.set_loads(loads)
    fs.set_preconditioner("amg")
    t0 = time.time()
    try:
        u, info = fs.solve(rtol=1e-10, max_it=max_it)
        print("tau %s: %d nodes  its %d conv %d  levels %s  solve %.3f s setup %.3f s  fp64 fallback %s  patch %s  err_est %.2e" % (
            tau, n, info["iterations"], info["converged"], [l["n_nodes"] for l in fs.amg_levels()], info["solve_seconds"], info["pc_setup_seconds"],
            info.get("pc_fp64_fallback"), fs.amg_patch_info(), info["error_estimate"]), flush=True)
    except pkg.FemShellError as ex:
        print("tau %s: ERROR %s (%.1f s)" % (tau, ex, time.time() - t0), flush=True)
    fs.close()
