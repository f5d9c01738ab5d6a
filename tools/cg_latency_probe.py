"""Fixed cost of one multi-rank CG iteration, measured on one GPU: the single-reduction recurrence through a one-rank RCCL
communicator (FEMSHELL_FORCE_COMM=1: real ncclAllReduce launches, no peers), folded (FEMSHELL_CG_FOLD=1) against unfolded,
on the panel at the per-GPU sizes of an 8-, 4-, 2- and 1-GPU run of the 4M-triangle mesh.
    python tools/cg_latency_probe.py [nx ...]"""
import importlib
import os
import sys
import time

sys.path.insert(0, ".")
from bench import panel_mesh  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")


def run(nx, comm, fold, single, iters=400):
    os.environ["FEMSHELL_CG_FOLD"] = "1" if fold else "0"
    os.environ["FEMSHELL_CG_SINGLE_REDUCTION"] = "1" if single else "0"
    if comm:
        os.environ["FEMSHELL_FORCE_COMM"] = "1"
    else:
        os.environ.pop("FEMSHELL_FORCE_COMM", None)
    m = panel_mesh(nx)
    fs = pkg.FemShell(0.3, 1e7, 0.5, rank=0, world_size=1)
    if comm:
        fs.comm_init(pkg.comm_unique_id())
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.assemble()
    fs.solve(rtol=0.0, max_it=100, fetch=False)
    best = 1e9
    for _ in range(3):
        _, info = fs.solve(rtol=0.0, max_it=iters, fetch=False)
        best = min(best, 1e3 * info["solve_seconds"] / info["iterations"])
    fs.close()
    return best


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [500, 707, 1000, 1414]
    print("%6s %10s | %12s %12s %12s %12s" % ("nx", "triangles", "classic", "1-red comm", "1-red fold", "classic comm"))
    for nx in sizes:
        t0 = run(nx, False, True, False)
        t1 = run(nx, True, False, True)
        t2 = run(nx, True, True, True)
        t3 = run(nx, True, True, False)
        print("%6d %10d | %9.4f ms %9.4f ms %9.4f ms %9.4f ms" % (nx, 2 * nx * nx, t0, t1, t2, t3), flush=True)
