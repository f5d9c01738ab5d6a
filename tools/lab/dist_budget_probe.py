"""Per-level kernel times of the row-partitioned multigrid cycle at the per-GPU size of an N-rank run of the 4M-triangle panel,
measured on ONE GPU: a strip of 1414 x (1414 / N) squares on a one-rank RCCL communicator (FEMSHELL_FORCE_COMM=1: the
row-partitioned code path with real ncclAllReduce launches, no peers) with FEMSHELL_AMG_DIST_MIN lowered so that level 1 stays
split like the 223k-node level of the full mesh does.  Run under rocprofv3 --kernel-trace and feed the trace to
tools/kernel_trace_by_grid.py:    python tools/lab/dist_budget_probe.py N"""
import importlib, json, os, sys
sys.path.insert(0, ".")
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ny = 1414 // N
m = meshes.structured(1414, ny, 0, 0, 10, 10.0 / N, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
os.environ["FEMSHELL_FORCE_COMM"] = "1"
os.environ["FEMSHELL_AMG_DIST_MIN"] = str(max(1400, 18000 // N))
fs = pkg.FemShell(0.3, 1e7, 0.5, rank=0, world_size=1)
fs.comm_init(pkg.comm_unique_id())
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
fs.assemble()
fs.set_preconditioner("amg")
u, info = fs.solve(rtol=1e-10, max_it=400, fetch=False)
u, info = fs.solve(rtol=1e-10, max_it=400, fetch=False)
print(json.dumps({"N": N, "triangles": len(m.tri), "iterations": info["iterations"], "solve_seconds": info["solve_seconds"],
                  "ms_per_iteration": 1e3 * info["solve_seconds"] / info["iterations"], "levels": info["amg_levels"],
                  "partition": fs.amg_partition_info(), "levels_info": fs.amg_levels()}))
