"""Assembly and CG rates on an unstructured (Delaunay) shell mesh, nodes numbered along a Morton curve
(profiling aid; BASELINE's configurations are structured).
python tools/unstructured_probe.py [points] [numbering]     numbering: morton (caller numbers along a Morton curve, default),
shuffled (random caller numbering), shuffled+morton / shuffled+rcm (random caller numbering, the library renumbers)
a third argument "amg" adds a multigrid solve to 1e-10; FEMSHELL_PROBE_POINTS=jittered takes a grid with jittered interior
points (no slivers on the hull) instead of uniformly random points"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '.')
from scipy.spatial import Delaunay
pkg = importlib.import_module("fem-shell_amd")
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
numbering = sys.argv[2] if len(sys.argv) > 2 else "morton"
rng = np.random.default_rng(5)
if os.environ.get("FEMSHELL_PROBE_POINTS") == "jittered":
    side = int(np.sqrt(n_pts))
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    inner = np.all((g > 0) & (g < side - 1), axis=1)[:, None]
    uv = (g + inner * rng.uniform(-0.35, 0.35, size=g.shape)) / (side - 1)
    n_pts = len(uv)
else:
    uv = rng.uniform(0.0, 1.0, size=(n_pts, 2))
t0 = time.time()
tri = Delaunay(uv).simplices.astype(np.int32)
p, q, r = uv[tri[:, 0]], uv[tri[:, 1]], uv[tri[:, 2]]
area = 0.5 * np.abs((q[:, 0] - p[:, 0]) * (r[:, 1] - p[:, 1]) - (q[:, 1] - p[:, 1]) * (r[:, 0] - p[:, 0]))
tri = tri[area > 1e-3 * area.mean()]
# Morton numbering of the nodes
ix = (uv[:, 0] * 65535).astype(np.int64); iy = (uv[:, 1] * 65535).astype(np.int64)
key = np.zeros(n_pts, dtype=np.int64)
for b in range(16):
    key |= ((ix >> b) & 1) << (2 * b)
    key |= ((iy >> b) & 1) << (2 * b + 1)
order = np.argsort(key, kind="stable") if numbering == "morton" else rng.permutation(n_pts); inv = np.empty(n_pts, dtype=np.int64); inv[order] = np.arange(n_pts)
uv = uv[order]; tri = inv[tri].astype(np.int32)
xyz = np.stack([10 * uv[:, 0], 10 * uv[:, 1], 0.5 * np.sin(3 * uv[:, 0]) * np.cos(2 * uv[:, 1])], axis=1)
print("mesh: %d nodes, %d triangles (%.1f s)" % (n_pts, len(tri), time.time() - t0))
flags = pkg.REF_DEFAULT | (pkg.REORDER_MORTON if numbering.endswith("+morton") else pkg.REORDER_RCM if numbering.endswith("+rcm") else 0)
fs = pkg.FemShell(0.3, 1e7, 0.05, flags=flags)
print("numbering:", numbering)
t0 = time.time()
fs.set_mesh(xyz, tri)
print("set_mesh %.2f s" % (time.time() - t0))
dm = np.zeros(n_pts, dtype=np.uint8); dm[(uv[:, 0] < 0.01) | (uv[:, 0] > 0.99)] = 0x3F
fs.set_dirichlet(dm); loads = np.zeros((n_pts, 6)); loads[:, 2] = 1.0; fs.set_loads(loads)
fs.assemble()
for _ in range(8):  # (the first launches of a process run slow: clocks; the last figure is the steady one)
    ms, by = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 20)
print("%s: %.3f ms  %.1f Melem/s  %.0f GB/s (algorithmic)" % (fs.assembly_kernel(), ms, len(tri) / ms / 1e3, by / ms / 1e6))
ms_s, by_s = fs.time_kernel(pkg.KERNEL_SPMV, 20)
print("k_spmv: %.4f ms %.0f GB/s (algorithmic)" % (ms_s, by_s / ms_s / 1e6))
_, info = fs.solve(rtol=0.0, max_it=300, fetch=False)
print("cg: %.4f ms/iter" % (1e3 * info["solve_seconds"] / info["iterations"]))
if len(sys.argv) > 3 and sys.argv[3] == "amg":
    fs.set_preconditioner("amg")
    _, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
    print("amg: %d iterations, converged %d, %.3f s solve + %.3f s setup, %d levels, true residual %.2e" % (
        info["iterations"], info["converged"], info["solve_seconds"], info["pc_setup_seconds"], info["amg_levels"],
        info["true_rel_residual"]))
