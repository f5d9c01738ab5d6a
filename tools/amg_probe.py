"""Multigrid-preconditioned solve on a BASELINE-size mesh: iterations, time to solution, hierarchy facts.
usage: amg_probe.py panel|roof|cylinder N [V|K] [rtol]"""
import importlib
import json
import sys
import time

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
kind, n = sys.argv[1], int(sys.argv[2])
cycle = sys.argv[3] if len(sys.argv) > 3 else "K"
rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-10
if kind == "panel":
    m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
    mat = (0.3, 1e7, 0.5)
elif kind == "roof":
    m = meshes.scordelis_lo(n)
    mat = m.material
else:
    m = meshes.pinched_cylinder(n, n)
    mat = m.material
fs = pkg.FemShell(*mat, device=0)
t0 = time.time()
fs.set_mesh(m.xyz, m.tri, m.quad)
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(m.loads)
fs.assemble()
print("mesh %s %d: %d tri, symbolic+assembly %.2fs" % (kind, n, len(m.tri), time.time() - t0), flush=True)
fs.set_preconditioner("amg", cycle=cycle)
t0 = time.time()
u, info = fs.solve(rtol=rtol, max_it=3000, fetch=False)
wall = time.time() - t0
print(json.dumps({"wall_s": wall, **info}))
for l in fs.amg_levels():
    print("  ", l)
st = fs.amg_setup_stats()
print("first coarsening step on the device: P %.2f ms, AP %.2f ms, R %.2f ms, Galerkin %.2f ms (%s): %.1f GFLOP useful%s" % (
    st["prolongator_ms"], st["ap_ms"], st["restriction_ms"], st["galerkin_ms"],
    "v_mfma_f64_16x16x4_f64" if st["galerkin_on_matrix_cores"] else "vector ALUs", st["galerkin_useful_flops"] / 1e9,
    ", %.1f GFLOP issued on the matrix cores = %.1f TFLOP/s" % (st["galerkin_mfma_flops_issued"] / 1e9,
                                                               st["galerkin_mfma_flops_issued"] / max(st["galerkin_ms"], 1e-9) / 1e9)
    if st["galerkin_on_matrix_cores"] else ""))
ds = fs.amg_dense_stats()
if ds["n"]:
    print("dense inverse of the coarsest operator on the matrix cores: n = %d, %.2f ms, %.1f GFLOP issued = %.1f TFLOP/s (%.1f useful), "
          "%.1f GB moved = %.2f TB/s, %d dropped directions" % (ds["n"], ds["ms"], ds["mfma_flops_issued"] / 1e9,
          ds["mfma_flops_issued"] / ds["ms"] / 1e9, ds["useful_flops"] / ds["ms"] / 1e9, ds["bytes"] / 1e9, ds["bytes"] / ds["ms"] / 1e9, ds["dropped_directions"]))
u2, info2 = fs.solve(rtol=rtol, max_it=3000, fetch=False)
print("second solve (hierarchy reused):", json.dumps(info2))
h = fs.residual_history()
print("residual history (every 10th):", " ".join("%.1e" % v for v in h[::10]))
