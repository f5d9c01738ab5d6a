"""Iteration counts of the multigrid-preconditioned solve on a spread of shell problems (rtol 1e-10), with the error against a
direct solve of the exported matrix where the problem is small enough:  python tools/amg_robustness_probe.py"""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import meshes, oracle
pkg = importlib.import_module("fem-shell_amd")


def run(name, xyz, tri, quad, dmask, loads, mat, check=True):
    fs = pkg.FemShell(*mat)
    fs.set_mesh(xyz, tri, quad)
    fs.set_dirichlet(dmask)
    fs.set_loads(loads)
    fs.set_preconditioner("amg")
    try:
        u, info = fs.solve(rtol=1e-10, max_it=3000)
    except pkg.FemShellError as ex:
        print("%-44s ERROR %s" % (name, ex), flush=True)
        fs.close()
        return
    err = float("nan")
    n_el = (0 if tri is None else len(tri)) + (0 if quad is None else len(quad))
    if check and n_el <= 60000:
        rg, cg, vg, Fg = fs.export_bsr()
        ud = oracle.refined_solve(rg, cg, vg, Fg)
        err = np.linalg.norm(u.ravel() - ud) / np.linalg.norm(ud)
    print("%-44s %7d elements  %4d iterations  conv %d  levels %d  solve %.3f s  err %.1e" % (
        name, n_el, info["iterations"], info["converged"], info["amg_levels"], info["solve_seconds"], err), flush=True)
    fs.close()


def plate(nx, ny, kind="t", bc=(0, 0, 0, 0), t=0.5, inplane=False, lx=10.0, ly=10.0):
    m = meshes.structured(nx, ny, 0, 0, lx, ly, kind=kind, ul_lr=True, bcids=bc, factor=300.0, loading=2)
    loads = m.loads.copy()
    if inplane:
        loads[:] = 0.0
        loads[:, 0] = 1.0
    return m.xyz, m.tri, m.quad, m.dirichlet_mask(), loads, (0.3, 1e7, t)


run("panel ss 128x128 tri, pressure", *plate(128, 128))
run("panel ss 128x128 tri, in-plane load", *plate(128, 128, inplane=True))
run("panel clamped 128x128 quads, pressure", *plate(128, 128, kind="q", bc=(1, 1, 1, 1)))
run("cantilever strip 256x16 tri, t=0.01", *plate(256, 16, bc=(-1, -1, 1, -1), t=0.01, lx=16.0, ly=1.0))
run("cantilever strip 256x16 quads, t=0.01", *plate(256, 16, kind="q", bc=(-1, -1, 1, -1), t=0.01, lx=16.0, ly=1.0))
# folded plate: the right half of a clamped panel bent up by 90 degrees along x = 5
x, tri, quad, dm, ld, mat = plate(96, 96, bc=(-1, -1, 1, -1))
xf = x.copy()
right = x[:, 0] > 5.0
xf[right, 0] = 5.0
xf[right, 2] = x[right, 0] - 5.0
ldf = np.zeros_like(ld)
ldf[:, 2] = 1.0
run("folded plate 96x96 tri (90 degree kink)", xf, tri, quad, dm, ldf, mat)
ldf2 = np.zeros_like(ld)
ldf2[:, 1] = 1.0
run("folded plate 96x96 tri, load along the fold", xf, tri, quad, dm, ldf2, mat)
m = meshes.pinched_cylinder(128, 128)
run("pinched cylinder 128x128", m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, m.material)
m = meshes.scordelis_lo(128)
run("Scordelis-Lo roof 128", m.xyz, m.tri, m.quad, m.dirichlet_mask(), m.loads, m.material)
# unstructured Delaunay shell (tests/test_gpu_parity.py)
from tests.test_gpu_parity import delaunay_shell
xyz, tri = delaunay_shell(20000, 3)
n = len(xyz)
dmask = np.zeros(n, dtype=np.uint8)
dmask[xyz[:, 0] < 0.15] = 0x3F
loads = np.zeros((n, 6))
loads[:, 2] = 1.0
run("Delaunay shell, 20k points, clamped side", xyz, tri, None, dmask, loads, (0.3, 7.0e4, 0.03), check=True)
for name in ("test_A_uv_t", "test_C_w_tA16", "test_E_uvw_t", "test_F_032_ss_uni", "test_G_mpi_64_q", "bending_tower_tri_test"):
    try:
        m = meshes.load_example(name)
    except Exception as ex:
        print(name, "not loadable:", ex)
        continue
    mat = {"test_A_uv_t": (0.25, 30000.0, 1.0), "test_C_w_tA16": (0.3, 10.92, 1.0), "test_E_uvw_t": (0.25, 10000.0, 0.25),
           "test_F_032_ss_uni": (0.3, 1.7472e7, 0.01), "test_G_mpi_64_q": (0.3, 1e7, 0.5), "bending_tower_tri_test": (0.3, 1e6, 0.1)}[name]
    loads = m.loads if np.abs(m.loads).max() > 0 else np.tile([1.0, 0, 0, 0, 0, 0], (m.n_nodes, 1))
    run("shipped " + name, m.xyz, m.tri, m.quad, m.dirichlet_mask(), loads, mat)
