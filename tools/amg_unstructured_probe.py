"""Multigrid on unstructured Delaunay shells: random points (slivers on the hull kept down to an area of 1e-7), a jittered
grid (irregular valence, no slivers), with random and with Morton node numbering:  python tools/amg_unstructured_probe.py"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import oracle
from tests.test_gpu_parity import delaunay_shell
pkg = importlib.import_module("fem-shell_amd")


def run(name, xyz, tri, pc="amg", flags=None, max_it=3000):
    n = len(xyz)
    dmask = np.zeros(n, dtype=np.uint8)
    dmask[xyz[:, 0] < 0.15] = 0x3F
    loads = np.zeros((n, 6))
    loads[:, 2] = 1.0
    kw = {} if flags is None else {"flags": flags}
    fs = pkg.FemShell(0.3, 7.0e4, 0.03, **kw)
    fs.set_mesh(xyz, tri)
    fs.set_dirichlet(dmask)
    fs.set_loads(loads)
    if pc == "amg":
        fs.set_preconditioner("amg")
    try:
        u, info = fs.solve(rtol=1e-10, max_it=max_it)
        lv = [l.get("n_nodes", l) if isinstance(l, dict) else l for l in fs.amg_levels()] if pc == "amg" else []
        print("%-50s %6d tri %5d its conv %d  %.3f s  levels %s" % (name, len(tri), info["iterations"], info["converged"], info["solve_seconds"], lv), flush=True)
    except pkg.FemShellError as ex:
        print("%-50s ERROR %s" % (name, ex), flush=True)
    fs.close()


for npts in (3000, 20000):
    xyz, tri = delaunay_shell(npts, 3)
    # triangle quality
    p, q, r = xyz[tri[:, 0]], xyz[tri[:, 1]], xyz[tri[:, 2]]
    a = np.linalg.norm(q - p, axis=1); b = np.linalg.norm(r - q, axis=1); c = np.linalg.norm(p - r, axis=1)
    area = 0.5 * np.linalg.norm(np.cross(q - p, r - p), axis=1)
    qual = 4 * np.sqrt(3) * area / (a * a + b * b + c * c)
    print("random points %d: quality min %.2e, 1%% %.2e, median %.2f" % (npts, qual.min(), np.quantile(qual, 0.01), np.median(qual)))
    run("random points %d" % npts, xyz, tri)
    keep = qual > 0.05
    used = np.unique(tri[keep])
    remap = -np.ones(len(xyz), dtype=np.int64); remap[used] = np.arange(len(used))
    run("random points %d, slivers (q < 0.05) removed" % npts, xyz[used], remap[tri[keep]].astype(np.int32))
    xj, tj = delaunay_shell(npts, 3, jittered=True)
    run("jittered grid %d" % npts, xj, tj)
    run("jittered grid %d, block-Jacobi" % npts, xj, tj, pc="jacobi", max_it=200000)

# aggregation follows the node numbering: shuffled (above) against a Morton curve through the mesh
for npts in (20000, 250000):
    xj, tj = delaunay_shell(npts, 3, jittered=True)
    run("jittered grid %d, shuffled numbering" % npts, xj, tj)
    run("jittered grid %d, Morton renumbering" % npts, xj, tj, flags=pkg.REF_DEFAULT | pkg.REORDER_MORTON)
    run("jittered grid %d, RCM renumbering" % npts, xj, tj, flags=pkg.REF_DEFAULT | pkg.REORDER_RCM)
