"""w at the centre of the BASELINE panel under uniform pressure against Navier's series, over mesh sizes and over mathematically
equivalent FP64 assemblies (node renumbering = another summation order; a rigid translation of the mesh; the other diagonal
orientation): how much of the deviation at 4M triangles is the conditioning of K meeting the rounding of its entries?"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.helpers import fullsize, meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
navier = fullsize.navier_centre_deflection(300.0, 10.0, 1e7, 0.3, 0.5)
print("Navier series: %.10f" % navier)


def run(n, shift=(0.0, 0.0, 0.0), flags=0, ul_lr=True, label=""):
    m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=ul_lr, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
    xyz = m.xyz + np.asarray(shift)[None, :]
    fs = pkg.FemShell(0.3, 1e7, 0.5, device=0, flags=flags)
    fs.set_mesh(xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.assemble()
    fs.set_preconditioner("amg", refine_passes=1)
    u, info = fs.solve(rtol=1e-10, max_it=3000)
    c = (n // 2) * (n + 1) + n // 2
    w = float(u[c, 2])
    print("n %5d %-34s w_c %.10f  (w_c - Navier)/Navier %+.3e  iterations %d estimate %.1e" % (n, label, w, (w - navier) / navier, info["iterations"],
          info["error_estimate"]), flush=True)
    fs.close()
    return w


for n in (64, 128, 256, 354, 512, 708, 1000, 1414):
    run(n, label="as generated")
for n in (708, 1414):
    run(n, shift=(3.0, 7.0, 0.0), label="translated by (3, 7, 0)")
    run(n, shift=(-5.0, -5.0, 0.0), label="centred on the origin")
    run(n, flags=pkg.REORDER_MORTON, label="Morton numbering")
    run(n, ul_lr=False, label="other diagonal")
