#!/usr/bin/env python3
"""Writes tests/golden/tri3_elements.npz: element matrices of a fixed set of TRI3 shapes
computed by the CPU oracle (oracle/femshell_oracle.c), which itself is pinned to the thesis
known answers.  These are regression vectors (oracle-generated, not reference output: the
reference cannot be built here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import oracle  # noqa: E402

rng = np.random.default_rng(20151201)
shapes = [
    [[0, 0, 0], [1, 0, 0], [0, 1, 0]],               # isosceles right
    [[0, 0, 0], [2, 0, 0], [0, 1, 0]],               # 2:1 right (SA:586 active)
    [[0, 0, 0], [0.125, 0, 0], [0, 0, 0.1]],         # the coupled example's cell shape
    [[0.3, -1.0, 2.0], [4.1, 0.2, 2.5], [-2.0, 0.7, 3.1]],
    [[0, 0, 0], [1, 0, 0], [0.5, 1e-3, 0]],          # sliver
    [[5, 5, 5], [5, 6, 5], [5, 5, 6.5]],
]
for _ in range(10):
    shapes.append((rng.normal(size=(3, 3)) * rng.uniform(0.1, 5.0)).tolist())
xyz = np.array(shapes, dtype=np.float64).reshape(-1, 3)
tri = np.arange(len(xyz), dtype=np.int32).reshape(-1, 3)
nu, E, t = 0.3, 1.0e7, 0.5
mat = oracle.material(nu, E, t)
Ke = np.stack([oracle.element_tri3(xyz[c], mat) for c in tri])
out = os.path.join(ROOT, "tests", "golden", "tri3_elements.npz")
np.savez(out, xyz=xyz, tri=tri, Ke=Ke, nu=nu, E=E, t=t)
print(out, Ke.shape)
