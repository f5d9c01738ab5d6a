#!/usr/bin/env python3
"""Writes tests/golden/quad4_elements.npz and tests/golden/example_solutions.npz.

quad4_elements: element matrices (variable-major 24x24, plus membrane 8x8, plate 12x12) of a fixed set of
planar QUAD4 shapes.  example_solutions: displacement vectors of the reference's shipped example meshes A-G
(parameters of run_examples.sh) from a refined sparse direct solve of the oracle-assembled system.  Both are
regression vectors computed by the CPU oracle (oracle/femshell_oracle.c), which is pinned to the thesis
tables (tests/test_oracle_known_answers.py); the reference itself cannot be built in this image."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import meshes, oracle  # noqa: E402

EXAMPLES = [("test_A_uv_t", 0.25, 30000.0, 1.0), ("test_B_uv_q", 0.25, 30000.0, 1.0),
            ("test_C_w_tA16", 0.3, 10.92, 1.0), ("test_D_w_q_uni16", 0.3, 1.0e7, 0.5),
            ("test_E_uvw_t", 0.25, 10000.0, 0.25), ("test_F_032_ss_uni", 0.3, 1.7472e7, 0.01),
            ("test_G_mpi_64_q", 0.3, 1.0e7, 0.5)]


def quad_shapes():
    rng = np.random.default_rng(20151202)
    planar = [
        [[0, 0], [1, 0], [1, 1], [0, 1]],               # unit square
        [[0, 0], [2, 0], [2, 1], [0, 1]],               # 2:1 rectangle
        [[0, 0], [1, 0], [1.3, 1], [0.3, 1]],           # parallelogram
        [[0, 0], [2, 0], [1.5, 1], [0.5, 1]],           # trapezoid
        [[0, 0], [1, 0.1], [1.2, 1.3], [-0.1, 0.9]],    # general convex
        [[0, 0], [0.625, 0], [0.625, 0.625], [0, 0.625]],  # the example meshes' cell
    ]
    out = []
    for k, p in enumerate(planar):
        p = np.c_[np.array(p, dtype=np.float64), np.zeros(4)]
        if k >= 2:  # tilt and shift into 3-D
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            p = p @ q.T + rng.normal(size=3)
        out.append(p)
    return np.array(out).reshape(-1, 3)


def main():
    xyz = quad_shapes()
    quad = np.arange(len(xyz), dtype=np.int32).reshape(-1, 4)
    nu, E, t = 0.3, 1.0e7, 0.5
    mat = oracle.material(nu, E, t)
    Ke, Km, Kp = [], [], []
    for c in quad:
        ke, parts = oracle.element_quad4(xyz[c], mat, want_parts=True)
        Ke.append(ke)
        Km.append(parts["Ke_m"])
        Kp.append(parts["Ke_p"])
    out = os.path.join(ROOT, "tests", "golden", "quad4_elements.npz")
    np.savez(out, xyz=xyz, quad=quad, Ke=np.stack(Ke), Ke_m=np.stack(Km), Ke_p=np.stack(Kp), nu=nu, E=E, t=t)
    print(out, np.stack(Ke).shape)

    sols = {}
    for name, nu, E, t in EXAMPLES:
        m = meshes.load_example(name)
        r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(nu, E, t), m.dirichlet_mask(), m.loads)
        u = oracle.refined_solve(r, c, v, F)
        res = np.linalg.norm(F - oracle.spmv(r, c, v, u)) / np.linalg.norm(F)
        print("%-20s n=%5d  residual %.2e" % (name, m.n_nodes, res))
        sols[name] = u.reshape(-1, 6)
        sols[name + "_params"] = np.array([nu, E, t])
    out = os.path.join(ROOT, "tests", "golden", "example_solutions.npz")
    np.savez_compressed(out, **sols)
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
