"""Randomised matrix parity sweep on the GPU (one-off stress run, slower than the test suite):
random Delaunay and structured meshes, tri / quad / mixed, random Dirichlet sets and behaviour flags, sorted and
shuffled node numbering; the assembled K and F must equal the oracle's (1e-12 relative to max |K|), and after a
change of the Dirichlet set on the same context as well; the product K x against the oracle's on the same matrix.   python tools/stress_parity.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import meshes, oracle
from tests.test_gpu_parity import delaunay_shell
pkg = importlib.import_module("fem-shell_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
worst_spmv = 0.0
for case in range(cases):
    kind = rng.choice(["delaunay", "tri", "quad", "mixed"])
    if kind == "delaunay":
        xyz, tri = delaunay_shell(int(rng.integers(60, 6000)), int(rng.integers(1, 1 << 30)))
        quad = np.zeros((0, 4), np.int32)
        if rng.random() < 0.5:  # sort along x: narrow slices; otherwise keep the shuffled numbering
            order = np.argsort(xyz[:, 0]); inv = np.empty_like(order); inv[order] = np.arange(len(order))
            xyz, tri = xyz[order], inv[tri].astype(np.int32)
    else:
        nx, ny = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        lx, ly = float(rng.uniform(0.5, 9)), float(rng.uniform(0.5, 9))
        # aspect ratios up to 50:1 (beyond that the element math itself loses digits -- at 1000:1 the oracle and the
        # kernels differ by 2e-12 of max|K| in the diagonal blocks, both in double precision)
        ly = float(np.clip(ly, lx / nx * ny / 50.0, lx / nx * ny * 50.0))
        m = meshes.structured(nx, ny, 0, 0, lx, ly,
                              kind="t" if kind == "tri" else "q", ul_lr=bool(rng.integers(0, 2)))
        xyz, tri, quad = m.xyz.copy(), m.tri.copy(), m.quad.copy()
        xyz[:, 2] = (0.0 if kind != "tri" else 0.2 * np.sin(xyz[:, 0]) * np.cos(0.7 * xyz[:, 1]))
        if kind == "mixed" and len(quad):  # split a random half of the quads into triangles
            pick = rng.random(len(quad)) < 0.5
            q = quad[pick]
            tri = np.concatenate([tri.reshape(-1, 3), q[:, [0, 1, 2]], q[:, [0, 2, 3]]]).astype(np.int32)
            quad = quad[~pick]
        if kind != "tri":  # planar quads, tilted rigidly
            qm, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            xyz = xyz @ qm.T
    n = len(xyz)
    flags = int(rng.integers(0, 4))
    nu, E, t = float(rng.uniform(0.0, 0.45)), float(10 ** rng.uniform(1, 8)), float(10 ** rng.uniform(-2, 0.3))
    fs = pkg.FemShell(nu, E, t, flags=flags)
    fs.set_mesh(xyz, tri, quad)
    kern = fs.assembly_kernel()
    mat = oracle.material(nu, E, t, flags)
    for rep in range(2):
        dmask = np.where(rng.random(n) < rng.uniform(0, 0.3), rng.integers(1, 64, n), 0).astype(np.uint8)
        loads = rng.normal(size=(n, 6))
        fs.set_dirichlet(dmask); fs.set_loads(loads); fs.assemble()
        rg, cg, vg, Fg = fs.export_bsr()
        r0, c0, v0, F0 = oracle.assemble(xyz, tri, quad, mat, dmask, loads)
        assert np.array_equal(rg, r0) and np.array_equal(cg, c0), (case, kind)
        err = np.abs(vg - v0).max() / np.abs(v0).max()
        worst = max(worst, err)
        if err > 1e-12:  # where, and how large are the entries there
            k = int(np.argmax(np.abs(vg - v0).reshape(len(v0), -1).max(axis=1)))
            row = int(np.searchsorted(rg, k, side="right") - 1)
            print("case %d %s: err %.2e at block %d (row %d, col %d), |block| max %.3e, max|K| %.3e, nu %.3f E %.3e t %.3e, row in a quad: %s"
                  % (case, kind, err, k, row, cg[k], np.abs(v0[k]).max(), np.abs(v0).max(), nu, E, t, bool(len(quad)) and bool(np.any(quad == row))), flush=True)
        assert err <= (1e-12 if not os.environ.get("STRESS_KEEP_GOING") else 1e-10), (case, kind, err)
        assert np.array_equal(Fg, F0), (case, kind)
        # the product with the assembled matrix (symmetric storage: diagonal triangles, transposed products through LDS
        # and through HBM) against the oracle's product with the oracle's matrix
        xv = rng.normal(size=6 * n)
        yg = np.asarray(fs.spmv(xv)).ravel()
        y0 = oracle.spmv(r0, c0, v0, xv)
        serr = np.abs(yg - y0).max() / (np.abs(v0).max() * np.abs(xv).max())
        worst_spmv = max(worst_spmv, serr)
        assert serr <= 1e-12, (case, kind, serr)
    fs.close()
    print("case %3d %-9s nodes %6d tri %6d quad %5d flags %d  %-15s ok" % (case, kind, n, len(tri), len(quad), flags, kern), flush=True)
print("all %d cases equal to the oracle; worst relative difference %.2e (matrix), %.2e (product, relative to max|K| max|x|)" % (cases, worst, worst_spmv))
