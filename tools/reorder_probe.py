"""Does a space-filling-curve node numbering help the kernels?  The 4M-tri panel with its row-major numbering,
with a Hilbert numbering and with a Morton numbering of the nodes (profiling aid; the library keeps the caller's
numbering).  python tools/reorder_probe.py [nx]"""
import importlib, sys
import numpy as np
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")


def hilbert_index(ix, iy, bits):
    """Hilbert curve index of integer grid points (vectorised xy2d)."""
    ix = ix.astype(np.int64).copy(); iy = iy.astype(np.int64).copy()
    d = np.zeros_like(ix)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((ix & s) > 0).astype(np.int64); ry = ((iy & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        flip = (ry == 0) & (rx == 1)
        ix = np.where(flip, s - 1 - ix, ix); iy = np.where(flip, s - 1 - iy, iy)
        swap = ry == 0
        ix, iy = np.where(swap, iy, ix), np.where(swap, ix, iy)
        s >>= 1
    return d


def morton_index(ix, iy, bits):
    d = np.zeros(len(ix), dtype=np.int64)
    for b in range(bits):
        d |= ((ix >> b) & 1).astype(np.int64) << (2 * b)
        d |= ((iy >> b) & 1).astype(np.int64) << (2 * b + 1)
    return d


def tile_index(ix, iy, tw, th, nxn):
    tiles_per_row = (nxn + tw - 1) // tw
    return ((iy // th) * tiles_per_row + ix // tw) * (tw * th) + (iy % th) * tw + ix % tw


nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1414
m = panel_mesh(nx)
n = m.n_nodes
ix = np.arange(n) % (nx + 1); iy = np.arange(n) // (nx + 1)
orders = {"row-major (as generated)": None,
          "tiles 8x4": np.argsort(tile_index(ix, iy, 8, 4, nx + 1), kind="stable"),
          "tiles 4x8": np.argsort(tile_index(ix, iy, 4, 8, nx + 1), kind="stable"),
          "tiles 16x2": np.argsort(tile_index(ix, iy, 16, 2, nx + 1), kind="stable"),
          "hilbert": np.argsort(hilbert_index(ix, iy, 11), kind="stable"),
          "morton": np.argsort(morton_index(ix, iy, 11), kind="stable")}
for name, order in orders.items():
    if order is None:
        xyz, tri, dmask, loads = m.xyz, m.tri, m.dirichlet_mask(), m.loads
    else:
        inv = np.empty(n, dtype=np.int64); inv[order] = np.arange(n)
        xyz, tri, dmask, loads = m.xyz[order], inv[m.tri].astype(np.int32), m.dirichlet_mask()[order], m.loads[order]
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    fs.set_mesh(xyz, tri); fs.set_dirichlet(dmask); fs.set_loads(loads)
    fs.assemble()
    ms_a, _ = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 10)
    ms_s, _ = fs.time_kernel(pkg.KERNEL_SPMV, 20)
    _, info = fs.solve(rtol=0.0, max_it=200, fetch=False)
    plan_slots = fs.nnz_blocks() if hasattr(fs, "nnz_blocks") else 0
    print("%-26s assemble %.3f ms | spmv %.3f ms | cg %.4f ms/iter" % (name, ms_a, ms_s, 1e3 * info["solve_seconds"] / info["iterations"]), flush=True)
    fs.close()
