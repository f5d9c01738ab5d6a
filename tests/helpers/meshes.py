"""Mesh fixtures of the tests: the Python mirror of the reference's meshGen and XDA/_f readers lives in the
product package (fem-shell_amd/meshgen.py, also used by bench.py); re-exported here under its old name."""
import importlib as _importlib

_m = _importlib.import_module("fem-shell_amd.meshgen")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})


def delaunay_patch(n_pts, seed, strips=True):
    """Random Delaunay triangulation of a curved patch (valences 3..12) whose nodes are numbered strip by strip, so that 32
    consecutive nodes are a compact group: irregular slot widths, chunk counts and element lists per slice, but few enough
    elements per slice for the pipelined assembly kernel (tests of both assembly kernels).  Returns (xyz, tri)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    uv = rng.uniform(0.0, 1.0, size=(n_pts, 2))
    tri = Delaunay(uv).simplices.astype(np.int32)
    p, q, r = uv[tri[:, 0]], uv[tri[:, 1]], uv[tri[:, 2]]
    area = 0.5 * np.abs((q[:, 0] - p[:, 0]) * (r[:, 1] - p[:, 1]) - (q[:, 1] - p[:, 1]) * (r[:, 0] - p[:, 0]))
    tri = tri[area > 0.02 * area.mean()]  # no slivers on the hull
    used = np.unique(tri)
    width = 4.0 / np.sqrt(n_pts)
    key = np.floor(uv[used, 1] / width) * 4.0 + uv[used, 0] if strips else rng.uniform(size=len(used))
    order = used[np.argsort(key, kind="stable")]
    remap = -np.ones(n_pts, dtype=np.int64)
    remap[order] = np.arange(len(order))
    tri = remap[tri].astype(np.int32)
    uv2 = uv[order]
    xyz = np.stack([3.0 * uv2[:, 0], 2.0 * uv2[:, 1], 0.3 * np.sin(3.0 * uv2[:, 0]) * np.cos(2.0 * uv2[:, 1])], axis=1)
    return xyz, tri
