"""A Delaunay shell mesh of random points, numbered along a Morton curve, as a binary file for tools/lab/asm_lab:
python tools/lab/write_mesh.py out.bin [points]"""
import sys
import numpy as np
from scipy.spatial import Delaunay
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
rng = np.random.default_rng(5)
uv = rng.uniform(0.0, 1.0, size=(n_pts, 2))
tri = Delaunay(uv).simplices.astype(np.int32)
p, q, r = uv[tri[:, 0]], uv[tri[:, 1]], uv[tri[:, 2]]
area = 0.5 * np.abs((q[:, 0] - p[:, 0]) * (r[:, 1] - p[:, 1]) - (q[:, 1] - p[:, 1]) * (r[:, 0] - p[:, 0]))
tri = tri[area > 1e-3 * area.mean()]
ix = (uv[:, 0] * 65535).astype(np.int64); iy = (uv[:, 1] * 65535).astype(np.int64)
key = np.zeros(n_pts, dtype=np.int64)
for b in range(16):
    key |= ((ix >> b) & 1) << (2 * b)
    key |= ((iy >> b) & 1) << (2 * b + 1)
order = np.argsort(key, kind="stable"); inv = np.empty(n_pts, dtype=np.int64); inv[order] = np.arange(n_pts)
uv = uv[order]; tri = inv[tri].astype(np.int32)
xyz = np.stack([10 * uv[:, 0], 10 * uv[:, 1], 0.5 * np.sin(3 * uv[:, 0]) * np.cos(2 * uv[:, 1])], axis=1)
with open(sys.argv[1], "wb") as f:
    np.array([n_pts, len(tri)], dtype=np.int64).tofile(f)
    xyz.astype(np.float64).tofile(f)
    tri.astype(np.int32).tofile(f)
print("wrote", sys.argv[1], n_pts, "nodes", len(tri), "triangles")
