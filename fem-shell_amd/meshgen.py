"""Python mirror of the reference's meshGen and of its mesh / force file readers (tests and bench.py; the C++
twins are host/meshgen_main.cpp and host/mesh_io.cpp): reader for the reference's XDA + "_f" files and
structured generators with the topology of the reference's meshGen, plus the BASELINE.json geometries.

File formats: /root/reference/doc/implementation.tex:76-146.
Generator topology / side-BC numbering / force rule:
/root/reference/src/meshgen/main_all.cpp:144-224, 283-338, 341-387.
Boundary-id semantics: /root/reference/src/fem-shell/fem-shell.cpp:90-120
(ids 0,20 fix u,v,w; ids 1,21 fix all six dofs; id 2 is the coupling interface).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MESH_DIR = os.path.join(GOLDEN, "meshes")

MASK_SS = 0b000111
MASK_CLAMPED = 0b111111


class Mesh:
    def __init__(self, xyz, tri, quad, bcs, loads=None):
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
        self.quad = np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
        self.bcs = list(bcs)  # (element index in file order, side, bc id)
        self.n_nodes = len(self.xyz)
        self.loads = np.zeros((self.n_nodes, 6)) if loads is None else loads
        self.elem_order = None  # file order: list of ("t"|"q", index)

    def element_nodes(self, e):
        """Nodes of element e in file order (tris and quads may be interleaved)."""
        if self.elem_order is None:
            if e < len(self.tri):
                return self.tri[e]
            return self.quad[e - len(self.tri)]
        kind, idx = self.elem_order[e]
        return self.tri[idx] if kind == "t" else self.quad[idx]

    def dirichlet_mask(self):
        """Per-node bit mask; both nodes of each flagged side are constrained."""
        mask = np.zeros(self.n_nodes, dtype=np.uint8)
        for e, side, bid in self.bcs:
            nodes = self.element_nodes(e)
            n = len(nodes)
            a, b = nodes[side], nodes[(side + 1) % n]
            if bid in (0, 20):
                mask[a] |= MASK_SS
                mask[b] |= MASK_SS
            elif bid in (1, 21):
                mask[a] |= MASK_CLAMPED
                mask[b] |= MASK_CLAMPED
        return mask

    def interface_nodes(self, ids=(2, 20, 21)):
        out = set()
        for e, side, bid in self.bcs:
            if bid in ids:
                nodes = self.element_nodes(e)
                out.add(int(nodes[side]))
                out.add(int(nodes[(side + 1) % len(nodes)]))
        return sorted(out)


def _tokens(path):
    with open(path) as f:
        for line in f:
            line = line.split("#")[0]
            for tok in line.split():
                yield tok


def read_xda(path):
    """ASCII XDA as written by libMesh 0.7+/meshGen (TRI3 = type 3, QUAD4 = type 5)."""
    with open(path) as f:
        lines = [ln.split("#")[0].strip() for ln in f]
    assert lines[0].startswith("libMesh"), "not an XDA file"
    n_elem = int(lines[1].split()[0])
    n_nodes = int(lines[2].split()[0])
    pos = 8
    tri, quad, order = [], [], []
    for _ in range(n_elem):
        t = lines[pos].split()
        pos += 1
        if t[0] == "3":
            order.append(("t", len(tri)))
            tri.append([int(v) for v in t[1:4]])
        elif t[0] == "5":
            order.append(("q", len(quad)))
            quad.append([int(v) for v in t[1:5]])
        else:
            raise ValueError("unsupported element type " + t[0])
    xyz = []
    for _ in range(n_nodes):
        xyz.append([float(v) for v in lines[pos].split()[:3]])
        pos += 1
    n_bc = int(lines[pos].split()[0])
    pos += 1
    bcs = []
    for _ in range(n_bc):
        e, s, b = (int(v) for v in lines[pos].split()[:3])
        bcs.append((e, s, b))
        pos += 1
    m = Mesh(np.array(xyz), np.array(tri, dtype=np.int32).reshape(-1, 3),
             np.array(quad, dtype=np.int32).reshape(-1, 4), bcs)
    m.elem_order = order
    return m


def read_forces(path, n_nodes):
    """"_f" file: n, factor, n rows of 6; short files leave the missing rows zero
    (fem-shell.cpp:52-67; meshGen writes n-1 rows, main_all.cpp:352,377)."""
    toks = list(_tokens(path))
    n = int(toks[0])
    factor = float(toks[1])
    vals = [float(t) for t in toks[2:2 + 6 * n]]
    loads = np.zeros((n_nodes, 6))
    flat = np.zeros(6 * n)
    flat[: len(vals)] = vals
    loads[: min(n, n_nodes)] = flat.reshape(n, 6)[: min(n, n_nodes)] * factor
    return loads


def load_example(name):
    """name e.g. 'test_A_uv_t' -> Mesh with loads from the shipped files."""
    m = read_xda(os.path.join(MESH_DIR, name + ".xda"))
    fpath = os.path.join(MESH_DIR, name + "_f")
    if os.path.exists(fpath):
        m.loads = read_forces(fpath, m.n_nodes)
    return m


def structured(nx, ny, x0, y0, x1, y1, kind="t", ul_lr=True, dead_axis="z",
               bcids=(-1, -1, -1, -1), factor=1.0, loading=0, meshgen_quirk=True):
    """Rectangle meshed like the reference's meshGen.

    bcids = (top, bottom, left, right), -1 for none.  loading: 0 none, 1 unit
    load on node n_nodes//2, 2 uniform (nodal force = factor*hx*hy on every node
    but, with meshgen_quirk, the last one).  Loads act along the dead axis.
    """
    n_nodes = (nx + 1) * (ny + 1)
    hx, hy = (x1 - x0) / nx, (y1 - y0) / ny
    jj, ii = np.meshgrid(np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    p = (x0 + ii * hx).ravel()
    s = (y0 + jj * hy).ravel()
    xyz = np.zeros((n_nodes, 3))
    if dead_axis == "z":
        xyz[:, 0], xyz[:, 1] = p, s
    elif dead_axis == "y":
        xyz[:, 0], xyz[:, 2] = p, s
    else:
        xyz[:, 1], xyz[:, 2] = p, s
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    nid = (xx + yy * (nx + 1)).ravel()
    up = nx + 1
    tri = np.zeros((0, 3), dtype=np.int32)
    quad = np.zeros((0, 4), dtype=np.int32)
    if kind == "q":
        quad = np.stack([nid, nid + 1, nid + up + 1, nid + up], axis=1).astype(np.int32)
    else:
        if ul_lr:
            t1 = np.stack([nid, nid + 1, nid + up], axis=1)
            t2 = np.stack([nid + 1, nid + up + 1, nid + up], axis=1)
        else:
            t1 = np.stack([nid, nid + up + 1, nid + 1], axis=1)
            t2 = np.stack([nid + up + 1, nid, nid + up], axis=1)
        tri = np.empty((2 * nx * ny, 3), dtype=np.int32)
        tri[0::2], tri[1::2] = t1, t2
    t_id, b_id, l_id, r_id = bcids
    bcs = []
    for i in range(nx):
        if kind == "t":
            if b_id >= 0:
                bcs.append((2 * i, 0 if ul_lr else 2, b_id))
            if t_id >= 0:
                bcs.append((2 * nx * ny - 2 * i - 1, 1 if ul_lr else 2, t_id))
        else:
            if b_id >= 0:
                bcs.append((i, 0, b_id))
            if t_id >= 0:
                bcs.append((nx * ny - 1 - i, 2, t_id))
    for i in range(ny):
        if kind == "t":
            if ul_lr:
                if l_id >= 0:
                    bcs.append((2 * nx * i, 2, l_id))
                if r_id >= 0:
                    bcs.append((2 * nx * (i + 1) - 1, 0, r_id))
            else:
                if l_id >= 0:
                    bcs.append((2 * nx * i + 1, 1, l_id))
                if r_id >= 0:
                    bcs.append((2 * nx * (i + 1) - 2, 1, r_id))
        else:
            if l_id >= 0:
                bcs.append((nx * i, 3, l_id))
            if r_id >= 0:
                bcs.append((nx * (i + 1) - 1, 1, r_id))
    loads = np.zeros((n_nodes, 6))
    axis = {"x": 0, "y": 1, "z": 2}[dead_axis]
    last = n_nodes - 1 if meshgen_quirk else n_nodes
    if loading == 1:
        loads[n_nodes // 2, axis] = factor
    elif loading == 2:
        loads[:last, axis] = factor * hx * hy
    return Mesh(xyz, tri, quad, bcs, loads)


def map_surface(mesh, fn):
    """Replace coordinates by fn(primary, secondary) -> (x,y,z); for curved shells."""
    p, s = mesh.xyz[:, 0].copy(), mesh.xyz[:, 1].copy()
    mesh.xyz = np.ascontiguousarray(np.stack(fn(p, s), axis=1), dtype=np.float64)
    return mesh


# ---- BASELINE.json configurations 2 and 3 -------------------------------------

def scordelis_lo(n, ul_lr=True):
    """Scordelis-Lo roof: R=25, L=50, 80 degree arc, n x n squares split into triangles; the two curved
    ends carry boundary id 0 (the reference cannot express diaphragm/symmetry constraints, SA:90-116);
    gravity 90 per unit area as area-weighted nodal Fz.  Material: E=4.32e8, nu=0, t=0.25."""
    R, L, arc = 25.0, 50.0, np.deg2rad(80.0)
    m = structured(n, n, 0.0, 0.0, 1.0, 1.0, kind="t", ul_lr=ul_lr, bcids=(0, 0, -1, -1))
    th = (m.xyz[:, 0] - 0.5) * arc  # primary axis -> angle, secondary -> length
    y = m.xyz[:, 1] * L
    m.xyz = np.ascontiguousarray(np.stack([R * np.sin(th), y, R * np.cos(th)], axis=1))
    # lumped gravity: a third of each triangle's area to each of its nodes
    p, q, r = m.xyz[m.tri[:, 0]], m.xyz[m.tri[:, 1]], m.xyz[m.tri[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(q - p, r - p), axis=1)
    w = np.zeros(m.n_nodes)
    for k in range(3):
        np.add.at(w, m.tri[:, k], area / 3.0)
    m.loads = np.zeros((m.n_nodes, 6))
    m.loads[:, 2] = -90.0 * w
    m.material = (0.0, 4.32e8, 0.25)
    return m


def pinched_cylinder(n_theta, n_axial):
    """Pinched cylinder: R=300, L=600, closed in theta (periodic connectivity), both end rings boundary
    id 0, two opposite radial unit loads at mid-length.  Material: E=3e6, nu=0.3, t=3."""
    R, L = 300.0, 600.0
    nt, na = n_theta, n_axial
    n_nodes = nt * (na + 1)
    jj, ii = np.meshgrid(np.arange(na + 1), np.arange(nt), indexing="ij")
    th = 2.0 * np.pi * ii.ravel() / nt
    xyz = np.stack([R * np.cos(th), R * np.sin(th), L * jj.ravel() / na], axis=1)
    a, b = np.meshgrid(np.arange(na), np.arange(nt), indexing="ij")
    n00 = (b + a * nt).ravel()
    n10 = ((b + 1) % nt + a * nt).ravel()
    n01 = n00 + nt
    n11 = n10 + nt
    tri = np.empty((2 * nt * na, 3), dtype=np.int32)
    tri[0::2] = np.stack([n00, n10, n01], axis=1)
    tri[1::2] = np.stack([n10, n11, n01], axis=1)
    bcs = []
    for i in range(nt):
        bcs.append((2 * i, 0, 0))                          # bottom ring: side (n00, n10)
        bcs.append((2 * (nt * (na - 1) + i) + 1, 1, 0))    # top ring: side (n11, n01)
    m = Mesh(xyz, tri, np.zeros((0, 4), dtype=np.int32), bcs)
    mid = (na // 2) * nt
    m.loads[mid, 0] = -1.0                 # at theta = 0, pointing inwards
    m.loads[mid + nt // 2, 0] = 1.0        # at theta = pi
    m.material = (0.3, 3.0e6, 3.0)
    return m
