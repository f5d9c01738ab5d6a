"""CPU tests of the host logic: the C-ABI library loads and exports every declared symbol, the
symbolic plan (block slots, gather lists, slices) reproduces the oracle's assembled matrix, and
the row partition covers the mesh consistently.  No GPU compute call is made."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.helpers import meshes, oracle, sell
from tests.helpers.product import ROOT, ensure_built

pkg = ensure_built()


def declared_symbols():
    names = set()
    for hdr in ("femshell.h", "femshell_plan.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(femshell_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pkg.library_path())
    syms = declared_symbols()
    assert len(syms) >= 24
    for name in syms:
        assert hasattr(lib, name), name


def test_no_device_fails_loudly():
    import shutil
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.FemShellError) as ei:
        pkg.FemShell(0.3, 1e7, 0.5)
    assert ei.value.code == -2
    assert "no CPU path" in str(ei.value)


def plan_matrix(plan, mat, dmask_global):
    n_local = plan["n_pad"] + plan["n_ghost"]
    dm = np.zeros(n_local, dtype=np.uint8)
    for a in range(plan["n_own"]):
        dm[a] = dmask_global[plan["row_begin"] + a]
    for g in range(plan["n_ghost"]):
        dm[plan["n_pad"] + g] = dmask_global[plan["ghost_global"][g]]
    return sell.assemble_from_plan(plan, mat, oracle, dm)


@pytest.mark.parametrize("name,nu,E,t", [("test_A_uv_t", 0.25, 30000.0, 1.0), ("test_E_uvw_t", 0.25, 10000.0, 0.25),
                                         ("test_B_uv_q", 0.25, 30000.0, 1.0)])
def test_gather_lists_reproduce_oracle_matrix(name, nu, E, t):
    m = meshes.load_example(name)
    mat = oracle.material(nu, E, t)
    dmask = m.dirichlet_mask()
    rowptr, colidx, vals, _ = oracle.assemble(m.xyz, m.tri, m.quad, mat, dmask, m.loads)
    plan = pkg.build_plan(m.xyz, m.tri, m.quad)
    assert plan["nnz_blocks"] == len(colidx)
    blocks = plan_matrix(plan, mat, dmask)
    assert len(blocks) == len(colidx)
    scale = np.abs(vals).max()
    for slot, (row, col, blk) in blocks.items():
        q = rowptr[row] + np.searchsorted(colidx[rowptr[row]:rowptr[row + 1]], col)
        assert colidx[q] == col
        assert np.abs(blk - vals[q]).max() <= 1e-13 * scale


def test_diagonal_is_slot_zero_and_columns_ascend():
    m = meshes.structured(7, 5, 0, 0, 7, 5, kind="t", ul_lr=False)
    plan = pkg.build_plan(m.xyz, m.tri, m.quad)
    for s in range(plan["n_slices"]):
        base, w = int(plan["slice_base"][s]), int(plan["slice_width"][s])
        for n in range(32):
            row = s * 32 + n
            if row >= plan["n_own"]:
                continue
            assert plan["cols"][base + n] == row
            prev = -1
            for k in range(1, w):
                slot = base + k * 32 + n
                if plan["pair_ptr"][slot + 1] > plan["pair_ptr"][slot]:
                    assert plan["cols"][slot] > prev
                    prev = plan["cols"][slot]


def test_partition_balances_the_element_incidences_not_the_rows():
    # a mesh whose last part is much finer than the rest: equal row counts would give the last ranks a multiple of the
    # elements of the first ones per row... here the valences differ: quads (4 incidences per interior node) against
    # triangles (6), so equal rows would be unequal work; the partition balances valence + 1 per node
    a = meshes.structured(40, 40, 0, 0, 1, 1, kind="q")
    b = meshes.structured(40, 40, 0, 2, 1, 3, kind="t", ul_lr=True)
    xyz = np.vstack([a.xyz, b.xyz])
    tri = (b.tri + len(a.xyz)).astype(np.int32)
    quad = a.quad.astype(np.int32)
    world = 4
    plans = [pkg.build_plan(xyz, tri, quad, rank=r, world_size=world) for r in range(world)]
    assert plans[0]["row_begin"] == 0 and plans[-1]["row_end"] == len(xyz)
    for p, q in zip(plans[:-1], plans[1:]):
        assert p["row_end"] == q["row_begin"] and p["row_end"] % 32 == 0
    inc = np.bincount(np.concatenate([tri.ravel(), quad.ravel()]), minlength=len(xyz)) + 1
    share = np.array([inc[p["row_begin"]:p["row_end"]].sum() for p in plans], dtype=float)
    assert share.max() / share.mean() < 1.03, share
    rows = np.array([p["row_end"] - p["row_begin"] for p in plans], dtype=float)
    assert rows.max() / rows.min() > 1.15  # not the equal split of the rows


@pytest.mark.parametrize("world", [2, 3, 5])
def test_partition_covers_all_rows_and_halo_lists_match(world):
    m = meshes.structured(23, 17, 0, 0, 1, 1, kind="t", ul_lr=True)
    plans = [pkg.build_plan(m.xyz, m.tri, m.quad, rank=r, world_size=world) for r in range(world)]
    # contiguous cover
    assert plans[0]["row_begin"] == 0 and plans[-1]["row_end"] == m.n_nodes
    for a, b in zip(plans[:-1], plans[1:]):
        assert a["row_end"] == b["row_begin"]
    assert sum(p["nnz_blocks"] for p in plans) == pkg.build_plan(m.xyz, m.tri, m.quad)["nnz_blocks"]
    # what r sends to q is exactly what q expects from r, in the same order
    for r, pr in enumerate(plans):
        for i, q in enumerate(pr["peer_ranks"]):
            send = pr["peer_send_nodes"][pr["peer_send_ptr"][i]:pr["peer_send_ptr"][i + 1]] + pr["row_begin"]
            pq = plans[q]
            j = list(pq["peer_ranks"]).index(r)
            off, cnt = pq["peer_recv_offset"][j], pq["peer_recv_count"][j]
            np.testing.assert_array_equal(send, pq["ghost_global"][off:off + cnt])
    # SpMV order: a permutation of the slices; the interior part reads owned columns only, every
    # boundary slice reads at least one ghost column (those wait for the halo exchange)
    for pr in plans:
        order, ni = pr["spmv_order"], pr["n_interior_slices"]
        np.testing.assert_array_equal(np.sort(order), np.arange(pr["n_slices"]))
        assert ni < pr["n_slices"] and (ni > 0 or pr["n_slices"] <= 2)  # (a rank of two slices may have no interior one)
        for pos, s in enumerate(order):
            cols = pr["cols"][pr["slice_base"][s]:pr["slice_base"][s + 1]]
            assert (cols.max() >= pr["n_pad"]) == (pos >= ni)
    single = pkg.build_plan(m.xyz, m.tri, m.quad)
    assert single["n_interior_slices"] == single["n_slices"]


def test_partitioned_matrix_equals_global_matrix():
    m = meshes.structured(9, 11, 0, 0, 3, 2, kind="t", ul_lr=True, bcids=(0, 0, 1, -1))
    m.xyz[:, 2] = 0.1 * np.sin(m.xyz[:, 0]) * m.xyz[:, 1]  # curved, so frames differ per element
    mat = oracle.material(0.3, 2.0e5, 0.02)
    dmask = m.dirichlet_mask()
    rowptr, colidx, vals, _ = oracle.assemble(m.xyz, m.tri, m.quad, mat, dmask, None)
    K = oracle.to_scipy(rowptr, colidx, vals).toarray()
    scale = np.abs(K).max()
    for r in range(3):
        plan = pkg.build_plan(m.xyz, m.tri, m.quad, rank=r, world_size=3)
        for slot, (row, col, blk) in plan_matrix(plan, mat, dmask).items():
            gr, gc = sell.to_global_id(plan, row), sell.to_global_id(plan, col)
            assert np.abs(K[6 * gr:6 * gr + 6, 6 * gc:6 * gc + 6] - blk).max() <= 1e-13 * scale


def test_invalid_meshes_are_rejected():
    m = meshes.structured(2, 2, 0, 0, 1, 1)
    bad = m.tri.copy()
    bad[0, 0] = 99
    with pytest.raises(pkg.FemShellError):
        pkg.build_plan(m.xyz, bad)
    bad = m.tri.copy()
    bad[1, 1] = bad[1, 0]
    with pytest.raises(pkg.FemShellError):
        pkg.build_plan(m.xyz, bad)
    with pytest.raises(pkg.FemShellError):  # node 8 unused
        pkg.build_plan(m.xyz, m.tri[:-2])


def test_symmetric_storage_balances_unstructured_meshes():
    """Symmetric storage: one block per owned pair, held by the row with fewer blocks (balanced orientation).  On a
    Delaunay mesh (valences 3..12) the slices come out 4-5 slots wide -- 'lower row keeps' gives 6-7, full storage 10-11 --
    and on a structured grid every interior row keeps exactly its three higher neighbours."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(5)
    n = 20000
    pts = rng.random((n, 2))
    key = np.zeros(n, dtype=np.int64)
    ix, iy = (pts[:, 0] * 65535).astype(np.int64), (pts[:, 1] * 65535).astype(np.int64)
    for b in range(16):
        key |= ((ix >> b) & 1) << (2 * b)
        key |= ((iy >> b) & 1) << (2 * b + 1)
    pts = pts[np.argsort(key, kind="stable")]
    tri = Delaunay(pts).simplices.astype(np.int32)
    xyz = np.column_stack([pts, 0.1 * np.sin(3 * pts[:, 0])])
    plan = pkg.build_plan(xyz, tri)
    assert plan["symmetric"] == 1
    rowptr, colidx = oracle.bsr_pattern(n, tri, np.zeros((0, 4), np.int32))
    assert plan["nnz_blocks"] == len(colidx)
    assert plan["stored_blocks"] == (len(colidx) + n) // 2  # diagonal + one block per pair
    assert plan["slice_width"].max() <= 6 and plan["total_slots"] <= 1.25 * plan["stored_blocks"]
    # every off-diagonal stored block (a, c) appears exactly once in the in-list of row c, and nowhere else
    real = plan["pair_ptr"][1:] > plan["pair_ptr"][:-1]
    slots = np.nonzero(real)[0]
    rows = np.empty(len(slots), dtype=np.int64)
    sb = plan["slice_base"]
    sl = np.searchsorted(sb, slots, side="right") - 1
    rows = sl * 32 + (slots - sb[sl]) % 32
    cols = plan["cols"][slots]
    off = slots[rows != cols]
    listed = plan["in_slots"][plan["in_slots"] >= 0]
    assert sorted(listed.tolist()) == sorted(off.tolist())
    pairs = set()
    for a, c in zip(rows[rows != cols].tolist(), cols[rows != cols].tolist()):
        key2 = (min(a, c), max(a, c))
        assert key2 not in pairs  # one block per pair
        pairs.add(key2)
    # structured grid: unchanged layout (width 4, three transposed blocks per interior row)
    m = meshes.structured(40, 40, 0, 0, 1, 1, kind="t", ul_lr=True)
    p2 = pkg.build_plan(m.xyz, m.tri)
    assert p2["slice_width"].max() == 4 and p2["in_width"].max() == 3
    interior = [a for a in range(m.n_nodes) if 0 < a % 41 < 40 and 0 < a // 41 < 40]
    for a in interior[:200]:
        s, nn = a // 32, a % 32
        stored = [int(p2["cols"][p2["slice_base"][s] + k * 32 + nn]) for k in range(int(p2["slice_width"][s]))]
        assert stored == [a, a + 1, a + 40, a + 41]  # right, upper-left (the diagonal), upper


@pytest.mark.parametrize("kind", ["morton", "rcm"])
def test_optional_renumbering_is_a_permutation_that_narrows_the_slices(kind):
    """FEMSHELL_REORDER_MORTON / _RCM (csrc/reorder.cpp): the ordering is a permutation of the caller's ids, and on a
    mesh whose numbering was shuffled the plan built in the new numbering needs far fewer distinct x cache lines per
    32-node slice (what the SpMV gathers) than the plan in the shuffled numbering."""
    m = meshes.structured(48, 40, 0.0, 0.0, 6.0, 5.0, "t")
    rng = np.random.default_rng(11)
    shuffle = rng.permutation(m.n_nodes).astype(np.int32)      # new id of old node
    xyz = np.zeros_like(m.xyz)
    xyz[shuffle] = m.xyz
    tri = shuffle[m.tri].astype(np.int32)
    perm = pkg.reorder_host(kind, xyz, tri)
    assert sorted(perm.tolist()) == list(range(m.n_nodes))
    iperm = np.empty_like(perm)
    iperm[perm] = np.arange(m.n_nodes, dtype=np.int32)

    def lines_per_slice(plan):
        tot = 0
        for s in range(plan["n_slices"]):
            b, e = plan["slice_base"][s], plan["slice_base"][s + 1]
            tot += len(np.unique(plan["cols"][b:e] * 48 // 128))
        return tot / plan["n_slices"]

    before = lines_per_slice(pkg.build_plan(xyz, tri))
    after = lines_per_slice(pkg.build_plan(xyz[perm], iperm[tri].astype(np.int32)))
    assert after < 0.35 * before, (before, after)
    # quads and mixed meshes go through the same graph
    q = meshes.structured(9, 7, 0.0, 0.0, 1.0, 1.0, "q")
    pq = pkg.reorder_host(kind, q.xyz, None, q.quad)
    assert sorted(pq.tolist()) == list(range(q.n_nodes))
    with pytest.raises(pkg.FemShellError):
        pkg.reorder_host(kind, q.xyz, np.array([[0, 1, q.n_nodes]], np.int32))


def test_products_that_stay_inside_a_slice():
    """Symmetric storage: a stored block (a, c) with c in a's own slice hands K_ac^T x_a to row c through LDS.  Every
    in-list entry is served exactly one way -- from LDS (loc_list) or from HBM (gat_slots) -- and the slot and the entry
    agree on the LDS position; positions of a slice are 0..m-1 without gaps."""
    for m in (meshes.structured(40, 40, 0, 0, 1, 1, kind="t", ul_lr=True), meshes.structured(19, 23, 0, 0, 1, 1, kind="q")):
        p = pkg.build_plan(m.xyz, m.tri, m.quad)
        assert p["symmetric"] == 1
        ins, gat, ll, li, sb, ib = p["in_slots"], p["gat_slots"], p["loc_list"], p["loc_index"], p["slice_base"], p["in_base"]
        assert len(gat) == len(ins) == len(ll) and len(li) == p["total_slots"]
        n_local = 0
        for s in range(p["n_slices"]):
            e = np.arange(ib[s], ib[s + 1])
            slot = ins[e]
            inside = (slot >= sb[s]) & (slot < sb[s + 1])
            assert np.all(slot[~inside] == gat[e][~inside]) and np.all(ll[e][~inside] == 255)
            assert np.all(gat[e][inside] == -1)
            pos = ll[e][inside]
            assert sorted(pos.tolist()) == list(range(len(pos)))            # one LDS position each, no gaps
            assert np.all(li[slot[inside]] == pos)                          # the producer writes where the consumer reads
            n_local += len(pos)
            own = li[sb[s]:sb[s + 1]]
            assert np.count_nonzero(own != 255) == len(pos)                 # no slot writes to LDS without a reader
        assert n_local > 0
        # row-major grid: the block towards the previous node of the row is the one that stays in the slice
        if len(m.tri):
            assert 0.25 < n_local / np.count_nonzero(ins >= 0) < 0.40


def _host_side_digest():
    import hashlib
    import importlib

    import scipy.sparse as sp

    b = importlib.import_module("fem-shell_amd.binding")
    h = hashlib.sha256()
    m = meshes.structured(150, 140, 0, 0, 10, 9, kind="t", ul_lr=True, bcids=(0, 0, 1, 1), factor=1.0, loading=2)
    rng = np.random.default_rng(5)
    perm = rng.permutation(m.n_nodes)  # an unstructured numbering: wide slices, repairs of the balanced orientation
    inv = np.empty_like(perm)
    inv[perm] = np.arange(m.n_nodes)
    tri = inv[m.tri].astype(np.int32)
    xyz = m.xyz[perm]
    for rank, world in ((0, 1), (1, 3)):
        for xs, ts in ((m.xyz, m.tri), (xyz, tri)):
            p = b.build_plan(xs, ts, None, rank=rank, world_size=world)
            for k in sorted(p):
                h.update(k.encode())
                h.update(np.ascontiguousarray(p[k]).tobytes())
    # (a mesh whose plan arrays exceed 4 MiB: the allocations that are 2 MiB-aligned and ask for transparent huge pages)
    big = meshes.structured(330, 300, 0, 0, 11, 10, kind="t", ul_lr=True, bcids=(0, 0, 1, 1), factor=1.0, loading=2)
    pb = b.build_plan(big.xyz, big.tri, None)
    for k in sorted(pb):
        h.update(np.ascontiguousarray(pb[k]).tobytes())
    for kind in ("morton", "rcm"):
        h.update(np.ascontiguousarray(b.reorder_host(kind, xyz, tri)).tobytes())
    q = meshes.structured(40, 40, 0, 0, 10, 10, kind="q", bcids=(1, 1, 1, 1), factor=1.0, loading=2)
    pq = b.build_plan(q.xyz, None, q.quad)
    for k in sorted(pq):
        h.update(np.ascontiguousarray(pq[k]).tobytes())
    # a level operator for the host coarsening: block graph of the quad mesh with diagonally dominant SPD blocks
    A = sp.lil_matrix((q.n_nodes, q.n_nodes))
    for e in q.quad:
        for i in e:
            for j in e:
                A[i, j] = 1.0
    A = A.tocsr()
    A.sort_indices()
    vals = np.zeros((A.nnz, 6, 6))
    rows = np.repeat(np.arange(q.n_nodes), np.diff(A.indptr))
    vals[:] = -0.05 * np.eye(6)
    vals[rows == A.indices] = 2.0 * np.eye(6)
    B = b.amg_host_rbm(q.xyz, q.dirichlet_mask())
    c = b.amg_host_coarsen(A.indptr.astype(np.int32), A.indices.astype(np.int32), vals, B, 2.0)
    for k in sorted(c):
        h.update(k.encode())
        h.update(np.ascontiguousarray(c[k]).tobytes())
    return h.hexdigest()


def _plans_only_digest():
    """sha256 over every array of the plans of three meshes (structured, the same under a random numbering, quads) on 1 and
    on 3 ranks: integer arrays and copies of the input coordinates only, nothing a compiler flag could round differently."""
    import hashlib
    import importlib

    b = importlib.import_module("fem-shell_amd.binding")
    h = hashlib.sha256()
    m = meshes.structured(150, 140, 0, 0, 10, 9, kind="t", ul_lr=True, bcids=(0, 0, 1, 1), factor=1.0, loading=2)
    perm = np.random.default_rng(5).permutation(m.n_nodes)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(m.n_nodes)
    q = meshes.structured(40, 40, 0, 0, 10, 10, kind="q", bcids=(1, 1, 1, 1), factor=1.0, loading=2)
    for rank, world in ((0, 1), (1, 3)):
        for xs, ts, qs in ((m.xyz, m.tri, None), (m.xyz[perm], inv[m.tri].astype(np.int32), None), (q.xyz, None, q.quad)):
            p = b.build_plan(xs, ts, qs, rank=rank, world_size=world)
            for k in sorted(p):
                h.update(k.encode())
                h.update(np.ascontiguousarray(p[k]).tobytes())
    return h.hexdigest()


def test_plans_are_the_ones_of_round_three():
    """The symbolic phase was reworked for speed in round 4 (per-thread lists without shared cache lines, arrays that are
    not zero-filled before they are written, a bitmap instead of a sort for the element list of a slice): the plans it
    produces are bit for bit the ones of the round-3 library (tests/golden/plan_digest.txt was written by that build)."""
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "plan_digest.txt")).read().split()[0]
    assert _plans_only_digest() == golden


def test_host_threads_do_not_change_the_plan(monkeypatch):
    """The symbolic phase, the renumbering and the host coarsening run on FEMSHELL_HOST_THREADS threads (plan.cpp,
    reorder.cpp, amg_setup.cpp): every array they produce is the same bit for bit on 1, 3 and 8 threads.  Under
    tools/run_sanitizers.sh this is also the test that drives the threaded paths under ASan / UBSan / TSan."""
    digests = []
    for threads in (1, 3, 8):
        monkeypatch.setenv("FEMSHELL_HOST_THREADS", str(threads))
        digests.append(_host_side_digest())
    assert digests[0] == digests[1] == digests[2], digests
    # ... and the two ways a slice's element list is built (bitmap + rank table for element ids that lie close together,
    # sort + search otherwise) give the same plan
    monkeypatch.setenv("FEMSHELL_PLAN_DENSE_SPAN", "0")
    assert _host_side_digest() == digests[0]
    # ... and ordinary pages instead of the transparent huge pages the large arrays ask for
    monkeypatch.delenv("FEMSHELL_PLAN_DENSE_SPAN")
    monkeypatch.setenv("FEMSHELL_HUGEPAGES", "0")
    assert _host_side_digest() == digests[0]


def _slot_lists_from_items(plan):
    """Per slice: {slot in slice: contributions in chunk order}, lane of every item, the items themselves."""
    items = plan["items"].reshape(-1, 4)
    out = []
    for s in range(plan["n_slices"]):
        i0, i1 = plan["item_ptr"][s], plan["item_ptr"][s + 1]
        slots = {}
        for lane, (x, y, z, w) in enumerate(items[i0:i1]):
            slot, chunk, nch, cnt = int(x & 0xffff), int((x >> 16) & 0xff), int(x >> 24), int((z >> 16) & 0xff)
            if nch == 0:
                assert slot == 0xffff and cnt == 0  # inert
                continue
            prs = [int(y & 0xffff), int(y >> 16), int(z & 0xffff)][:cnt]
            slots.setdefault(slot, {})[chunk] = (lane, nch, prs)
        out.append(slots)
    return out


@pytest.mark.parametrize("mesh", ["panel", "patch", "hub", "quads"])
@pytest.mark.parametrize("symmetric", ["1", "0"])
def test_work_items_of_both_assembly_kernels_carry_the_gather_lists(monkeypatch, mesh, symmetric):
    """The work items (Plan::Item) are the gather lists cut into chunks of three contributions, laid out for the kernel that
    reads them: k_assemble takes rounds of 256 ordered by work; k_assemble_pipe (csrc/plan.cpp pack_items_pipe) rounds of
    192 lanes with the chunks of a slot in consecutive lanes of one wave, and a word per wave (most chunks of a slot, all
    items diagonal).  Either way every slot's chunks, in order, are its gather list."""
    monkeypatch.setenv("FEMSHELL_SYMMETRIC", symmetric)
    quad = None
    if mesh == "panel":
        m = meshes.structured(70, 45, 0, 0, 7, 4.5, kind="t", ul_lr=True)
        xyz, tri = m.xyz, m.tri
    elif mesh == "patch":
        xyz, tri = meshes.delaunay_patch(3000, 7)
    elif mesh == "quads":
        m = meshes.structured(40, 30, 0, 0, 4.0, 3.3, kind="q")
        xyz, tri, quad = m.xyz, None, m.quad
    else:  # a fan of 40 triangles around one node among a regular grid: a slot with 14 chunks, others with one
        m = meshes.structured(20, 20, 0, 0, 2, 2, kind="t", ul_lr=True)
        ang = np.linspace(0.0, 2.0 * np.pi, 41)[:-1]
        hub = len(m.xyz)
        ring = np.stack([3.0 + 0.5 * np.cos(ang), 1.0 + 0.5 * np.sin(ang), np.zeros(40)], axis=1)
        xyz = np.concatenate([m.xyz, [[3.0, 1.0, 0.2]], ring])
        fan = np.array([[hub, hub + 1 + k, hub + 1 + (k + 1) % 40] for k in range(40)], dtype=np.int32)
        tri = np.concatenate([m.tri, fan]).astype(np.int32)
    plans = {}
    for pipe, env in (("1", "2"), ("0", "0")):  # 2: the pipelined layout wherever the kernel can run
        monkeypatch.setenv("FEMSHELL_ASM_PIPE", env)
        plans[pipe] = pkg.build_plan(xyz, tri, quad)
    assert plans["0"]["pipe"] == 0
    assert plans["1"]["pipe"] == 1, plans["1"]["max_slice_elems"]  # (all three meshes are numbered compactly enough)
    # the default takes it where a slice's items fit one round of 192 lanes (full storage: 192 off-diagonal slots per
    # structured slice beside the diagonal ones)
    monkeypatch.delenv("FEMSHELL_ASM_PIPE")
    assert pkg.build_plan(xyz, tri, quad)["pipe"] == (1 if symmetric == "1" else 0)
    if mesh == "patch":  # diagonal slots beyond the first wave: chunks of two, marked for the general routine
        items = plans["1"]["items"].reshape(-1, 4)
        marked = items[(items[:, 2] >> 31) == 1]
        assert len(marked) > 0 and np.all((marked[:, 0] & 0xffff) < 32) and np.all(((marked[:, 2] >> 16) & 0xff) <= 2)
        lanes = np.concatenate([np.arange(a, b) - a for a, b in zip(plans["1"]["item_ptr"][:-1], plans["1"]["item_ptr"][1:])])
        assert np.all(lanes[(items[:, 2] >> 31) == 1] % 192 >= 64)  # never in the first wave of a round
    for pipe, plan in plans.items():
        per_slice = _slot_lists_from_items(plan)
        items = plan["items"].reshape(-1, 4)
        for s, slots in enumerate(per_slice):
            base, width = plan["slice_base"][s], plan["slice_width"][s]
            seen = 0
            for k in range(width):
                for n in range(32):
                    idx = base + k * 32 + n
                    want = list(plan["pairs16"][plan["pair_ptr"][idx]:plan["pair_ptr"][idx + 1]])
                    if not want:
                        assert k * 32 + n not in slots
                        continue
                    chunks = slots[k * 32 + n]
                    nch = len(chunks)
                    assert sorted(chunks) == list(range(nch)) and all(c[1] == nch for c in chunks.values())
                    got = [p for c in range(nch) for p in chunks[c][2]]
                    assert got == [int(v) for v in want]
                    if pipe == "0":
                        assert all(len(chunks[c][2]) == 3 for c in range(nch - 1))  # only the last chunk may be short
                    else:  # (a half-empty last wave of off-diagonal slots is cut into chunks of one or two contributions)
                        assert all(1 <= len(chunks[c][2]) <= 3 for c in range(nch))
                    seen += 1
                    if pipe == "1":  # consecutive lanes of one wave of a round of 192
                        lanes = [chunks[c][0] for c in range(nch)]
                        assert lanes == list(range(lanes[0], lanes[0] + nch))
                        assert (lanes[0] % 192) // 64 == (lanes[-1] % 192) // 64 and lanes[0] // 192 == lanes[-1] // 192
            assert seen == len(slots)
            if pipe == "1":  # the wave words
                i0, i1 = plan["item_ptr"][s], plan["item_ptr"][s + 1]
                for w0 in range(i0, i1, 64):
                    wv = items[w0:min(w0 + 64, i1)]
                    live = wv[(wv[:, 0] >> 24) > 0]
                    most = int((live[:, 0] >> 24).max()) if len(live) else 0
                    all_diag = int(len(live) == 0 or bool(np.all(((live[:, 0] & 0xffff) < 32) & ((live[:, 2] >> 31) == 0))))
                    assert np.all(wv[:, 3] == (most | (all_diag << 8)))
    if mesh == "hub":
        assert max(len(c) for slots in _slot_lists_from_items(plans["1"]) for c in slots.values()) >= 14


@pytest.mark.parametrize("mesh", ["delaunay", "mixed", "folded", "two_ranks"])
def test_node_normals_from_the_gather_lists_are_the_element_walks_bit_for_bit(mesh):
    """Round 6: the multigrid setup reads a node's elements from the gather list of its diagonal slot (ascending local element
    order: the order of the sums of node_normals) instead of letting every host thread walk all elements -- 11 ms of the setup at
    4M triangles, threads x elements in general.  Same array, bit for bit: unstructured valences, triangles and quadrilaterals in
    one mesh, a folded plate (normals that nearly cancel), the owned rows of a two-rank partition."""
    rank, world = 0, 1
    quad = None
    if mesh == "delaunay":
        xyz, tri = meshes.delaunay_patch(5000, 11)
    elif mesh == "mixed":
        m = meshes.structured(24, 18, 0, 0, 3, 2, kind="q", bcids=(1, 1, 1, 1), factor=1.0, loading=2)
        xyz = m.xyz.copy()
        xyz[:, 2] = 0.2 * np.sin(2.0 * xyz[:, 0]) * np.cos(xyz[:, 1])
        q = m.quad
        half = len(q) // 2  # the first half of the squares stays quadrilateral, the rest is cut into triangles
        quad = q[:half]
        tri = np.concatenate([q[half:, [0, 1, 2]], q[half:, [0, 2, 3]]]).astype(np.int32)
    elif mesh == "folded":
        m = meshes.structured(20, 20, 0, 0, 2, 2, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=1.0, loading=2)
        xyz, tri = m.xyz.copy(), m.tri
        xyz[:, 2] = np.abs(xyz[:, 0] - 1.0) * 5.0
    else:
        xyz, tri = meshes.delaunay_patch(4000, 3)
        rank, world = 1, 2
    a = pkg.plan_node_normals(xyz, tri, quad, from_gather_lists=False, rank=rank, world_size=world)
    b = pkg.plan_node_normals(xyz, tri, quad, from_gather_lists=True, rank=rank, world_size=world)
    assert a.shape == b.shape and len(a) > 100
    np.testing.assert_array_equal(a, b)
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0)
