"""Host side of the multigrid setup (csrc/amg_setup.cpp) against the numpy restatement oracle/amg_oracle.py:
aggregation, tentative prolongator, prolongator smoothing, Galerkin product, coarsest inverse, device layout.
CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import amg_oracle
from tests.helpers import meshes, oracle
from tests.helpers.product import ensure_built


def _binding():
    import importlib

    return importlib.import_module("fem-shell_amd.binding")


def _problem(kind):
    if kind == "panel":
        m = meshes.structured(24, 20, 0, 0, 6, 5, kind="t", ul_lr=True, bcids=(0, 0, 1, -1), factor=300.0, loading=2)
        mat = oracle.material(0.3, 1e7, 0.5)
    elif kind == "roof":
        m = meshes.scordelis_lo(18)
        mat = oracle.material(*m.material)
    elif kind == "quads":
        m = meshes.structured(12, 12, 0, 0, 10, 10, kind="q", bcids=(1, 1, 1, 1), factor=1.0, loading=2)
        mat = oracle.material(0.3, 10.92, 1.0)
    elif kind == "shuffled":
        # the panel with its nodes numbered at random: the aggregation then visits the nodes in breadth-first order of the
        # graph (csrc/amg_setup.cpp: aggregation_order), in the library and in the restatement alike
        m = meshes.structured(48, 40, 0, 0, 6, 5, kind="t", ul_lr=True, bcids=(0, 0, 1, -1), factor=300.0, loading=2)
        perm = np.random.default_rng(5).permutation(m.n_nodes)  # new id of old node i
        inv = np.argsort(perm)
        m.xyz = m.xyz[inv]
        m.loads = m.loads[inv]
        m.tri = perm[m.tri].astype(np.int32)
        mat = oracle.material(0.3, 1e7, 0.5)
        dm_old = meshes.structured(48, 40, 0, 0, 6, 5, kind="t", ul_lr=True, bcids=(0, 0, 1, -1), factor=300.0, loading=2).dirichlet_mask()
        dm = dm_old[inv]
        rp, ci, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, dm, m.loads)
        return m, dm, rp, ci, vals, F
    dm = m.dirichlet_mask()
    rp, ci, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, dm, m.loads)
    return m, dm, rp, ci, vals, F


def _bsr(rowptr, cols, vals, nc):
    n = len(rowptr) - 1
    return sp.bsr_matrix((vals, cols, rowptr), shape=(6 * n, 6 * nc))


def test_aggregation_order_and_node_normals_of_the_restatement():
    m = meshes.structured(24, 20, 0, 0, 6, 5, kind="t", ul_lr=True)
    _, _, rp, ci, _, _ = _problem("panel")
    assert list(amg_oracle.aggregation_order(rp, ci)) == list(range(m.n_nodes))  # a sweeping numbering is kept
    ms, _, rps, cis, _, _ = _problem("shuffled")
    order = list(amg_oracle.aggregation_order(rps, cis))
    assert sorted(order) == list(range(ms.n_nodes)) and order != list(range(ms.n_nodes)) and order[0] == 0
    pos = np.empty(ms.n_nodes, dtype=np.int64)
    pos[order] = np.arange(ms.n_nodes)
    # breadth first: every node but the first has a neighbour that comes before it
    for i in order[1:]:
        assert pos[cis[rps[i]:rps[i + 1]]].min() < pos[i]
    roof = meshes.scordelis_lo(12)
    N = amg_oracle.node_normals(roof.xyz, roof.tri)
    assert np.allclose(np.linalg.norm(N, axis=1), 1.0)
    radial = roof.xyz.copy()
    radial[:, 1] = 0.0  # the roof is a cylinder about the y axis through the origin of x and z (meshes.scordelis_lo)
    cosang = np.abs(np.sum(N * radial, axis=1)) / np.linalg.norm(radial, axis=1)
    assert cosang.min() > 0.99


@pytest.mark.parametrize("kind", ["panel", "roof", "quads", "shuffled"])
def test_one_coarsening_step_equals_the_restatement(kind):
    ensure_built()
    m, dm, rp, ci, vals, F = _problem(kind)
    B = _binding().amg_host_rbm(m.xyz, dm)
    B0 = amg_oracle.rigid_body_modes(m.xyz, dm)
    np.testing.assert_allclose(B, B0, rtol=0, atol=1e-13 * np.abs(B0).max())
    lam = 2.5
    h = _binding().amg_host_coarsen(rp, ci, vals, B, lam)
    A = _bsr(rp, ci, vals, m.n_nodes)
    agg, P, Ac, Bc = amg_oracle.coarsen(A, B0, lam)
    na = int(agg.max()) + 1
    np.testing.assert_array_equal(h["agg"], agg)
    assert len(h["Ac_rowptr"]) == na + 1
    # every node has an aggregate, aggregates are connected neighbourhoods of ~7 nodes on these meshes
    assert h["agg"].min() == 0 and 3.0 < m.n_nodes / na < 12.0
    Pl = _bsr(h["P_rowptr"], h["P_cols"], h["P_vals"], na)
    assert abs(Pl - P).max() <= 1e-12 * abs(P).max()
    Acl = _bsr(h["Ac_rowptr"], h["Ac_cols"], h["Ac_vals"], na)
    assert abs(Acl - Ac).max() <= 1e-12 * abs(Ac).max()
    np.testing.assert_allclose(h["Bc"], Bc, rtol=0, atol=1e-11 * np.abs(Bc).max())
    # the smoothed prolongator reproduces the near-null space where no Dirichlet row interferes: P Bc = B - omega D^-1 A B,
    # and A B = 0 on rows whose whole neighbourhood is unconstrained -- except in the drilling rotation: the
    # reference's drilling block (max/1000 on every node block, SA:1035-1052) is positive definite, it penalises
    # theta_z itself, so a rigid rotation about the normal has a little energy there (flat meshes: row 5 only)
    PB = (Pl @ sp.bsr_matrix((h["Bc"], np.arange(na, dtype=np.int32), np.arange(na + 1, dtype=np.int32)),
                            shape=(6 * na, 6 * na))).toarray().reshape(m.n_nodes, 6, na, 6)
    free = np.ones(m.n_nodes, dtype=bool)
    for a in range(m.n_nodes):
        nb = ci[rp[a]:rp[a + 1]]
        nb2 = np.concatenate([ci[rp[b]:rp[b + 1]] for b in nb])
        free[a] = not dm[nb2].any()
    scale = np.abs(B0).max()
    for a in (np.nonzero(free)[0][:50] if kind != "roof" else []):
        got = PB[a].sum(axis=1)  # sum over aggregates of the block row of P Bc
        assert np.abs(got - B0[a])[:5].max() <= 1e-8 * scale
    # symmetric and positive definite coarse operator
    Ad = Acl.toarray()
    assert np.abs(Ad - Ad.T).max() <= 1e-12 * np.abs(Ad).max()
    assert np.linalg.eigvalsh(0.5 * (Ad + Ad.T)).min() > 0.0


def test_rank_deficient_aggregates_get_inert_coarse_dofs():
    # a clamped strip: aggregates made of fully fixed nodes carry no near-null-space vector at all; their coarse
    # dofs must come out as decoupled unit rows (the level matrix stays SPD)
    ensure_built()
    m = meshes.structured(6, 30, 0, 0, 1, 5, kind="t", ul_lr=True, bcids=(1, 1, 1, 1))
    m.loads[:, 2] = 1.0
    dm = m.dirichlet_mask()
    dm[m.xyz[:, 1] < 1.01] = 0x3F  # clamp a whole region
    mat = oracle.material(0.3, 1e5, 0.1)
    rp, ci, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, dm, m.loads)
    B = _binding().amg_host_rbm(m.xyz, dm)
    h = _binding().amg_host_coarsen(rp, ci, vals, B, 2.5)
    na = len(h["Ac_rowptr"]) - 1
    Ad = _bsr(h["Ac_rowptr"], h["Ac_cols"], h["Ac_vals"], na).toarray()
    inert = np.nonzero(np.abs(h["Bc"]).sum(axis=2).reshape(-1) == 0.0)[0]  # zero rows of the coarse near-null space
    assert len(inert) >= 6
    for k in inert:
        assert Ad[k, k] == 1.0 and np.abs(Ad[k]).sum() == 1.0 and np.abs(Ad[:, k]).sum() == 1.0
    assert np.linalg.eigvalsh(0.5 * (Ad + Ad.T)).min() > 0.0


def test_dense_inverse_of_the_coarsest_operator():
    ensure_built()
    m, dm, rp, ci, vals, F = _problem("panel")
    B = _binding().amg_host_rbm(m.xyz, dm)
    h = _binding().amg_host_coarsen(rp, ci, vals, B, 2.5)
    na = len(h["Ac_rowptr"]) - 1
    inv = _binding().amg_host_dense_inverse(h["Ac_rowptr"].astype(np.int32), h["Ac_cols"], h["Ac_vals"])
    Ad = _bsr(h["Ac_rowptr"], h["Ac_cols"], h["Ac_vals"], na).toarray()
    assert np.abs(inv @ Ad - np.eye(6 * na)).max() < 1e-8
    assert np.abs(inv - inv.T).max() == 0.0


@pytest.mark.parametrize("diag_first", [True, False])
def test_sliced_ell_image_of_a_level_operator(diag_first):
    # the layout k_spmv reads: vals[base*36 + (((k*3 + j/2)*6 + i)*32 + n)*2 + j%2] for slot k of node n of a slice
    ensure_built()
    m, dm, rp, ci, vals, F = _problem("roof")
    sw, sb, cols, ev = _binding().amg_host_pack(rp, ci, vals, diag_first)
    n = m.n_nodes
    assert len(sw) == (n + 31) // 32 and sb[-1] == len(cols)
    x = np.random.default_rng(3).standard_normal((((n + 31) // 32) * 32, 6))
    x[n:] = 0.0
    y = np.zeros_like(x)
    for s in range(len(sw)):
        blk = ev[sb[s] * 36:sb[s + 1] * 36].reshape(sw[s], 3, 6, 32, 2)  # k, jp, i, n, jj
        for k in range(sw[s]):
            c = cols[sb[s] + k * 32: sb[s] + (k + 1) * 32]
            xs = x[c].reshape(32, 3, 2)  # n, jp, jj
            y[32 * s:32 * s + 32] += np.einsum("pind,npd->ni", blk[k], xs)
            if diag_first and k == 0:
                rows = np.arange(32 * s, 32 * s + 32)
                assert np.array_equal(c[rows < n], rows[rows < n])
    y0 = oracle.spmv(rp, ci, vals, x[:n].ravel()).reshape(n, 6)
    assert np.abs(y[:n] - y0).max() <= 1e-13 * np.abs(y0).max()
    assert np.abs(y[n:]).max() == 0.0


def test_symmetric_storage_image_of_a_coarse_operator():
    """Coarse level operators are stored like K: diagonal + upper blocks, and per row the list of stored blocks that act on
    it through their transpose.  Numpy model of the two phases of k_spmv_sym / k_sym_gather on the packed arrays against
    the full product."""
    ensure_built()
    m, dm, rp, ci, vals, F = _problem("panel")
    B = _binding().amg_host_rbm(m.xyz, dm)
    h = _binding().amg_host_coarsen(rp, ci, vals, B, 2.5)
    na = len(h["Ac_rowptr"]) - 1
    Ac = _bsr(h["Ac_rowptr"], h["Ac_cols"], h["Ac_vals"], na)
    S = _binding().amg_host_pack_sym(h["Ac_rowptr"].astype(np.int32), h["Ac_cols"], h["Ac_vals"])
    n_pad = len(S["slice_width"]) * 32
    x = np.zeros((n_pad, 6))
    x[:na] = np.random.default_rng(2).standard_normal((na, 6))
    y = np.zeros_like(x)
    T = np.zeros((len(S["cols"]), 6))
    stored = 0
    for s in range(len(S["slice_width"])):
        w, base = int(S["slice_width"][s]), int(S["slice_base"][s])
        blk = S["vals"][base * 36:(base + w * 32) * 36].reshape(w, 3, 6, 32, 2)  # k, jp, i, n, jj
        for k in range(w):
            for n in range(32):
                a = 32 * s + n
                c = int(S["cols"][base + k * 32 + n])
                Kb = blk[k, :, :, n, :].transpose(1, 0, 2).reshape(6, 6)  # [i][j = 2 jp + jj]
                if k == 0:
                    assert c == min(a, n_pad - 1)
                elif c == a:
                    assert not Kb.any()  # padding slot
                    continue
                else:
                    assert c > a
                    T[base + k * 32 + n] = Kb.T @ x[a]
                    stored += 1
                y[a] += Kb @ x[c]
    for s in range(len(S["in_width"])):
        for k in range(int(S["in_width"][s])):
            for n in range(32):
                e = int(S["in_base"][s]) + k * 32 + n
                if S["in_slots"][e] >= 0:
                    assert S["in_rows"][e] < 32 * s + n
                    y[32 * s + n] += T[S["in_slots"][e]]
    # the packed image is exactly symmetric (upper blocks mirrored): compare with the symmetrised operator
    Ad = Ac.toarray()
    blockmask = np.kron(np.triu(np.ones((na, na)), 1), np.ones((6, 6)))
    diagmask = np.kron(np.eye(na), np.ones((6, 6)))
    Asym = Ad * diagmask + Ad * blockmask + (Ad * blockmask).T
    y0 = Asym @ x[:na].ravel()
    assert np.abs(y[:na].ravel() - y0).max() <= 1e-12 * np.abs(y0).max()
    assert stored == (Ac.tobsr((6, 6)).nnz // 36 - na) // 2
    assert np.abs(Asym - Ad).max() <= 1e-10 * np.abs(Ad).max()  # the Galerkin operator is symmetric to rounding


@pytest.mark.parametrize("kind", ["panel", "cylinder"])
def test_aggregates_of_a_renumbered_graph_are_those_of_the_callers_numbering(kind):
    """femshell_set_mesh may renumber the nodes (FEMSHELL_REORDER_*); it then hands the caller's order to the aggregation
    (csrc/amg_device_setup.cpp), whose passes -- visiting order and the choice among a leftover's neighbours -- follow
    it: the partition into aggregates is the one the caller's numbering gives, whatever the internal numbering is.
    Library and restatement alike."""
    ensure_built()
    if kind == "panel":
        _, _, rp, ci, _, _ = _problem("panel")
    else:
        m = meshes.pinched_cylinder(40, 36)
        rp, ci, _, _ = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*m.material), m.dirichlet_mask(), m.loads)
    n = len(rp) - 1
    agg0, na0 = _binding().amg_host_aggregate(rp, ci)
    agg0_o, na0_o = amg_oracle.aggregate(rp, ci)
    np.testing.assert_array_equal(agg0, agg0_o)
    assert na0 == na0_o
    # internal numbering: new = perm[old]; the graph in internal numbering, columns ascending as the library stores them
    perm = np.random.default_rng(11).permutation(n)
    iperm = np.argsort(perm)  # iperm[new] = old
    G = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    Gp = G[iperm][:, iperm].tocsr()
    Gp.sort_indices()
    rpp, cip = Gp.indptr.astype(np.int32), Gp.indices.astype(np.int32)
    # visiting order = the caller's order expressed in internal ids: caller node 0 first, ... -> perm[0], perm[1], ...
    agg1, na1 = _binding().amg_host_aggregate(rpp, cip, visit=perm)
    agg1_o, na1_o = amg_oracle.aggregate(rpp, cip, visit=perm)
    np.testing.assert_array_equal(agg1, agg1_o)
    assert na1 == na0
    np.testing.assert_array_equal(agg1[perm], agg0)  # same aggregate, same aggregate number, node by node
    # without the order the partition is another one (breadth-first over a scattered numbering)
    agg2, na2 = _binding().amg_host_aggregate(rpp, cip)
    assert not np.array_equal(agg2[perm], agg0)
    with pytest.raises(Exception):
        _binding().amg_host_aggregate(rpp, cip, visit=np.zeros(n, np.int32))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_rank_local_aggregation_of_the_restatement(world):
    """Row-partitioned contexts (csrc/amg_dist.cpp): every rank aggregates the graph of its own rows without the edges that
    leave it, aggregates are numbered rank by rank, everything else is the single-rank method.  CPU side of that contract: the
    restatement's aggregate_by_rank equals the library's host aggregation (femshell_amg_host_aggregate) run rank by rank on
    the rank's sub-graph, no aggregate spans two ranks, and the hierarchy built from such aggregates still solves the system
    in about the iterations of the single-rank hierarchy."""
    ensure_built()
    b = _binding()
    m = meshes.structured(48, 40, 0, 0, 6, 5, kind="t", ul_lr=True, bcids=(0, 0, 1, -1), factor=300.0, loading=2)
    mat = oracle.material(0.3, 1e7, 0.5)
    r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    A = oracle.to_scipy(r, c, v).tobsr((6, 6))
    A.sort_indices()
    n = m.n_nodes
    bounds = amg_oracle.partition_bounds_equal(n, world)
    assert bounds[0] == 0 and bounds[-1] == n and all(x % 32 == 0 for x in bounds[1:-1])
    agg, na, cb = amg_oracle.aggregate_by_rank(A.indptr, A.indices, bounds)
    assert len(cb) == world + 1 and cb[-1] == na
    rank_of_node = np.searchsorted(bounds, np.arange(n), side="right") - 1
    rank_of_agg = np.searchsorted(cb, agg, side="right") - 1
    np.testing.assert_array_equal(rank_of_node, rank_of_agg)  # no aggregate spans two ranks
    for k in range(world):  # the library's host aggregation on the rank's own sub-graph
        b0, b1 = bounds[k], bounds[k + 1]
        sub = A.tocsr()[6 * b0:6 * b1, 6 * b0:6 * b1].tobsr((6, 6))
        sub.sort_indices()
        got, got_na = b.amg_host_aggregate(sub.indptr, sub.indices)
        assert got_na == cb[k + 1] - cb[k]
        np.testing.assert_array_equal(got + cb[k], agg[b0:b1])
    one = amg_oracle.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=60, tri=m.tri)
    u1, h1 = amg_oracle.solve(A, F.ravel(), one, rtol=1e-10, max_it=300, refine_passes=1)
    for dist_min in (60000, 100):
        lv = amg_oracle.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=60, tri=m.tri, bounds=bounds, dist_min=dist_min)
        assert [L.bounds is not None for L in lv][:2] == [True, dist_min == 100]
        u, h = amg_oracle.solve(A, F.ravel(), lv, rtol=1e-10, max_it=300, refine_passes=1)
        assert len(h) <= 1.2 * len(h1) + 2, (world, dist_min, len(h), len(h1))
        assert np.linalg.norm(u - u1) <= 1e-9 * np.linalg.norm(u1)


def test_restatement_from_an_initial_guess_and_with_the_adaptive_pass():
    """oracle/amg_oracle.py solve: (i) x0 -- the first phase solves the correction equation of the guess down to the threshold it runs
    to from zero (csrc/amg_solve.cpp cg_amg, femshell_set_initial_guess): same answer, fewer iterations from a good guess, the
    iterations of a solve from zero from a useless one; (ii) the refinement pass stops on its own error estimate (a fifth of the
    tolerance) instead of at a flat drop of 1e-4: never more iterations, the estimate it leaves stays below the tolerance."""
    m, dm, rp, ci, vals, F = _problem("panel")
    A = _bsr(rp, ci, vals, m.n_nodes)
    levels = amg_oracle.setup(A, m.xyz, dm, coarsest_nodes=60, tri=m.tri)
    b = F.ravel()
    x, h = amg_oracle.solve(A, b, levels, rtol=1e-10, max_it=300, refine_passes=1)
    xf, hf = amg_oracle.solve(A, b, levels, rtol=1e-10, max_it=300, refine_passes=1, adaptive=False)
    assert len(h) <= len(hf)
    assert np.linalg.norm(x - xf) <= 1e-9 * np.linalg.norm(xf)
    xw, hw = amg_oracle.solve(A, b, levels, rtol=1e-10, max_it=300, refine_passes=1, x0=0.96 * x)
    assert len(hw) < len(h) and np.linalg.norm(xw - x) <= 1e-9 * np.linalg.norm(x)
    xs, hs = amg_oracle.solve(A, b, levels, rtol=1e-10, max_it=300, refine_passes=1, x0=x)
    assert len(hs) <= 0.5 * len(h) and np.linalg.norm(xs - x) <= 1e-9 * np.linalg.norm(x)
    xz, hz = amg_oracle.solve(A, b, levels, rtol=1e-10, max_it=300, refine_passes=1, x0=np.zeros_like(b))
    assert abs(len(hz) - len(h)) <= 1 and np.linalg.norm(xz - x) <= 1e-9 * np.linalg.norm(x)


def test_aggregation_looks_at_the_twelve_closest_neighbours_of_a_node(monkeypatch):
    """Rows of more than twelve neighbours (the Galerkin operators of some structured meshes: 17 per node on level 2 of the
    10M-triangle cylinder) are aggregated on the graph of the neighbours a node shares most neighbours with
    (csrc/amg_setup.cpp graph_for_aggregation): library and restatement agree on it, and the aggregates get smaller."""
    ensure_built()
    b = _binding()
    rng = np.random.default_rng(11)
    n = 2500
    xy = rng.uniform(0.0, 1.0, (n, 2))
    order = np.lexsort((xy[:, 0], np.floor(xy[:, 1] * 25)))  # a numbering that sweeps the square in strips
    xy = xy[order]
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(axis=2)
    G = sp.csr_matrix(d2 <= (2.3 / np.sqrt(n)) ** 2)  # about 16 neighbours per node, the node itself included
    G.sort_indices()
    rowptr, cols = G.indptr.astype(np.int32), G.indices.astype(np.int32)
    assert np.diff(rowptr).max() > 13
    fp, fc = amg_oracle.graph_for_aggregation(rowptr, cols, 12)
    F = sp.csr_matrix((np.ones(len(fc)), fc, fp), shape=(n, n))
    assert (F != F.T).nnz == 0 and F.diagonal().all() and F.nnz < G.nnz
    kept = np.diff(fp) - 1
    assert kept.min() >= np.minimum(np.diff(rowptr) - 1, 12).min() and kept.mean() < (np.diff(rowptr) - 1).mean()
    agg_lib = np.asarray(b.amg_host_aggregate(rowptr, cols)[0])
    agg_ref, na_ref = amg_oracle.aggregate(rowptr, cols)
    assert np.array_equal(agg_lib, agg_ref)
    monkeypatch.setenv("FEMSHELL_AMG_AGG_KEEP", "0")
    agg_all = np.asarray(b.amg_host_aggregate(rowptr, cols)[0])
    agg_ref0, na_ref0 = amg_oracle.aggregate(rowptr, cols)
    assert np.array_equal(agg_all, agg_ref0)
    assert agg_lib.max() + 1 > 1.05 * (agg_all.max() + 1)


@pytest.mark.parametrize("threads", ["1", "5"])
def test_chunked_aggregation_equals_the_restatement_whatever_the_thread_count(monkeypatch, threads):
    """Round 6 (an experiment knob, off by default -- the seams cost iterations on structured meshes,
    profiles/r06_chunked_aggregation.txt): with FEMSHELL_AMG_AGG_CHUNK rows per chunk set, graphs of more than one and a half
    chunks are aggregated piece by piece -- consecutive rows, the boundaries of a row partition, no edge across a boundary, aggregates numbered piece by piece --
    on the host's threads (csrc/amg_setup.cpp aggregate_nodes).  The pieces depend on the row count alone: same aggregates on one
    thread and on five, equal to the restatement's (oracle/amg_oracle.py aggregate), with and without a visiting order; no
    aggregate spans two pieces; a graph below the threshold is aggregated in one piece as before."""
    ensure_built()
    b = _binding()
    monkeypatch.setenv("FEMSHELL_HOST_THREADS", threads)
    m = meshes.pinched_cylinder(60, 50)
    rp, ci, _, _ = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*m.material), m.dirichlet_mask(), m.loads)
    n = len(rp) - 1
    whole, na_whole = b.amg_host_aggregate(rp, ci)  # default: one piece
    monkeypatch.setenv("FEMSHELL_AMG_AGG_CHUNK", "0")
    off, na_off = b.amg_host_aggregate(rp, ci)
    np.testing.assert_array_equal(whole, off)
    monkeypatch.setenv("FEMSHELL_AMG_AGG_CHUNK", "700")
    agg, na = b.amg_host_aggregate(rp, ci)
    agg_o, na_o = amg_oracle.aggregate(rp, ci)
    np.testing.assert_array_equal(agg, agg_o)
    assert na == na_o and not np.array_equal(agg, whole)
    nc = (n + 699) // 700
    bounds = amg_oracle.partition_bounds_equal(n, nc)
    piece_of_node = np.searchsorted(bounds, np.arange(n), side="right") - 1
    first = {}
    for i in range(n):  # every aggregate lies in one piece, and the aggregates are numbered piece by piece
        assert first.setdefault(int(agg[i]), int(piece_of_node[i])) == piece_of_node[i]
    assert all(first[a] <= first[a + 1] for a in range(na - 1))
    # close to the aggregates of the whole graph in number (the seams cost a few)
    assert na_whole <= na <= 1.25 * na_whole
    # with a visiting order (the caller's numbering of a renumbered mesh): restricted to each piece
    perm = np.random.default_rng(3).permutation(n)
    iperm = np.argsort(perm)
    G = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    Gp = G[iperm][:, iperm].tocsr()
    Gp.sort_indices()
    rpp, cip = Gp.indptr.astype(np.int32), Gp.indices.astype(np.int32)
    agg1, na1 = b.amg_host_aggregate(rpp, cip, visit=perm)
    agg1_o, na1_o = amg_oracle.aggregate(rpp, cip, visit=perm)
    np.testing.assert_array_equal(agg1, agg1_o)
    assert na1 == na1_o


def _poor_shell(n_pts, seed):
    from tests.test_gpu_parity import delaunay_shell

    xyz, tri = delaunay_shell(n_pts, seed)
    n = len(xyz)
    dmask = np.zeros(n, dtype=np.uint8)
    dmask[xyz[:, 0] < 0.15] = 0x3F
    loads = np.zeros((n, 6))
    loads[:, 2] = 1.0
    rp, ci, vals, F = oracle.assemble(xyz, tri, np.zeros((0, 4), np.int32), oracle.material(0.3, 7.0e4, 0.03), dmask, loads)
    return xyz, tri, dmask, rp, ci, vals, F


def test_patch_smoother_clusters_and_glued_aggregation_follow_the_restatement():
    """Round 6, csrc/amg_patch.hpp: on a random-point Delaunay shell (nodes a hundredth of the mesh width apart) the edges with
    sigma_max(D_i^-1/2 A_ij D_j^-1/2) > tau are united into clusters of bounded size, and the aggregation glues every cluster into one
    node first.  Host side of the library (the device finds the edges with the same arithmetic, amg_patch.hpp patch_sigma2) against
    oracle/amg_oracle.py: same edges' count, same labels, same aggregates; the estimate of sigma against numpy's SVD; a structured
    mesh has no such edge."""
    ensure_built()
    b = _binding()
    xyz, tri, dmask, rp, ci, vals, F = _poor_shell(1200, 4)
    A = oracle.to_scipy(rp, ci, vals).tobsr((6, 6))
    A.sort_indices()
    n = len(xyz)
    Dinv = amg_oracle.block_diag_inverse(A)
    for tau, mx in ((0.8, 8), (0.7, 6), (0.9, 3)):
        labels, nc, edges = b.amg_host_patch_clusters(rp, ci, vals, tau=tau, max_nodes=mx)
        ea, ec, s2 = amg_oracle.patch_edges(A, Dinv, tau)
        lab_o, ptr_o, nodes_o = amg_oracle.patch_clusters(n, ea, ec, s2, mx)
        assert edges == len(ea) and nc == len(ptr_o) - 1 and nc > 50
        np.testing.assert_array_equal(labels, lab_o)
        sizes = np.bincount(labels[labels >= 0])
        assert sizes.min() >= 2 and sizes.max() <= mx
        first = [int(np.flatnonzero(labels == k)[0]) for k in range(nc)]
        assert first == sorted(first)  # clusters numbered by their smallest node
    # the estimate against the singular values: every edge it keeps is rigid by numpy's SVD too (the power steps approach from
    # below), and it misses none that is rigid by a margin
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    up = np.flatnonzero(A.indices > rows)
    Li = np.linalg.inv(np.linalg.cholesky(amg_oracle.block_diag(A)))
    S = np.einsum("eab,ebc,edc->ead", Li[rows[up]], A.data[up], Li[A.indices[up]])
    sv = np.linalg.svd(S, compute_uv=False)[:, 0]
    ea, ec, s2 = amg_oracle.patch_edges(A, Dinv, 0.8)
    kept = set(zip(ea.tolist(), ec.tolist()))
    exact = {(int(rows[e]), int(A.indices[e])): sv[k] for k, e in enumerate(up)}
    assert all(exact[e] > 0.8 * (1 - 1e-9) for e in kept)
    assert all(e in kept for e, s in exact.items() if s > 0.81)
    est = dict(zip(zip(ea.tolist(), ec.tolist()), np.sqrt(s2)))
    assert max(abs(est[e] - exact[e]) / exact[e] for e in kept) < 0.05  # (16 power steps on a non-symmetric T)
    # glued aggregation: library == restatement, the nodes of a cluster share an aggregate, with and without a visiting order
    labels, nc, _ = b.amg_host_patch_clusters(rp, ci, vals, tau=0.8, max_nodes=8)
    agg, na = b.amg_host_aggregate_glued(rp, ci, labels)
    agg_o, na_o = amg_oracle.aggregate_glued(rp, ci, labels.astype(np.int64))
    np.testing.assert_array_equal(agg, agg_o)
    assert na == na_o
    for k in range(nc):
        assert len(set(agg[labels == k].tolist())) == 1
    perm = np.random.default_rng(9).permutation(n)
    agg_v, na_v = b.amg_host_aggregate_glued(rp, ci, labels, visit=perm)
    agg_vo, na_vo = amg_oracle.aggregate_glued(rp, ci, labels.astype(np.int64), visit=perm)
    np.testing.assert_array_equal(agg_v, agg_vo)
    # a structured panel: no rigid edge at all
    _, _, rps, cis, vs, _ = _problem("panel")
    lab_s, nc_s, edges_s = b.amg_host_patch_clusters(rps, cis, vs, tau=0.8, max_nodes=8)
    assert nc_s == 0 and edges_s == 0 and (lab_s == -1).all()


def test_patch_smoother_restatement_converges_where_point_blocks_do_not():
    """The numpy restatement of the method on a 1200-point shell: with the cluster blocks (smoother, spectral bound, smoothing of P,
    glued aggregates) the solve converges in a few hundred iterations; with point blocks alone it does not within 600."""
    xyz, tri, dmask, rp, ci, vals, F = _poor_shell(1200, 4)
    A = oracle.to_scipy(rp, ci, vals).tobsr((6, 6))
    A.sort_indices()
    lv = amg_oracle.setup(A, xyz, dmask, tri=tri, coarsest_nodes=200, patch_tau=0.8, patch_max=8)
    assert lv[0].patch is not None
    u, h = amg_oracle.solve(A, F.ravel(), lv, rtol=1e-10, max_it=600, refine_passes=1)
    assert len(h) < 450 and h[-1] < 1e-6, (len(h), h[-1])
    u0 = oracle.refined_solve(rp, ci, vals, F)
    assert np.linalg.norm(u - u0) <= 1e-6 * np.linalg.norm(u0)
    plain = amg_oracle.setup(A, xyz, dmask, tri=tri, coarsest_nodes=200)
    _, hp = amg_oracle.solve(A, F.ravel(), plain, rtol=1e-10, max_it=600, refine_passes=1)
    assert len(hp) == 600 or len(hp) > 1.5 * len(h), (len(hp), len(h))
