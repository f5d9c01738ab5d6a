"""The multi-rank driver on real kernels: 2 and 3 processes share the one GPU of the test box and talk
through tests/helpers/fake_rccl (a shared-memory stand-in for librccl, RCCL itself refuses two ranks on
one device).  Everything of the N>1 path except RCCL's own transport runs: row partition, ghost elements,
halo packing and placement, the all-reduce points of CG, the final row gather."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers.product import ROOT, ensure_built

pytestmark = pytest.mark.gpu
FAKE_DIR = os.path.join(ROOT, "tests", "helpers", "fake_rccl")
WORKER = os.path.join(ROOT, "tests", "helpers", "multirank_worker.py")


def run_ranks(world, kind, tmp_path, overlap=True, single_reduction=None, pc=None, async_assembly=False, extra_env=None):
    ensure_built()
    subprocess.check_call(["make", "-C", FAKE_DIR, "-s"])
    env = dict(os.environ, FEMSHELL_RCCL_LIB=os.path.join(FAKE_DIR, "libfake_rccl.so"),
               FEMSHELL_HALO_OVERLAP="1" if overlap else "0")
    if pc is not None:
        env["FEMSHELL_TEST_PC"] = pc
    if extra_env:
        env.update(extra_env)
    if async_assembly:
        env["FEMSHELL_TEST_ASYNC"] = "1"
    if single_reduction is not None:  # default: multi-rank solves use the single-reduction recurrence
        env["FEMSHELL_CG_SINGLE_REDUCTION"] = "1" if single_reduction else "0"
    uid = str(tmp_path / ("uid_%d_%s.npy" % (world, kind)))
    outs = [str(tmp_path / ("out_%d_%s_%d.npz" % (world, kind, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), uid, outs[r], kind], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):  # (a rank that failed first usually explains the others' communication errors: show all)
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, "\n".join("--- rank %d\n%s" % (q, logs[q][-1500:]) for q in range(world)))
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("world,kind", [(2, "panel"), (3, "panel"), (2, "cylinder"), (4, "panel")])
def test_partitioned_solve_on_one_gpu_matches_single_rank(world, kind, tmp_path):
    single = run_ranks(1, kind, tmp_path)[0]
    ranks = run_ranks(world, kind, tmp_path)
    assert single["converged"] == 1
    n = single["u"].shape[0]
    covered = np.zeros(n, dtype=bool)
    for r in ranks:
        assert r["converged"] == 1
        assert abs(int(r["iterations"]) - int(single["iterations"])) <= 3
        assert int(r["iterations"]) == int(ranks[0]["iterations"])  # every rank takes the same decisions
        covered[int(r["begin"]):int(r["end"])] = True
        # every rank holds the full gathered solution (build_solution_vector + broadcast semantics)
        np.testing.assert_array_equal(r["u"], ranks[0]["u"])
        # re-solve on the same context (state left by the first solve must not leak): linear in the loads
        assert r["converged2"] == 1 and abs(int(r["iterations2"]) - int(r["iterations"])) <= 3
        assert np.linalg.norm(r["u2"] - 2.0 * r["u"]) <= 1e-8 * np.linalg.norm(r["u2"])
        # the explicit residual ||b - K u|| / ||b|| is a global quantity: every rank reports the same number and it
        # is the single-rank one up to the rounding of two different CG runs (the reduction in front of the
        # all-reduce once skipped its work on a finished solve and reported stale sums times sqrt(world))
        assert float(r["true_res"]) == float(ranks[0]["true_res"])
        assert 0.2 * float(single["true_res"]) <= float(r["true_res"]) <= 5.0 * float(single["true_res"]), (
            float(r["true_res"]), float(single["true_res"]))
    assert covered.all()
    err = np.linalg.norm(ranks[0]["u"] - single["u"]) / np.linalg.norm(single["u"])
    assert err < 1e-8, err  # two CG runs with different summation order on an ill-conditioned system


@pytest.mark.parametrize("world,kind", [(2, "panel"), (3, "cylinder")])
def test_row_partitioned_assembly_and_solve_against_the_oracle(world, kind, tmp_path, monkeypatch):
    """Parity of the partitioned path with the CPU oracle, not with the single-rank HIP path: every rank's rows of K and F
    (femshell_export_bsr on a row-partitioned context: rows [row_begin, row_end), global column ids) equal the oracle's
    assembly of the whole mesh, the ranks tile the matrix, and the multigrid-preconditioned solve of the partitioned
    context equals the oracle's refined direct solve up to the sensitivity of the solution to the rounding of two FP64
    assemblies (what the single-rank path shows too)."""
    from tests.helpers import oracle
    from tests.helpers.multirank_worker import build_problem

    monkeypatch.setenv("FEMSHELL_TEST_EXPORT", "1")
    ranks = run_ranks(world, kind, tmp_path, pc="amg")
    m, mat = build_problem(kind)
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), m.dirichlet_mask(), m.loads)
    scale = np.abs(v0).max()
    covered = 0
    for r in sorted(ranks, key=lambda q: int(q["begin"])):
        b, e = int(r["begin"]), int(r["end"])
        assert b == covered  # contiguous, ascending, no overlap
        covered = e
        rp = np.asarray(r["k_rowptr"], dtype=np.int64)
        np.testing.assert_array_equal(rp - rp[0], r0[b:e + 1] - r0[b])
        lo, hi = int(r0[b]), int(r0[e])
        np.testing.assert_array_equal(r["k_cols"][:hi - lo], c0[lo:hi])
        assert np.abs(r["k_vals"][:hi - lo] - v0[lo:hi]).max() <= 1e-12 * scale
        np.testing.assert_array_equal(r["k_F"], F0[6 * b:6 * e])
    assert covered == m.n_nodes
    u0 = oracle.refined_solve(r0, c0, v0, F0)
    err = np.linalg.norm(ranks[0]["u"].ravel() - u0) / np.linalg.norm(u0)
    assert all(int(r["converged"]) == 1 for r in ranks)
    assert err < 1e-8, err  # solver term ~1e-13 + kappa x (1e-16 rounding differences of the two assemblies)


@pytest.mark.parametrize("world,order", [(2, "morton"), (3, "rcm")])
def test_renumbered_row_partition_of_a_randomly_numbered_mesh_against_the_oracle(world, order, tmp_path, monkeypatch):
    """FEMSHELL_REORDER on a row-partitioned context (doc/implementation.tex:103-124: libMesh partitions for locality whatever the
    file's numbering is): every rank computes the same permutation and owns a stretch of it.  A 20k-node Delaunay shell in
    the generator's random numbering: the ranks' owned nodes (femshell_owned_nodes) tile the mesh, their rows of K and F equal
    the oracle's assembly in the CALLER's numbering, the gathered solution (caller's numbering on every rank) equals the
    oracle's solve, and the renumbered mesh gets the pipelined assembly kernel."""
    from tests.helpers import oracle
    from tests.helpers.multirank_worker import build_problem

    monkeypatch.setenv("FEMSHELL_TEST_EXPORT", "1")
    ranks = run_ranks(world, "jittered_random", tmp_path, pc="amg", extra_env={"FEMSHELL_REORDER": order})
    m, mat = build_problem("jittered_random")
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, np.zeros((0, 4), np.int32), oracle.material(*mat), m.dirichlet_mask(), m.loads)
    scale = np.abs(v0).max()
    seen = np.zeros(m.n_nodes, dtype=np.int32)
    for r in ranks:
        own = np.asarray(r["own"], dtype=np.int64)
        assert len(own) == int(r["end"]) - int(r["begin"])
        seen[own] += 1
        rp = np.asarray(r["k_rowptr"], dtype=np.int64)
        np.testing.assert_array_equal(np.diff(rp), (r0[own + 1] - r0[own]))
        idx = np.concatenate([np.arange(r0[a], r0[a + 1]) for a in own])  # the oracle's blocks of these rows, row after row
        np.testing.assert_array_equal(r["k_cols"][:len(idx)], c0[idx])
        assert np.abs(r["k_vals"][:len(idx)] - v0[idx]).max() <= 1e-12 * scale
        np.testing.assert_array_equal(r["k_F"].reshape(-1, 6), F0.reshape(-1, 6)[own])
        assert str(r["assembly_kernel"]) == "k_assemble_pipe"
        # a compact patch each: what a rank sends in the first solve (setup of the hierarchy included) stays far below the
        # 180 MB the setup alone moved when a rank's rows were scattered over the shell
        assert int(r["first_solve_bytes"][0]) < 40e6, r["first_solve_bytes"]
        np.testing.assert_array_equal(r["u"], ranks[0]["u"])
        assert int(r["converged"]) == 1
    assert (seen == 1).all()
    u0 = oracle.refined_solve(r0, c0, v0, F0)
    err = np.linalg.norm(ranks[0]["u"].ravel() - u0) / np.linalg.norm(u0)
    assert err < 1e-8, err


@pytest.mark.parametrize("pc", ["amg", None])
def test_solve_from_an_initial_guess_on_a_row_partitioned_context(pc, tmp_path):
    """femshell_set_initial_guess with several ranks: every rank keeps its own rows of the guess (or of its previous solution), the
    ghost entries travel as for any iterate; same answer on every rank, fewer iterations than from zero."""
    ranks = run_ranks(2, "panel", tmp_path, pc=pc, extra_env={"FEMSHELL_TEST_WARM": "1"})
    for r in ranks:
        assert int(r["converged3"]) == 1 and int(r["converged4"]) == 1
        np.testing.assert_array_equal(r["u3"], ranks[0]["u3"])
        np.testing.assert_array_equal(r["u4"], ranks[0]["u4"])
        scale = np.linalg.norm(r["u2"])
        assert np.linalg.norm(r["u3"] - r["u2"]) <= 1e-8 * scale            # the same loads again
        assert np.linalg.norm(r["u4"] - r["u2"]) <= 1e-8 * scale            # from u = u2 / 2: half way there
        assert int(r["iterations4"]) < int(r["iterations2"])
        if pc == "amg":  # (plain-FP64 block-Jacobi CG gains from a guess down to kappa x eps only: tests/test_gpu_amg.py)
            assert int(r["iterations3"]) <= 0.5 * int(r["iterations2"]), (int(r["iterations3"]), int(r["iterations2"]))


def test_multigrid_setup_traffic_of_a_randomly_numbered_mesh_with_and_without_renumbering(tmp_path):
    """What the setup of the row-partitioned hierarchy sends per rank on the 20k-node Delaunay shell of poor element quality in
    random numbering: the rows of Q, P and A P of every node another rank reads.  In the caller's numbering half of all nodes are
    such nodes; after renumbering only those along the cut."""
    (tmp_path / "m").mkdir()
    ranks = run_ranks(2, "delaunay_hard_random", tmp_path / "m", extra_env={"FEMSHELL_REORDER": "morton"})
    for r in ranks:
        assert str(r["assembly_kernel"]) == "k_assemble_pipe"
        assert int(r["setup_bytes"][0]) <= 20e6, r["setup_bytes"]
    print("setup bytes per rank (renumbered):", [int(r["setup_bytes"][0]) for r in ranks])


@pytest.mark.parametrize("world,kind,dist_min", [(2, "panel", 60000), (3, "cylinder", 60000), (4, "panel", 60000),
                                                  (2, "cylinder", 100), (3, "panel", 100), (4, "cylinder", 100),
                                                  (2, "jittered", 60000), (3, "jittered", 100)])
def test_multigrid_on_a_row_partitioned_context(world, kind, dist_min, tmp_path):
    """The hierarchy of a row-partitioned context is row-partitioned itself (csrc/amg_dist.cpp): aggregates never span ranks,
    everything else is the single-rank method, so the iteration count stays within 15 % of the single-rank count and the
    solution is the same.  dist_min = 100: level 1 is split over the ranks as well (the 4M-triangle meshes' 223k-node level)."""
    (tmp_path / "one").mkdir()
    (tmp_path / "many").mkdir()
    (tmp_path / "jac").mkdir()
    env = {"FEMSHELL_AMG_DIST_MIN": str(dist_min), "FEMSHELL_TEST_AMG_EXPORT": "1"}
    single = run_ranks(1, kind, tmp_path / "one", pc="amg")[0]
    ranks = run_ranks(world, kind, tmp_path / "many", pc="amg", extra_env=env)
    jacobi = run_ranks(1, kind, tmp_path / "jac")[0]
    assert single["converged"] == 1 and int(single["levels"]) >= 2
    assert int(single["iterations"]) * 5 < int(jacobi["iterations"])  # what the preconditioner is for
    covered = np.zeros(single["u"].shape[0], dtype=bool)
    for r in ranks:
        assert r["converged"] == 1 and int(r["levels"]) >= 2
        assert int(r["amg_partitioned_levels"]) == (2 if dist_min == 100 else 1)
        assert int(r["iterations"]) <= 1.15 * int(single["iterations"]) + 1, (int(r["iterations"]), int(single["iterations"]))
        assert int(r["iterations"]) == int(ranks[0]["iterations"])
        covered[int(r["begin"]):int(r["end"])] = True
        np.testing.assert_array_equal(r["u"], ranks[0]["u"])
        assert r["converged2"] == 1
        assert np.linalg.norm(r["u2"] - 2.0 * r["u"]) <= 1e-9 * np.linalg.norm(r["u2"])
        assert float(r["true_res"]) == float(ranks[0]["true_res"])
    assert covered.all()
    err = np.linalg.norm(ranks[0]["u"] - single["u"]) / np.linalg.norm(single["u"])
    # both went through the refinement pass with the double-double residual (the thin unstructured shell, E t^3 five orders of
    # magnitude softer in bending than in its plane, leaves 1.7e-10 between two such solves)
    assert err < (5e-10 if kind == "jittered" else 1e-10), err
    errj = np.linalg.norm(ranks[0]["u"] - jacobi["u"]) / np.linalg.norm(jacobi["u"])
    assert errj < 1e-8, errj
    # no whole-K shadow: what a rank holds of the partitioned levels is its share
    part = [float(r["amg_bytes"][0]) for r in ranks]
    assert max(part) <= 1.6 * sum(part) / world and min(part) > 0.0, part


@pytest.mark.parametrize("world,kind,dist_min", [(2, "panel", 100), (3, "cylinder", 60000), (4, "panel", 100)])
def test_row_partitioned_hierarchy_follows_the_restatement(world, kind, dist_min, tmp_path):
    """oracle/amg_oracle.py restates the rank-local aggregation (aggregate_by_rank); tentative prolongator, smoothing and
    Galerkin product are the single-rank ones.  The ranks' rows of every level -- aggregates, P, the level operators -- put
    together equal the restatement's, level by level; so do the iteration count and the residual history; and the solution
    equals the oracle's refined direct solve of the oracle-assembled K up to the sensitivity of two FP64 assemblies."""
    import scipy.sparse as sp
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import amg_oracle
    from tests.helpers import oracle
    from tests.helpers.multirank_worker import build_problem

    env = {"FEMSHELL_AMG_DIST_MIN": str(dist_min), "FEMSHELL_TEST_AMG_EXPORT": "1", "FEMSHELL_TEST_EXPORT": "1"}
    ranks = sorted(run_ranks(world, kind, tmp_path, pc="amg", extra_env=env), key=lambda q: int(q["begin"]))
    m, mat = build_problem(kind)
    n = m.n_nodes

    def bsr(rowptr, cols, vals, nr, nc):
        return sp.bsr_matrix((np.asarray(vals).reshape(-1, 6, 6), np.asarray(cols), np.asarray(rowptr)), shape=(6 * nr, 6 * nc))

    def stack(level, what, nc):
        """the ranks' rows of an operator of a row-partitioned level, one under the other"""
        rp, ci, va = [np.zeros(1, dtype=np.int64)], [], []
        for r in ranks:
            p = np.asarray(r["amg_L%d_%s_rowptr" % (level, what)], dtype=np.int64)
            rp.append(p[1:] + rp[-1][-1])
            ci.append(r["amg_L%d_%s_cols" % (level, what)])
            va.append(r["amg_L%d_%s_vals" % (level, what)].reshape(-1, 6, 6))
        rp = np.concatenate(rp)
        return bsr(rp, np.concatenate(ci), np.concatenate(va), len(rp) - 1, nc)

    # K as the ranks assembled it
    off = np.cumsum([0] + [int(np.asarray(r["k_rowptr"])[-1]) for r in ranks])
    rp = np.concatenate([[0]] + [np.asarray(r["k_rowptr"], dtype=np.int64)[1:] + off[i] for i, r in enumerate(ranks)])
    ci = np.concatenate([r["k_cols"][:int(np.asarray(r["k_rowptr"])[-1])] for r in ranks])
    va = np.concatenate([r["k_vals"][:int(np.asarray(r["k_rowptr"])[-1])].reshape(-1, 6, 6) for r in ranks])
    A = bsr(rp, ci, va, n, n)
    F = np.concatenate([r["k_F"] for r in ranks])
    bounds = [int(r["begin"]) for r in ranks] + [n]
    n_nodes = [int(v) for v in ranks[0]["amg_n_nodes"]]
    levels = amg_oracle.setup(A, m.xyz, m.dirichlet_mask(), lams=[float(v) for v in ranks[0]["amg_lambda"]], coarsest_nodes=60,
                              tri=m.tri, quad=m.quad, bounds=bounds, dist_min=dist_min)
    assert [L.n for L in levels] == n_nodes
    d = int(ranks[0]["amg_partitioned_levels"])
    assert [L.bounds is not None for L in levels][:d + 1] == [True] * d + [False] and d == (2 if dist_min == 100 else 1)
    for li, L in enumerate(levels[:-1]):
        if li < d:
            np.testing.assert_array_equal(np.concatenate([r["amg_L%d_agg" % li] for r in ranks]), L.agg)
            P = stack(li, "P", n_nodes[li + 1])
            if li >= 1:
                Al = stack(li, "A", n_nodes[li])
                assert abs(Al - L.A).max() <= 1e-10 * abs(L.A).max()
        else:  # replicated: every rank holds the whole level
            for r in ranks:
                np.testing.assert_array_equal(r["amg_L%d_agg" % li], L.agg)
            r = ranks[-1]
            P = bsr(r["amg_L%d_P_rowptr" % li], r["amg_L%d_P_cols" % li], r["amg_L%d_P_vals" % li], n_nodes[li], n_nodes[li + 1])
            Al = bsr(r["amg_L%d_A_rowptr" % li], r["amg_L%d_A_cols" % li], r["amg_L%d_A_vals" % li], n_nodes[li], n_nodes[li])
            assert abs(Al - L.A).max() <= 1e-10 * abs(L.A).max()
        assert abs(P - L.P).max() <= 1e-11 * abs(L.P).max()
    u0, hist = amg_oracle.solve(A, F, levels, kcycle=True, rtol=1e-11, max_it=400, refine_passes=1)
    its = int(ranks[0]["iterations"])
    assert abs(len(hist) - its) <= 5, (len(hist), its)
    h = ranks[0]["residual_history"]
    k = min(len(h), len(hist), 20)
    np.testing.assert_allclose(h[:k], hist[:k], rtol=1e-5)
    assert np.linalg.norm(ranks[0]["u"].ravel() - u0) / np.linalg.norm(u0) < 1e-10
    # ... and against the oracle's own assembly and refined direct solve
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), m.dirichlet_mask(), m.loads)
    ud = oracle.refined_solve(r0, c0, v0, F0)
    assert np.linalg.norm(ranks[0]["u"].ravel() - ud) / np.linalg.norm(ud) < 1e-8


@pytest.mark.parametrize("async_assembly", [False, True])
def test_rank_local_failure_is_reported_by_every_rank(tmp_path, async_assembly):
    # a degenerate element exists on the rank that owns its rows only; the others must not walk on into the
    # collectives of the CG loop (they would hang): all ranks agree on the outcome after the assembly -- also when the
    # assembly was enqueued with femshell_assemble_async and the solve is the call that collects its status
    ranks = run_ranks(3, "panel_bad", tmp_path, async_assembly=async_assembly)
    codes = [int(r["code"]) for r in ranks]
    # the same class of error everywhere: FEMSHELL_ERR_MESH (-4) when the assembly flags the element, FEMSHELL_ERR_BREAKDOWN
    # (-5) when its zero block only shows in the block-Jacobi setup
    assert codes[0] in (-4, -5) and all(c == codes[0] for c in codes), codes
    local = [str(r["msg"]) for r in ranks if "other rank" not in str(r["msg"])]
    remote = [str(r["msg"]) for r in ranks if "other rank" in str(r["msg"])]
    assert len(local) >= 1 and len(remote) >= 1 and len(local) + len(remote) == 3, [str(r["msg"]) for r in ranks]


def test_a_rank_that_never_joins_ends_the_others_with_a_message(tmp_path):
    """The first real N > 1 run must fail loudly, not hang: ncclCommInitRank waits for every rank, RCCL has no timeout of its
    own there, and a blocked call cannot be cancelled.  The library watches its blocking phases (csrc/comm.hpp CommWatch): a
    monitor thread ends the process with status 86 and one line naming rank and phase once nothing has moved for
    FEMSHELL_COMM_TIMEOUT seconds.  Three ranks expected, the third never starts: the other two end within a minute."""
    import glob
    import time

    ensure_built()
    subprocess.check_call(["make", "-C", FAKE_DIR, "-s"])
    env = dict(os.environ, FEMSHELL_RCCL_LIB=os.path.join(FAKE_DIR, "libfake_rccl.so"), FEMSHELL_COMM_TIMEOUT="6")
    uid = str(tmp_path / "uid.npy")
    before = set(glob.glob("/dev/shm/fsfake_*"))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), "3", uid, str(tmp_path / ("out_%d.npz" % r)), "panel"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=60)
            logs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in set(glob.glob("/dev/shm/fsfake_*")) - before:  # (the test transport's segment: nobody reached the cleanup)
            os.remove(f)
    assert time.time() - t0 < 60
    for r, p in enumerate(procs):
        assert p.returncode == 86, (p.returncode, logs[r][-1500:])
        assert "[femshell watchdog] rank %d of 3" % r in logs[r] and "ncclCommInitRank" in logs[r], logs[r][-1500:]


def test_halo_overlap_and_single_stream_exchange_agree(tmp_path):
    # the halo exchange beside the interior SpMV (second stream, the default) and the single-stream
    # exchange multiply the same rows by the same numbers; only the order of the p.q partial sums differs
    (tmp_path / "on").mkdir()
    (tmp_path / "off").mkdir()
    a = run_ranks(2, "panel", tmp_path / "on", overlap=True)[0]
    b = run_ranks(2, "panel", tmp_path / "off", overlap=False)[0]
    assert a["converged"] == 1 and b["converged"] == 1
    assert abs(int(a["iterations"]) - int(b["iterations"])) <= 3
    err = np.linalg.norm(a["u"] - b["u"]) / np.linalg.norm(b["u"])
    assert err < 1e-8, err


@pytest.mark.parametrize("world,dist_min", [(2, 100), (3, 100)])
def test_multigrid_halo_exchanges_of_split_coarse_levels_run_beside_the_interior_slices(world, dist_min, tmp_path):
    """Round 5 (csrc/amg_solve.cpp Cycle::overlapped): the products of a row-partitioned level >= 1 -- smoothing products,
    residual increments, the K cycle's own products -- send the ghost entries of their input on the halo stream while the
    slices that read owned columns only are multiplied, as level 0 has done since round 2.  Same rows times the same numbers:
    the solution is the one of the blocking exchange (FEMSHELL_HALO_OVERLAP=0) up to the order of level 0's partial sums.  The counters of the second solve
    (hierarchy reused: the solve alone) show where the exchanges went: with the overlap only the transfers' exchanges (a
    restriction needs the residual, a prolongation the coarse correction of the neighbours' nodes along the cut) stay in the
    main stream's dependency chain."""
    (tmp_path / "on").mkdir()
    (tmp_path / "off").mkdir()
    env = {"FEMSHELL_AMG_DIST_MIN": str(dist_min)}
    on = run_ranks(world, "panel", tmp_path / "on", pc="amg", overlap=True, extra_env=env)
    off = run_ranks(world, "panel", tmp_path / "off", pc="amg", overlap=False, extra_env=env)
    for a, b in zip(on, off):
        assert a["converged"] == 1 and b["converged"] == 1
        assert abs(int(a["iterations"]) - int(b["iterations"])) <= 1 and abs(int(a["iterations2"]) - int(b["iterations2"])) <= 1
        # (the p.q partial sums of level 0 are added in another order when its slices run as two spans: rounding, nothing else)
        assert np.linalg.norm(a["u"] - b["u"]) <= 1e-10 * np.linalg.norm(b["u"])
        assert np.linalg.norm(a["u2"] - b["u2"]) <= 1e-10 * np.linalg.norm(b["u2"])
    second, main, allred, gathers = [float(v) / int(on[0]["iterations2"]) for v in on[0]["comm_solve2"]]
    second_off, main_off, allred_off, gathers_off = [float(v) / int(off[0]["iterations2"]) for v in off[0]["comm_solve2"]]
    assert second_off == 0.0 and main_off > 20.0, off[0]["comm_solve2"]    # (two split levels: every exchange blocks)
    assert abs((second + main) - main_off) < 0.5                            # the same exchanges ...
    assert main <= 12.5 and second >= 2.0 * main, on[0]["comm_solve2"]      # ... of which the products' now travel beside kernels
    assert abs(allred - allred_off) < 0.5 and abs(gathers - gathers_off) < 0.5
    print("per outer iteration on %d ranks: %.1f halo exchanges on the halo stream, %.1f on the main stream, %.1f all-reduces, %.1f row gathers"
          % (world, second, main, allred, gathers))


def test_classic_and_single_reduction_recurrences_agree_across_ranks(tmp_path):
    # multi-rank solves default to the single-reduction recurrence (one all-reduce of three sums per iteration);
    # the classic two-reduction recurrence stays available and must give the same answer
    (tmp_path / "fused").mkdir()
    (tmp_path / "classic").mkdir()
    a = run_ranks(2, "cylinder", tmp_path / "fused", single_reduction=True)[0]
    b = run_ranks(2, "cylinder", tmp_path / "classic", single_reduction=False)[0]
    assert a["converged"] == 1 and b["converged"] == 1
    assert abs(int(a["iterations"]) - int(b["iterations"])) <= 3
    err = np.linalg.norm(a["u"] - b["u"]) / np.linalg.norm(b["u"])
    assert err < 1e-8, err


def test_real_rccl_accepts_the_stream_usage_of_the_cg_driver():
    """Real librccl, one rank talking to itself: grouped send/recv on a second stream behind an event, all-reduces
    on the main stream, alternating on one communicator -- the pattern of cg_driver.cpp with the halo overlap on."""
    probe_dir = os.path.join(ROOT, "tests", "helpers", "rccl_probe")
    subprocess.check_call(["make", "-C", probe_dir, "-s"])
    out = subprocess.run([os.path.join(probe_dir, "two_streams"), "300"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=120)
    assert out.returncode == 0, out.stdout.decode(errors="replace")[-2000:]
    assert b"two_streams ok" in out.stdout


def test_bench_runs_row_partitioned_under_the_launcher_the_driver_uses(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on the one GPU of the test box (both ranks on
    device 0, fake transport): the N > 1 path of bench.py -- gloo control plane, unique id broadcast, barriers, max over
    ranks, the multigrid time-to-solution on the row-partitioned hierarchy -- prints one JSON line from rank 0."""
    import json
    ensure_built()
    subprocess.check_call(["make", "-C", FAKE_DIR, "-s"])
    env = dict(os.environ, FEMSHELL_RCCL_LIB=os.path.join(FAKE_DIR, "libfake_rccl.so"), FEMSHELL_BENCH_SAME_DEVICE="1",
               MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nx", "96",
           "--cg-iters", "5", "--jacobi-probe-iters", "0", "--no-cpu-baseline", "--no-full-parity"]
    env["FEMSHELL_BENCH_DETAIL_DIR"] = str(tmp_path)
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, out.stdout.decode()[-2000:]  # the compact line, and nothing else that looks like one
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["value"] > 0 and d["cg_iters_per_s"] > 0
    assert d["scaling"] == "strong" and d["config"]["parallelism"] == "row-partition x2"
    assert d["time_to_solution_s"] > 0 and d["time_to_solution_iterations"] < 200 and d["roofline"]["frac"] > 0
    with open(tmp_path / "bench_detail.json") as f:  # everything else: the detail record
        tts = json.load(f)["time_to_solution"]
    assert tts["converged"] == 1 and tts["levels"] >= 2 and tts["iterations"] == d["time_to_solution_iterations"]


def test_fp64_fallback_of_the_multigrid_is_taken_by_all_ranks_together(tmp_path):
    """A breakdown of the flexible CG under the single-precision copies of the hierarchy (tests/test_gpu_amg.py has the
    one-rank form) makes femshell_solve rebuild the hierarchy in FP64 and solve again; on a row-partitioned context that
    rebuild is collective, so the ranks must decide alike -- they do, from the all-reduced p.Ap."""
    # (without the patch smoother of round 6, which keeps such a level in FP64 from the start: the test below)
    ranks = run_ranks(2, "delaunay_hard", tmp_path, extra_env={"FEMSHELL_AMG_PATCH_TAU": "0"})
    for r in ranks:
        assert int(r["fallback"]) == 1 and int(r["iterations"]) == 120 and int(r["converged"]) == 0 and bool(r["finite"]), dict(r)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_patch_smoother_converges_on_the_poor_shell_on_one_two_and_three_ranks(world, tmp_path):
    """Round 6, csrc/amg_patch.hpp: the 20,000-point random Delaunay shell numbered along x, on which the
    point-block multigrid needs more than 1000 iterations on one rank or several.  With the cluster blocks -- every rank finds the
    clusters among its own rows, the ranks decide together that the mesh needs them -- the solve converges to rtol 1e-10 within 400
    iterations, without the FP64 rebuild, on 1, 2 and 3 ranks, to the same displacements."""
    (tmp_path / "w").mkdir()
    ranks = run_ranks(world, "delaunay_hard", tmp_path / "w", extra_env={"FEMSHELL_TEST_MAX_IT": "600"})
    for r in ranks:
        assert int(r["converged"]) == 1 and int(r["fallback"]) == 0 and int(r["iterations"]) <= 400, (int(r["iterations"]), int(r["fallback"]))
        assert r["patch"][1] > 1000 and r["patch"][3] == 0  # clusters on every rank, none of them indefinite
        np.testing.assert_array_equal(r["u"], ranks[0]["u"])
    if world > 1:
        (tmp_path / "one").mkdir()
        single = run_ranks(1, "delaunay_hard", tmp_path / "one", extra_env={"FEMSHELL_TEST_MAX_IT": "600"})[0]
        err = np.linalg.norm(ranks[0]["u"] - single["u"]) / np.linalg.norm(single["u"])
        assert err < 1e-5, err  # (two solves of a system whose refined direct solve is good to 1e-7: tests/test_gpu_amg.py)

