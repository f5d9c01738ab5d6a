"""Parity at BASELINE.json's full sizes (configs[2] pinched cylinder and configs[3] flat panel, 1414 x 1414 squares =
3,998,792 tri3, 12M dofs) -- VERDICT r2 item 1.

(a) the matrix the HIP path assembles against the oracle's assembly of the same mesh, all 14M blocks (the oracle needs
    8 s for it on one core);
(b) the solver term of the multigrid solve: a direct solve of 12M dofs is out of the checker's reach, so the solution
    is manufactured (tests/helpers/manufactured.py): u* smooth, b = K u* in double-double on the device, and the solve of
    K u = b has to come back to u* -- reference: equation_systems.solve() fem-shell.cpp:138, the Test-G family of
    doc/validation.tex:518;
(c) the error estimate femshell_solve_info returns against the true error of the same solve.
"""
import numpy as np
import pytest

from tests.helpers import fullsize
from tests.helpers.product import ensure_built, pkg

pytestmark = pytest.mark.gpu

N_FULL = 1414


@pytest.fixture(scope="module", params=["panel", "cylinder"])
def context(request):
    ensure_built()
    kind = request.param
    m, mat = fullsize.workload(kind, N_FULL)
    assert len(m.tri) == 3998792
    fs = pkg.FemShell(*mat, device=0)
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    yield kind, m, mat, fs
    fs.close()


def test_full_size_matrix_equals_oracle(context):
    kind, m, mat, fs = context
    out = fullsize.matrix_parity(fs, m, mat, products=True, kind=kind)
    assert out["same_pattern"], out
    assert out["blocks"] > 14000000
    assert out["F_bitwise_equal"], out
    assert out["max_entry_diff_over_max_entry"] <= 1e-12, out   # measured: 2e-15
    assert out["frobenius_rel_diff"] <= 1e-13, out
    # the device's products against the ORACLE's matrix at this size (the manufactured right-hand sides below are products of
    # the device itself): femshell_spmv against fso_bsr_spmv, femshell_residual (double-double) against longdouble row sums
    pr = out["products_vs_oracle_matrix"]
    assert pr["spmv_rel_diff"] <= 1e-12, pr
    assert pr["residual_dd_sampled_scalar_rows"] >= 200000, pr
    # against the oracle's blocks: what two correct FP64 assemblies differ by (2e-15 of the largest entry) times u
    assert pr["residual_dd_vs_oracle_blocks_max_err_over_largest_row_scale"] <= 1e-15, pr
    assert pr["residual_dd_vs_oracle_blocks_max_err_over_row_scale"] <= 2e-14, pr
    # against the blocks the device holds, by independent arithmetic (longdouble): one rounding of the double-double sum;
    # plain FP64 row sums of the same blocks are several times worse
    assert pr["residual_dd_vs_exported_blocks_max_err_over_row_scale"] <= 1.5e-16, pr
    assert pr["plain_fp64_row_sums_max_err_over_row_scale"] > 1.5e-16, pr


def test_headline_load_case_against_the_reference_held_plate_answer():
    """The 4M-triangle panel IS the plate of the thesis' tests D and G (10 x 10, t 0.5, E 1e7, nu 0.3, q 300:
    doc/validation.tex:283-295): w at the centre against Timoshenko's series value the thesis quotes (alpha = 0.00406 ->
    0.1064045, validation.tex:270), the thesis' own Tri-3 figure at 64 x 64 (0.106413, validation.tex:518) and the full
    Navier series -- an answer at the headline size that neither the solver's error estimate nor a manufactured right-hand
    side supplies.  What it shows (profiles/r05_headline_load_case_vs_navier.txt): the discretisation error falls with h^2
    (-5.0e-4, -1.2e-4, -3.1e-5, -8.9e-6 at 64^2 ... 512^2) and would be 1e-6 at 1414^2 -- where w_c is 1.2e-4 off instead,
    and moves by as much when the SAME matrix is assembled with other roundings (the mesh translated, renumbered, split along
    the other diagonal: -2e-5 ... -1.3e-4).  That is kappa(K) x the rounding of K's own FP64 entries -- the sensitivity term of
    DESIGN section 2, grown with h^-4 to 1e-4 at 4M triangles -- and no solver can remove it (the solver term of these solves
    is 1e-11).  The reference's assembly rounds the same entries."""
    ensure_built()
    navier = fullsize.navier_centre_deflection(300.0, 10.0, 1e7, 0.3, 0.5)
    assert abs(navier - 0.106466) < 2e-6                  # alpha = 0.00406235
    dev = {}
    for n in (64, 128, 256, 512, N_FULL):
        r = fullsize.panel_centre_deflection(n)
        assert r["converged"] == 1, r["iterations"]
        dev[n] = (r["w_centre"] - navier) / navier
    assert abs(dev[64] * navier + navier - 0.106413) < 5e-7     # the thesis' six digits for Tri-3 at 64 x 64
    # second-order convergence while the discretisation error dominates
    for coarse, fine in ((64, 128), (128, 256), (256, 512)):
        assert dev[coarse] < 0 and 3.3 < dev[coarse] / dev[fine] < 4.5, dev
    # the headline size: within 1e-3 of the thesis' analytic reference (three-digit alpha), within 3e-4 of the series ...
    w_full = navier * (1.0 + dev[N_FULL])
    assert abs(w_full - 0.1064045) < 1e-3 * 0.1064045, dev
    assert abs(dev[N_FULL]) < 3e-4, dev
    # ... and what is left is the rounding of K's entries times its condition number: the same plate translated by (3, 7, 0)
    # -- the same matrix in exact arithmetic -- gives another fifth digit
    moved = fullsize.panel_centre_deflection(N_FULL, shift=(3.0, 7.0, 0.0))
    assert moved["converged"] == 1 and moved["error_estimate"] < 1e-9
    change = abs(moved["w_centre"] - w_full) / navier
    assert 1e-7 < change < 3e-4, (change, dev)
    assert abs(dev[N_FULL]) < 20.0 * change + 1e-5, (change, dev)


def test_full_size_manufactured_solution_with_the_spectrum_of_the_load_case(context):
    """A manufactured solve whose u* is the CONVERGED SOLUTION OF THE LOAD CASE (uniform pressure on the panel, the two pinch
    loads on the cylinder) instead of a smooth analytic field: b = K u* has the spectrum of the real right-hand side (the
    smooth field's ||b|| is six decades above the pressure load's), and the solve has to come back to u*."""
    kind, m, mat, fs = context
    fs.set_loads(m.loads)
    fs.assemble()
    fs.set_preconditioner("amg", refine_passes=1)
    u_load, info = fs.solve(rtol=1e-10, max_it=3000)
    assert info["converged"] == 1
    out = fullsize.manufactured_solve(fs, m, kind, rtol=1e-10, passes=(1,), u_star=u_load)
    r1 = out["runs"][1]
    assert r1["converged"] == 1 and r1["refine_passes_done"] >= 1
    assert r1["rel_err_vs_manufactured"] < 1e-10, out
    fs.set_loads(m.loads)


def test_full_size_manufactured_solution(context):
    kind, m, mat, fs = context
    out = fullsize.manufactured_solve(fs, m, kind, rtol=1e-10, passes=(0, 1))
    r0, r1 = out["runs"][0], out["runs"][1]
    assert r0["converged"] == 1 and r1["converged"] == 1, out
    # the right-hand side was rounded to double: the reference moved by less than 1e-12 of itself
    assert out["rounding_of_b"]["delta_over_u_star"] < 1e-12, out
    # one refinement pass: solver term below 1e-10 (north star); without it the FP64 recurrence leaves 4e-9 ... 5e-8
    assert r1["refine_passes_done"] == 1
    assert r1["rel_err_vs_manufactured"] < 1e-10, out
    assert r0["rel_err_vs_manufactured"] < 1e-6, out
    assert r1["rel_err_vs_manufactured"] < 0.05 * r0["rel_err_vs_manufactured"], out
    # (c) the estimate of the info struct: ||e|| / ||x|| of the pass is the error of the iterate before it -- the one the
    # first phase leaves at 100 rtol -- within 20 %
    rf = out["runs"]["first_phase"]
    assert abs(r1["refine_correction_rel"] / rf["rel_err_vs_manufactured"] - 1.0) < 0.2, out
    # ... and estimate x drop bounds what the pass left within a factor of ten either way
    assert 0.1 * r1["rel_err_vs_manufactured"] <= max(r1["error_estimate"], 1e-15), out
    assert r1["iterations"] < 400, out


def test_full_size_properties(context):
    """Size-independent properties on the same contexts: symmetry, null space of the unconstrained operator is not
    testable with constraints in place, so: y.Kx = x.Ky, linearity of the solve in the loads."""
    kind, m, mat, fs = context
    fs.set_loads(m.loads)
    fs.assemble()
    rng = np.random.default_rng(7)
    x, y = rng.standard_normal(6 * m.n_nodes), rng.standard_normal(6 * m.n_nodes)
    Kx, Ky = fs.spmv(x), fs.spmv(y)
    assert abs(y @ Kx - x @ Ky) <= 1e-12 * (np.linalg.norm(y) * np.linalg.norm(Kx))
    fs.set_preconditioner("amg")
    u1, i1 = fs.solve(rtol=1e-10, max_it=1000)
    fs.set_loads(2.5 * m.loads)
    u2, i2 = fs.solve(rtol=1e-10, max_it=1000)
    assert i1["converged"] == 1 and i2["converged"] == 1
    assert np.linalg.norm(u2 - 2.5 * u1) <= 1e-9 * np.linalg.norm(u2)
    assert 0.0 <= i1["error_estimate"] < 1e-9 and i1["refine_passes_done"] >= 1


def test_full_size_coarsest_inverse_on_the_matrix_cores(context, monkeypatch):
    """BASELINE.json's north star asks for the matrix cores on the dense panels of the preconditioner: the inverse of the
    coarsest operator (7386 dofs on the panel, 58 block sweeps on v_mfma_f64_16x16x4_f64, csrc/amg_dense.hip) at its
    production size against numpy: A . A^-1 = I to 1e-9 for the operator the hierarchy hands it (exported, like the
    inverse; the operators themselves are held to the restatement level by level in tests/test_gpu_amg.py)."""
    import scipy.sparse as sp

    kind, m, mat, fs = context
    monkeypatch.setenv("FEMSHELL_AMG_DENSE_F32", "0")  # (the cycle's default keeps the inverse in single precision; the FP64 one is what 1e-9 is asked of)
    fs.set_loads(m.loads)
    fs.assemble()  # (a new hierarchy with the setting above)
    fs.set_preconditioner("amg")
    u, info = fs.solve(rtol=1e-8, max_it=400)
    assert info["converged"] == 1
    lv = fs.amg_levels()
    st = fs.amg_dense_stats()
    ex = fs.amg_export(len(lv) - 1)
    n = 6 * lv[-1]["n_nodes"]
    assert st["n"] == n and n > 6000 and st["dropped_directions"] == 0
    A = sp.bsr_matrix((ex["A_vals"], ex["A_cols"], ex["A_rowptr"]), shape=(n, n)).tocsr()
    inv = ex["coarse_inverse"]
    assert inv.shape == (n, n)
    assert np.abs(inv - inv.T).max() <= 1e-12 * np.abs(inv).max()
    defect = np.abs(A @ inv - np.eye(n)).max()
    assert defect <= 1e-9, defect
    assert st["mfma_flops_issued"] / (st["ms"] * 1e-3) > 15e12  # measured: 25 TFLOP/s of the 78.6 peak
    # the default: the same inverse rounded to single precision
    monkeypatch.delenv("FEMSHELL_AMG_DENSE_F32")
    fs.assemble()
    fs.set_preconditioner("amg")
    u32, info32 = fs.solve(rtol=1e-8, max_it=400)
    inv32 = fs.amg_export(len(lv) - 1)["coarse_inverse"]
    assert info32["converged"] == 1 and abs(info32["iterations"] - info["iterations"]) <= 4
    assert 0.0 < np.abs(inv32 - inv).max() <= 2e-7 * np.abs(inv).max()
    fs.assemble()


def test_a_mesh_size_that_draws_the_dense_tiling(monkeypatch):
    """Four structured sizes in ten end, on level 1, in the tiling whose Galerkin operator has 17 neighbours per node on level 2
    (profiles/r05_aggregation_lottery.txt): the 1300 x 1300 pinched cylinder is one of them, next to the north-star size.  With
    the aggregation on the graph of the twelve closest neighbours (csrc/amg_setup.cpp graph_for_aggregation) levels 2 and 3 are
    coarsened 15 : 1 as on the other sizes and the solve takes 91 iterations; on the whole graph 20 : 1 and 103."""
    ensure_built()
    m, mat = fullsize.workload("cylinder", 1300)
    fs = pkg.FemShell(*mat, device=0)
    try:
        fs.set_mesh(m.xyz, m.tri)
        fs.set_dirichlet(m.dirichlet_mask())
        fs.set_loads(m.loads)
        fs.assemble()
        fs.set_preconditioner("amg")
        u, info = fs.solve(rtol=1e-10, max_it=400, fetch=False)
        nodes = [lv["n_nodes"] for lv in fs.amg_levels()]
        assert info["converged"] == 1 and info["error_estimate"] <= 1e-10
        assert nodes[2] / nodes[3] < 16.5, nodes
        monkeypatch.setenv("FEMSHELL_AMG_AGG_KEEP", "0")
        fs.assemble()  # (a new hierarchy with the setting above)
        fs.set_preconditioner("amg")
        u0, info0 = fs.solve(rtol=1e-10, max_it=400, fetch=False)
        nodes0 = [lv["n_nodes"] for lv in fs.amg_levels()]
        assert info0["converged"] == 1
        assert nodes0[:3] == nodes[:3] and nodes0[2] / nodes0[3] > 18.5, (nodes, nodes0)
        assert info["iterations"] <= 96 and info["iterations"] + 6 <= info0["iterations"], (info["iterations"], info0["iterations"])
    finally:
        fs.close()


def test_full_size_setup_builds_its_patterns_in_hbm_and_keeps_the_hosts_hierarchy(context, monkeypatch):
    """Round 6: at BASELINE's size the three coarsening steps above 5,000 nodes build their patterns in HBM
    (csrc/amg_symbolic.hip; the host runs the greedy passes on K's own ELL pattern), and the hierarchy is the one the host's lists
    give -- level sizes, spectral bounds, operator complexity, iteration count and the residual history bit for bit (in half the
    setup's time on a quiet box: 0.075 against 0.15 s)."""
    kind, m, mat, fs = context
    fs.set_loads(m.loads)
    out = {}
    for where in ("host", "device"):
        monkeypatch.setenv("FEMSHELL_AMG_SYMBOLIC", where)
        fs.set_dirichlet(m.dirichlet_mask())  # (the constraint set handed over again: K is assembled anew and the hierarchy rebuilt)
        fs.set_preconditioner("amg")
        u, info = fs.solve(rtol=1e-10, max_it=400, fetch=False)
        assert info["converged"] == 1 and info["pc_setup_seconds"] > 0.0, info
        out[where] = (info, fs.amg_levels(), fs.amg_symbolic_info(), fs.residual_history().copy())
    assert out["device"][2]["in_hbm"] >= 3 and out["device"][2]["host_after_overflow"] == 0, out["device"][2]
    assert out["host"][2]["in_hbm"] == 0 and out["host"][2]["host_by_rule"] >= 3, out["host"][2]
    assert out["device"][1] == out["host"][1]
    assert out["device"][0]["iterations"] == out["host"][0]["iterations"]
    assert out["device"][0]["operator_complexity"] == out["host"][0]["operator_complexity"]
    np.testing.assert_array_equal(out["device"][3], out["host"][3])
    # (no assertion on the two setup times: the driver's allocator adds 1 to 70 ms to either by the state of the box --
    #  profiles/r06_setup_on_device.txt has the A/B)
    print("setup seconds: patterns in HBM %.3f, host lists %.3f" % (out["device"][0]["pc_setup_seconds"], out["host"][0]["pc_setup_seconds"]))
