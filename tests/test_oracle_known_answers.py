"""Pins the CPU oracle against the reference's own known answers.

The reference has no unit tests; its validation is the thesis tables obtained by
running the shipped example meshes (run_examples.sh:35-48).  Each test runs the
same mesh + parameters through the oracle (assembly + sparse direct solve) and
compares with the printed thesis value to its printed precision.  47 reference-held
scalars in all: Test A (4), B (4), C (1 + 4), D (1 + 6), F (1 + 24), G (2); the unshipped
meshes of C, D, F and G-tri come from the meshGen twin.
"""
import numpy as np
import pytest

from tests.helpers import meshes, oracle


def solve_example(name, nu, E, t):
    m = meshes.load_example(name)
    mat = oracle.material(nu, E, t)
    rowptr, colidx, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u = oracle.direct_solve(rowptr, colidx, vals, F).reshape(-1, 6)
    return m, u


def test_A_tri_cantilever_inplane():
    # doc/validation.tex:62-65
    _, u = solve_example("test_A_uv_t", 0.25, 30000.0, 1.0)
    assert u[22, 0] == pytest.approx(-0.0255988, abs=6e-8)
    assert u[22, 1] == pytest.approx(0.0629549, abs=6e-8)
    assert u[26, 0] == pytest.approx(-0.0342621, abs=6e-8)
    assert u[26, 1] == pytest.approx(0.1944070, abs=6e-7)


def test_B_quad_cantilever_inplane():
    # doc/validation.tex:133-136
    _, u = solve_example("test_B_uv_q", 0.25, 30000.0, 1.0)
    assert u[22, 0] == pytest.approx(-0.0427728, abs=6e-8)
    assert u[22, 1] == pytest.approx(0.1012620, abs=6e-7)
    assert u[26, 0] == pytest.approx(-0.0570728, abs=6e-8)
    assert u[26, 1] == pytest.approx(0.3160560, abs=6e-7)


def test_C_tri_plate_centre_load():
    # doc/validation.tex:200
    _, u = solve_example("test_C_w_tA16", 0.3, 10.92, 1.0)
    assert u[144, 2] == pytest.approx(1.15169, abs=6e-6)


def test_D_quad_plate_uniform():
    # doc/validation.tex:289
    _, u = solve_example("test_D_w_q_uni16", 0.3, 1e7, 0.5)
    assert u[144, 2] == pytest.approx(0.106454, abs=6e-7)


def test_F_quad_plate_ss_uniform_32():
    # doc/validation.tex:474
    _, u = solve_example("test_F_032_ss_uni", 0.3, 1.7472e7, 0.01)
    assert u[544, 2] == pytest.approx(12.9640e-6, abs=6e-11)


def test_G_quad_64():
    # doc/validation.tex:518
    _, u = solve_example("test_G_mpi_64_q", 0.3, 1e7, 0.5)
    assert u[2112, 2] == pytest.approx(0.106465, abs=6e-7)


def test_G_tri_64():
    # doc/validation.tex:518 (Tri-3 value of the same problem: 64x64 squares split ul_lr,
    # simply supported, uniform load 300 -> nodal 300*h^2; generated with the
    # meshGen-equivalent generator because the thesis' tri mesh is not shipped)
    m = meshes.structured(64, 64, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0),
                          factor=300.0, loading=2)
    mat = oracle.material(0.3, 1e7, 0.5)
    rowptr, colidx, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u = oracle.direct_solve(rowptr, colidx, vals, F).reshape(-1, 6)
    assert u[65 * 32 + 32, 2] == pytest.approx(0.106413, abs=6e-7)


def _generated(n, nx_len, ny_len, kind, bc, load, factor, nu, E, t, ul_lr=True, ny=None):
    """Mesh from the meshGen twin (tests/helpers/meshes.structured follows src/meshgen/main_all.cpp: topology :163-224,
    side-BC numbering :283-338, uniform load = factor*hx*hy on every node but the last :341-387), oracle assembly,
    direct solve; returns the deflection of the centre node."""
    m = meshes.structured(n, ny or n, 0, 0, nx_len, ny_len, kind=kind, ul_lr=ul_lr, bcids=(bc, bc, bc, bc),
                          factor=factor, loading=load)
    # the thesis ran files written by meshGen, which prints coordinates and forces with the stream's default six
    # significant digits (main_all.cpp:226-339, 341-387): 300*0.625^2 = 117.1875 went in as 117.188
    m.xyz = np.array([[float("%.6g" % v) for v in row] for row in m.xyz])
    m.loads = np.array([[float("%.6g" % v) for v in row] for row in m.loads])
    mat = oracle.material(nu, E, t)
    rowptr, colidx, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u = oracle.direct_solve(rowptr, colidx, vals, F).reshape(-1, 6)
    return u[m.n_nodes // 2, 2]


def _digits(value_text):
    """half a unit of the last printed digit of a thesis number"""
    frac = value_text.split(".")[1] if "." in value_text else ""
    return 0.6 * 10.0 ** (-len(frac))


@pytest.mark.parametrize("n,ul_lr,text", [(4, True, "1.06723"), (4, False, "1.06723"), (16, True, "1.15169"), (16, False, "1.15169")])
def test_C_tri_plate_both_orientations_and_subdivisions(n, ul_lr, text):
    # doc/validation.tex:196-201 (Test C table: 4x4 and 16x16, both diagonal orientations; Tri-3, centre load 1)
    w = _generated(n, 10, 10, "t", 0, 1, 1.0, 0.3, 10.92, 1.0, ul_lr=ul_lr)
    assert w == pytest.approx(float(text), abs=_digits(text))


@pytest.mark.parametrize("n,load,text", [(4, 2, "0.106032"), (8, 2, "0.106405"), (16, 2, "0.106454"),
                                         (4, 1, "0.332677"), (8, 1, "0.312851"), (16, 1, "0.306664")])
def test_D_quad_plate_uniform_and_concentrated(n, load, text):
    # doc/validation.tex:287-293 (Test D table: Quad-4, simply supported, uniform 300 / concentrated 30000)
    w = _generated(n, 10, 10, "q", 0, load, 300.0 if load == 2 else 30000.0, 0.3, 1e7, 0.5)
    assert w == pytest.approx(float(text), abs=_digits(text))


_TEST_F = {  # doc/validation.tex:470-494, values in 1e-6; (boundary id, loading) -> n = 2, 4, 8, 16, 32, 64
    (0, 2): ["14.4005", "12.6269", "12.8565", "12.9431", "12.9640", "12.9691"],   # SPL, uniform 1e-4
    (1, 2): ["3.82366", "2.45355", "2.60137", "2.60384", "2.60414", "2.60420"],   # CLA, uniform
    (0, 1): ["11.5204", "17.3048", "18.1158", "17.4961", "17.1495", "17.0215"],   # SPL, concentrated 4e-4
    (1, 1): ["3.05893", "6.06564", "7.78902", "7.66573", "7.40552", "7.29681"],   # CLA, concentrated
}


@pytest.mark.parametrize("bc,load", list(_TEST_F))
def test_F_quad_convergence_series(bc, load):
    # Test F: 10 x 2 plate, t = 0.01, E = 1.7472e7, nu = 0.3, Quad-4, n x n subdivisions.  Pins the clamped
    # (boundary id 1: all six dofs) constraint path and point loads on quads, which no shipped mesh exercises.
    for n, text in zip((2, 4, 8, 16, 32, 64), _TEST_F[(bc, load)]):
        w = _generated(n, 10, 2, "q", bc, load, 1e-4 if load == 2 else 4e-4, 0.3, 1.7472e7, 0.01)
        assert w * 1e6 == pytest.approx(float(text), abs=_digits(text)), (bc, load, n)


def test_A_orientation_variants_of_the_thesis_are_not_reproducible():
    # doc/validation.tex:66-73 lists Test A with all diagonals one way; the meshes are not shipped, and the
    # meshGen-style 8x2 cantilever with either orientation lands within 2 % of the mixed mesh, not at the
    # thesis' 5-43 % -- like Test E these rows cannot serve as pins; kept as a record
    for ul_lr, thesis in ((True, -0.0243863), (False, -0.0235617)):
        m = meshes.structured(8, 2, 0, 0, 48, 12, kind="t", ul_lr=ul_lr, bcids=(-1, -1, 0, -1))
        m.loads[8, 1] = m.loads[26, 1] = 40.0 / 6.0
        m.loads[17, 1] = 160.0 / 6.0
        mat = oracle.material(0.25, 30000.0, 1.0)
        rowptr, colidx, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
        u = oracle.direct_solve(rowptr, colidx, vals, F).reshape(-1, 6)
        assert abs(u[22, 0] / -0.0255988 - 1.0) < 0.02 and abs(u[22, 0] / thesis - 1.0) > 0.04


@pytest.mark.parametrize("nx,ny,wc", [(16, 32, 1.044156), (32, 16, 1.1234919), (20, 8, 1.0945778)])
def test_non_isosceles_as_coded_Y(nx, ny, wc):
    # centre deflection of a simply supported 10x10
    # Specht plate with centre load on NON-isosceles triangles, computed in the
    # survey with the reference element code as coded (SA:586 active).  Not a
    # thesis number; kept as a regression pin for the as-coded Y(2,1).
    m = meshes.structured(nx, ny, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0),
                          factor=1.0, loading=1)
    mat = oracle.material(0.3, 10.92, 1.0)
    rowptr, colidx, vals, F = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u = oracle.direct_solve(rowptr, colidx, vals, F).reshape(-1, 6)
    centre = (ny // 2) * (nx + 1) + nx // 2
    assert u[centre, 2] == pytest.approx(wc, rel=2e-6)


def test_tabulated_specht_curvatures_equal_the_polynomial_derivation():
    # the assembly evaluates the Specht curvature matrix from per-Gauss-point tables (filled once from the polynomial
    # derivation of thesis shellelements.tex:1023-1039, 1107-1111); rebuilding the polynomials for every element and
    # Gauss point must give the same element matrices
    rng = np.random.default_rng(11)
    mat = oracle.material(0.27, 3.1e5, 0.07)
    try:
        for _ in range(200):
            X = rng.standard_normal((3, 3)) * rng.uniform(0.1, 5.0)
            oracle.set_specht_polynomial(False)
            k_tab = oracle.element_tri3(X, mat)
            oracle.set_specht_polynomial(True)
            k_pol = oracle.element_tri3(X, mat)
            assert np.abs(k_tab - k_pol).max() <= 2e-13 * np.abs(k_pol).max()
    finally:
        oracle.set_specht_polynomial(False)


def test_committed_goldens_are_what_the_oracle_computes():
    """tests/golden/*.npz are oracle outputs (tools/gen_golden_elements.py, tools/gen_golden_solutions.py):
    a change of the oracle that moves them must be deliberate."""
    g = np.load(meshes.GOLDEN + "/tri3_elements.npz")
    mat = oracle.material(float(g["nu"]), float(g["E"]), float(g["t"]))
    for e, c in enumerate(g["tri"]):
        ke = oracle.element_tri3(g["xyz"][c], mat)
        assert np.abs(ke - g["Ke"][e]).max() <= 1e-13 * np.abs(ke).max()
    g = np.load(meshes.GOLDEN + "/quad4_elements.npz")
    mat = oracle.material(float(g["nu"]), float(g["E"]), float(g["t"]))
    for e, c in enumerate(g["quad"]):
        ke, parts = oracle.element_quad4(g["xyz"][c], mat, want_parts=True)
        assert np.abs(ke - g["Ke"][e]).max() <= 1e-13 * np.abs(ke).max()
        assert np.abs(parts["Ke_m"] - g["Ke_m"][e]).max() <= 1e-13 * np.abs(parts["Ke_m"]).max()
        assert np.abs(parts["Ke_p"] - g["Ke_p"][e]).max() <= 1e-13 * np.abs(parts["Ke_p"]).max()
        # element matrices are symmetric with exactly six zero modes
        w = np.linalg.eigvalsh(0.5 * (ke + ke.T))
        assert (np.abs(w) < 1e-9 * w.max()).sum() == 6 and w.min() > -1e-9 * w.max()
    sols = np.load(meshes.GOLDEN + "/example_solutions.npz")
    for name in ("test_A_uv_t", "test_B_uv_q", "test_C_w_tA16", "test_E_uvw_t"):
        nu, E, t = sols[name + "_params"]
        m = meshes.load_example(name)
        r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(nu, E, t), m.dirichlet_mask(), m.loads)
        u = oracle.direct_solve(r, c, v, F).reshape(-1, 6)
        assert np.linalg.norm(u - sols[name]) <= 1e-9 * np.linalg.norm(sols[name])
