"""Full-size (BASELINE.json's 4M-triangle configurations) checks shared by tests/test_gpu_fullsize.py and bench.py:
the assembled matrix against the oracle's, and the solver term of the multigrid solve against a manufactured solution.
Test infrastructure only."""
import time

import numpy as np

from tests.helpers import manufactured


def workload(kind, n):
    from tests.helpers import meshes

    if kind == "panel":
        m = meshes.structured(n, n, 0.0, 0.0, 10.0, 10.0, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
        return m, (0.3, 1e7, 0.5)
    if kind == "cylinder":
        m = meshes.pinched_cylinder(n, n)
        return m, m.material
    raise ValueError(kind)


def matrix_parity(fs, m, mat, chunk=1 << 20, products=False, kind=None):
    """HIP-assembled K (femshell_export_bsr) against the oracle's assembly of the same mesh: same pattern, largest entry
    difference relative to the largest entry, and F bitwise.  Compared in chunks of blocks so that the temporaries stay
    small beside the two 4 GB value arrays of a 4M-triangle mesh."""
    from tests.helpers import oracle

    t0 = time.perf_counter()
    fs.assemble()
    rg, cg, vg, Fg = fs.export_bsr()
    t_export = time.perf_counter() - t0
    t0 = time.perf_counter()
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), m.dirichlet_mask(), m.loads)
    t_oracle = time.perf_counter() - t0
    out = {"blocks": int(len(c0)), "same_pattern": bool(np.array_equal(rg, r0) and np.array_equal(cg, c0)),
           "F_bitwise_equal": bool(np.array_equal(Fg, F0)), "export_seconds": t_export, "oracle_assembly_seconds": t_oracle}
    if not out["same_pattern"]:
        return out
    scale, worst, sq_d, sq_v = 0.0, 0.0, 0.0, 0.0
    for lo in range(0, len(c0), chunk):
        a, b = vg[lo:lo + chunk], v0[lo:lo + chunk]
        d = a - b
        scale = max(scale, float(np.abs(b).max()))
        worst = max(worst, float(np.abs(d).max()))
        sq_d += float(np.einsum("ijk,ijk->", d, d))
        sq_v += float(np.einsum("ijk,ijk->", b, b))
    out["max_entry_diff_over_max_entry"] = worst / scale
    out["frobenius_rel_diff"] = float(np.sqrt(sq_d / sq_v))
    if products:
        out["products_vs_oracle_matrix"] = products_against_the_oracle_matrix(fs, m, kind, r0, c0, v0, vg)
    return out


def products_against_the_oracle_matrix(fs, m, kind, r0, c0, v0, vg, sampled_nodes=33334, seed=20251004):
    """The device's PRODUCTS at full size against the ORACLE's matrix (the manufactured right-hand sides of
    tests/helpers/manufactured.py are products of the device itself, so something independent has to look at them):
    (1) femshell_spmv (k_spmv_sym + its second phase, FP64) of a pseudo-random vector against fso_bsr_spmv of the oracle-assembled
        K: relative 2-norm difference;
    (2) femshell_residual (k_residual_dd_sym: double-double products and row sums) of the smooth manufactured field u* with zero
        loads, -K u*, against numpy longdouble (64-bit significand) row sums over the oracle's blocks on a random sample of node
        rows (6 scalar rows each), scaled by sum_j |K_ij| |u_j| of the row -- the quantity rounding errors are proportional to.
        Two references: the ORACLE's blocks (the difference then contains the 2e-15 by which two correct FP64 assemblies differ,
        times u) and the blocks the device exported (vg: the same matrix, evaluated by independent arithmetic -- the double-double
        sum rounded once must agree to 1.2e-16 of the row's scale; plain FP64 row sums of the same blocks are given for scale)."""
    from tests.helpers import manufactured, oracle

    n = m.n_nodes
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, 6 * n)
    mask = m.dirichlet_mask()
    y_dev = fs.spmv(x)
    y_or = oracle.spmv(r0, c0, v0, x)
    out = {"spmv_rel_diff": float(np.linalg.norm(y_dev - y_or) / np.linalg.norm(y_or)),
           "spmv_max_diff_over_max": float(np.abs(y_dev - y_or).max() / np.abs(y_or).max())}
    u_star = manufactured.smooth_field(m, kind)
    loads_before = None
    fs.set_loads(np.zeros((n, 6)))
    r_dev = fs.residual(u_star).reshape(n, 6)  # 0 - K u*
    nodes = np.sort(rng.choice(n, size=min(sampled_nodes, n), replace=False))
    worst, worst_fp64, worst_own, norm_scale, norm_diff = 0.0, 0.0, 0.0, 0.0, 0.0
    u = u_star.reshape(n, 6)
    for lo in range(0, len(nodes), 4096):
        rows = nodes[lo:lo + 4096]
        cnt = (r0[rows + 1] - r0[rows]).astype(np.int64)
        idx = np.concatenate([np.arange(r0[a], r0[a + 1]) for a in rows])
        owner = np.repeat(np.arange(len(rows)), cnt)
        blocks = v0[idx].astype(np.longdouble)                 # (nb, 6, 6)
        uc = u[c0[idx]].astype(np.longdouble)                  # (nb, 6)
        prod = np.einsum("bij,bj->bi", blocks, uc)
        absprod = np.einsum("bij,bj->bi", np.abs(blocks), np.abs(uc))
        acc = np.zeros((len(rows), 6), dtype=np.longdouble)
        scale = np.zeros((len(rows), 6), dtype=np.longdouble)
        np.add.at(acc, owner, prod)
        np.add.at(scale, owner, absprod)
        ref = -acc
        d = np.abs(r_dev[rows].astype(np.longdouble) - ref)
        ok = scale > 0
        worst = max(worst, float((d[ok] / scale[ok]).max()))
        norm_scale, norm_diff = max(norm_scale, float(scale.max())), max(norm_diff, float(d.max()))
        own = np.zeros((len(rows), 6), dtype=np.longdouble)    # the blocks the device itself holds (exported)
        np.add.at(own, owner, np.einsum("bij,bj->bi", vg[idx].astype(np.longdouble), uc))
        worst_own = max(worst_own, float((np.abs(r_dev[rows].astype(np.longdouble) + own)[ok] / scale[ok]).max()))
        fp64 = -np.einsum("bij,bj->bi", v0[idx], u[c0[idx]])   # the same sum in plain FP64, for scale: what double-double buys
        acc64 = np.zeros((len(rows), 6))
        np.add.at(acc64, owner, fp64)
        worst_fp64 = max(worst_fp64, float((np.abs(acc64.astype(np.longdouble) - ref)[ok] / scale[ok]).max()))
    out["residual_dd_sampled_scalar_rows"] = int(6 * len(nodes))
    out["residual_dd_vs_oracle_blocks_max_err_over_row_scale"] = worst
    out["residual_dd_vs_oracle_blocks_max_err_over_largest_row_scale"] = norm_diff / norm_scale
    out["residual_dd_vs_exported_blocks_max_err_over_row_scale"] = worst_own
    out["plain_fp64_row_sums_max_err_over_row_scale"] = worst_fp64
    del loads_before
    return out


def navier_centre_deflection(q, a, E, nu, t, terms=399):
    """w at the centre of a simply supported square plate under uniform pressure q (Navier's double series, Timoshenko &
    Woinowsky-Krieger art. 30; the thesis quotes alpha = 0.00406 of it, doc/validation.tex:270): 16 q a^4 / (pi^6 D) x
    sum over odd m, n of (-1)^((m+n)/2 - 1) / (m n (m^2 + n^2)^2)."""
    D = E * t ** 3 / (12.0 * (1.0 - nu * nu))
    odd = np.arange(1, terms + 1, 2, dtype=np.float64)
    sgn = np.where(((odd - 1) // 2) % 2 == 0, 1.0, -1.0)
    M, N = np.meshgrid(odd, odd, indexing="ij")
    S = float(np.sum(np.outer(sgn, sgn) / (M * N * (M * M + N * N) ** 2)))
    return 16.0 * q * a ** 4 / (np.pi ** 6 * D) * S


def panel_centre_deflection(n, rtol=1e-10, fs=None, shift=(0.0, 0.0, 0.0)):
    """w at the centre node of the BASELINE panel (10 x 10, t 0.5, E 1e7, nu 0.3, pressure 300: the plate of the thesis' tests D
    and G, doc/validation.tex:283-295, 518) on n x n squares, solved by the multigrid-preconditioned CG with one refinement pass."""
    import importlib

    pkg = importlib.import_module("fem-shell_amd")
    m, mat = workload("panel", n)
    own = fs is None
    if own:
        fs = pkg.FemShell(*mat, device=0)
        # (shift: a rigid translation of the mesh -- the same stiffness matrix in exact arithmetic, other roundings in FP64)
        fs.set_mesh(m.xyz + np.asarray(shift, dtype=np.float64)[None, :], m.tri)
        fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.assemble()
    fs.set_preconditioner("amg", refine_passes=1)
    u, info = fs.solve(rtol=rtol, max_it=3000)
    centre = (n // 2) * (n + 1) + n // 2
    assert abs(m.xyz[centre, 0] - 5.0) < 1e-12 and abs(m.xyz[centre, 1] - 5.0) < 1e-12
    out = {"n": n, "w_centre": float(u[centre, 2]), "iterations": info["iterations"], "converged": info["converged"],
           "error_estimate": info["error_estimate"], "u": u}
    if own:
        fs.close()
    return out


def manufactured_solve(fs, m, kind, rtol=1e-10, passes=(1,), max_it=3000, with_rounding_correction=True, u_star=None):
    """Solver term at full size: ||u - u_ref|| / ||u_ref|| of multigrid solves of K u = fl(K u*) for the refinement
    pass counts in `passes`; u_ref = u* + delta (tests/helpers/manufactured.py).  The context's loads are replaced."""
    n = m.n_nodes
    # (u_star given: the manufactured solution is that field -- e.g. the converged solution of the real load case, whose
    #  right-hand side has the spectrum of the load case instead of that of a smooth analytic field)
    if u_star is None:
        u_star = manufactured.smooth_field(m, kind)
    u_star = np.ascontiguousarray(u_star, dtype=np.float64).reshape(n, 6)
    fs.assemble()
    b = manufactured.rhs_of(fs, u_star)
    out = {"u_star_norm": float(np.linalg.norm(u_star)), "b_norm": float(np.linalg.norm(b)), "rtol": rtol, "runs": {}}
    delta = np.zeros_like(u_star)
    if with_rounding_correction:
        fs.set_loads(b)
        fs.set_preconditioner("amg", refine_passes=1)
        delta, rho = manufactured.rounding_correction(fs, u_star)
        out["rounding_of_b"] = {"residual_of_u_star_over_b": rho / out["b_norm"],
                                "delta_over_u_star": float(np.linalg.norm(delta) / out["u_star_norm"])}
    u_ref = u_star + delta
    nrm = float(np.linalg.norm(u_ref))
    fs.set_loads(b)
    runs = [(int(p), int(p), rtol) for p in passes]
    if 1 in [int(p) for p in passes]:
        # the iterate a solve with a refinement pass has BEFORE its pass: its first phase stops at 100 rtol (csrc/amg_solve.cpp)
        runs.append(("first_phase", 0, 100.0 * rtol))
    for key, p, tol in runs:
        fs.set_preconditioner("amg", refine_passes=p)
        u, info = fs.solve(rtol=tol, max_it=max_it)
        out["runs"][key] = {
            "iterations": info["iterations"], "converged": info["converged"], "solve_seconds": info["solve_seconds"],
            "rel_err_vs_manufactured": float(np.linalg.norm(u - u_ref) / nrm),
            "rel_err_vs_u_star_alone": float(np.linalg.norm(u - u_star) / out["u_star_norm"]),
            "max_err_over_max_u": float(np.abs(u - u_ref).max() / np.abs(u_ref).max()),
            "true_rel_residual_double_double": info["true_rel_residual"],
            "refine_passes_done": info["refine_passes_done"], "refine_correction_rel": info["refine_correction_rel"],
            "refine_residual_reduction": info["refine_residual_reduction"], "error_estimate": info["error_estimate"]}
    return out


def coupled_flap_full_size(coupled_tool, meshgen, workdir, config, steps=3, nx=500, nz=1000):
    """BASELINE configs[4] at its own size: the flap 0.1 x 1 in the x-z plane (dead axis y), nx x nz squares = 1M tri3 at
    500 x 1000, E = 1e6, nu = 0.3, t = 0.1 (preCICE/run_example.sh:51-53), bottom edge id 20, other edges id 2, forces
    f_x = 1 + sin(t / 25.01) on the left-edge interface nodes (fluid_solver.cpp:192), `steps` time steps of the coupled
    program (fem-shell_precice.cpp:256-412).  Returns what the program reported plus the checks of the same mesh through
    the C ABI: K against the oracle's assembly, the tip series against (1 + sin(t / 25.01)) x the unit-load solution, and
    the solver term of that solve against a manufactured solution."""
    import importlib
    import os
    import re
    import subprocess

    from tests.helpers import meshes, oracle

    pkg = importlib.import_module("fem-shell_amd")
    name = os.path.join(str(workdir), "flap_full")
    subprocess.check_call([meshgen, "t", str(nx), str(nz), "0", "0", "0.1", "1", "2,20,2,2", "1", "0", "1", "y", name])
    t0 = time.perf_counter()
    r = subprocess.run([coupled_tool, "-nu", "0.3", "-e", "1e6", "-t", "0.1", "-mesh", name + ".xda", "-config", config,
                        "-dt", "0.01", "-axis", "y", "-steps", str(steps), "-fluid", "edge"], capture_output=True, text=True,
                       env=dict(os.environ, FEMSHELL_TIMING="1"))
    wall = time.perf_counter() - t0
    out = {"returncode": r.returncode, "stderr_tail": r.stderr[-500:], "wall_seconds_program": wall, "steps": steps}
    times = re.search(r"Times \[s\]: ([^\n]*)", r.stderr)  # the program's own account of its phases (FEMSHELL_TIMING=1)
    out["program_phase_seconds"] = times.group(1) if times else None
    if r.returncode != 0:
        return out
    g = re.search(r"Coupled run: (\d+) time steps, (\d+) coupling iterations, (\d+) CG iterations, (\d+) assemblies of K, "
                  r"assembly (\S+) s, solves (\S+) s", r.stdout)
    out.update(time_steps=int(g.group(1)), coupling_iterations=int(g.group(2)), cg_iterations=int(g.group(3)),
               assemblies_of_K=int(g.group(4)), assembly_ms=1e3 * float(g.group(5)), solve_seconds=float(g.group(6)),
               linear_solver=re.search(r"Linear solver: ([^\n]*)", r.stdout).group(1) if "Linear solver:" in r.stdout else None)
    tips = [float(v) for v in re.findall(r"tip\[\d+\] node \d+ = (\S+)", r.stdout)]
    probe = int(re.search(r"tip\[0\] node (\d+)", r.stdout).group(1))
    out["tips"] = tips
    # the same mesh through the C ABI
    m = meshes.read_xda(name + ".xda")
    out["triangles"] = int(len(m.tri))
    mat = (0.3, 1e6, 0.1)
    left = [n for n in m.interface_nodes() if abs(m.xyz[n, 0]) < 1e-12]
    loads = np.zeros((m.n_nodes, 6))
    loads[left, 0] = 1.0
    m.loads = loads
    fs = pkg.FemShell(*mat, device=0)
    try:
        fs.set_mesh(m.xyz, m.tri)
        fs.set_dirichlet(m.dirichlet_mask())
        fs.set_loads(loads)
        out["matrix_vs_oracle"] = matrix_parity(fs, m, mat)
        fs.set_preconditioner("amg")
        u_unit, info = fs.solve(rtol=1e-12, max_it=2000)
        out["unit_load_solve"] = {"iterations": info["iterations"], "converged": info["converged"], "solve_seconds": info["solve_seconds"],
                                  "pc_setup_seconds": info["pc_setup_seconds"], "error_estimate": info["error_estimate"]}
        want = [(1.0 + np.sin(t / 25.01)) * u_unit[probe, 0] for t in range(steps)]
        out["tip_series_expected"] = [float(v) for v in want]
        out["tip_series_max_rel_diff"] = float(max(abs(a - b) / abs(b) for a, b in zip(tips, want))) if len(tips) == steps else None
        out["left_edge_nodes"] = len(left)
        out["manufactured"] = manufactured_solve(fs, m, "flap", rtol=1e-10, passes=(1,))
    finally:
        fs.close()
    for ext in (".xda", "_f"):
        try:
            os.remove(name + ext)
        except OSError:
            pass
    return out
