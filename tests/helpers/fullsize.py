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


def matrix_parity(fs, m, mat, chunk=1 << 20):
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
    return out


def manufactured_solve(fs, m, kind, rtol=1e-10, passes=(1,), max_it=3000, with_rounding_correction=True):
    """Solver term at full size: ||u - u_ref|| / ||u_ref|| of multigrid solves of K u = fl(K u*) for the refinement
    pass counts in `passes`; u_ref = u* + delta (tests/helpers/manufactured.py).  The context's loads are replaced."""
    n = m.n_nodes
    u_star = manufactured.smooth_field(m, kind)
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
