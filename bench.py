#!/usr/bin/env python3
"""Benchmark of the fem-shell hot path on MI355X: elements assembled/s and CG iterations/s.

Workload (BASELINE.json, the 4M-tri configuration): 10x10 flat panel, 1414x1414 squares split
into 3,998,792 TRI3 (2,002,225 nodes, 12,013,350 dofs), E=1e7, nu=0.3, t=0.5, all edges simply
supported (boundary id 0), uniform pressure 300 as nodal Fz.  With --gpus N the same mesh is
row-partitioned over N ranks (strong scaling, as BASELINE.json's 1/2/4/8-GPU curve asks).

A "step" is one full femshell_assemble of K and F (inputs resident in HBM; launch + status round trip).  After W warm-up
steps (each an assembly + cg_iters CG iterations) the K timed assembly steps run, then K*cg_iters CG iterations (6x6
block-Jacobi, the oracle's method) are timed the same way (barrier + synchronize on both sides, max over ranks).

Output: rank 0 prints ONE compact JSON line on stdout (< 4 KB: metric, value, ms_per_step, config, roofline, cpu_baseline,
cg_iters_per_s, time_to_solution_s / _iterations, parity_max_rel) and writes everything else to bench_detail.json next to this
script (and to gpurun_out/ when it exists).  At N=1 the detail record carries
  time_to_solution  the same 4M-tri system solved to rtol 1e-10 with the multigrid preconditioner
  parity            the small panel against the oracle, BASELINE configs[1] at full size (Scordelis-Lo roof, 250,632 tri3,
                    rtol 1e-12) against the oracle's refined direct solve, configs[2] / [3] / [4] at full size
  cpu_baseline      the oracle (C port of the reference path, -O3 -march=native + OpenMP, built on this host)
"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _meshgen():
    return importlib.import_module("fem-shell_amd.meshgen")


def panel_mesh(nx):
    """meshGen-equivalent structured triangle mesh (src/meshgen/main_all.cpp:144-224), ul_lr
    diagonals, all four edges boundary id 0, uniform load 300 (main_all.cpp:373)."""
    return _meshgen().structured(nx, nx, 0.0, 0.0, 10.0, 10.0, kind="t", ul_lr=True, bcids=(0, 0, 0, 0),
                                 factor=300.0, loading=2)


def workload_mesh(name, nx):
    """BASELINE.json configurations: panel = configs[3] (flat panel, the default and the 1/2/4/8-GPU curve),
    cylinder = configs[2] (pinched cylinder, same size), roof = configs[1] (Scordelis-Lo, 354x354 squares)."""
    mg = _meshgen()
    if name == "panel":
        return panel_mesh(nx), (0.3, 1e7, 0.5)
    if name == "cylinder":
        m = mg.pinched_cylinder(nx, nx)
        return m, m.material
    if name == "roof":
        m = mg.scordelis_lo(nx)
        return m, m.material
    raise SystemExit("unknown workload " + name)


def kernel_source_digest():
    """Identifies the kernel sources a committed profile belongs to (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fem-shell_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = min(len(cores), logical) if cores else logical
    return model, max(physical, 1), logical


def cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max), None when unlimited or unreadable."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                w = f.read().split()
            if path.endswith("cpu.max"):
                return None if w[0] == "max" else float(w[0]) / float(w[1])
            q = float(w[0])
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                return None if q <= 0 else q / float(f.read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_baseline_worker(nx):
    """Runs in a process of its own (cpu_baseline below starts it with OMP_PROC_BIND=spread OMP_PLACES=cores, before any
    OpenMP runtime exists): the oracle's assembly and block-Jacobi PCG on the 4M-triangle panel itself, threads swept
    from 1 to the physical cores, plus a STREAM triad at every thread count.  All arrays the timed loops stream are first
    touched by the threads that stream them (K inside the threaded assembly, the PCG vectors inside fso_pcg_block_jacobi).
    Prints one JSON object."""
    import ctypes as C

    from tests.helpers import oracle

    model, physical, logical = host_cpu()
    oracle.use_fast_build(1)
    L = oracle.lib()
    L.fso_stream_triad.restype = C.c_double
    L.fso_stream_triad.argtypes = [C.c_int64, C.c_int32]
    m = panel_mesh(nx)
    mat = oracle.material(0.3, 1e7, 0.5)
    dmask = m.dirichlet_mask()
    rowptr, colidx = oracle.bsr_pattern(m.n_nodes, m.tri, m.quad)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    xyz = np.ascontiguousarray(m.xyz)
    tri = np.ascontiguousarray(m.tri, dtype=np.int32)
    loads = np.ascontiguousarray(m.loads)
    n_dof = 6 * m.n_nodes
    # bytes one PCG iteration streams on the host: K in full block CSR (288 + 4 per block), x gathered, y, and the
    # vector passes of the loop (dot p.q 2, x/r update 4+2 writes, r.r 1, z = Minv r 1 + 36/6 + 1, r.z 2, p update 3)
    bytes_per_iter = 292.0 * len(colidx) + 4.0 * (m.n_nodes + 1) + 8.0 * n_dof * (2 + 2 + 6 + 1 + 2 + 6 + 2 + 3)

    def asm_rate(vals, F, repeat):
        return L.fso_time_assembly(m.n_nodes, xyz.ctypes.data_as(dp), len(tri), tri.ctypes.data_as(ip), C.byref(mat),
                                   dmask.ctypes.data_as(C.POINTER(C.c_uint8)), loads.ctypes.data_as(dp),
                                   rowptr.ctypes.data_as(ip), colidx.ctypes.data_as(ip), vals.ctypes.data_as(dp),
                                   F.ctypes.data_as(dp), repeat)

    counts = sorted({t for t in (1, 8, 16, 32, 64, 128, 192, physical) if t <= physical}, reverse=True)
    sweep = []
    for th in counts:
        oracle.set_threads(th)
        # fresh pages per thread count: np.empty does not touch them, the threaded assembly does (first touch by owner)
        vals = np.empty((len(colidx), 6, 6))
        F = np.empty(n_dof)
        t0 = time.perf_counter()
        first = asm_rate(vals, F, 1)
        t_first = time.perf_counter() - t0
        reps = int(max(1, min(10, 2.0 / max(t_first, 1e-3))))
        rate = asm_rate(vals, F, reps) if th > 1 or t_first < 2.0 else first
        its = int(max(4, min(200, 40.0 * th ** 0.7)))
        _, info = oracle.pcg(rowptr, colidx, vals, F, rtol=0.0, max_it=its)
        cg = info["iterations"] / info["seconds"]
        triad = L.fso_stream_triad(1 << 27, 4)
        sweep.append({"threads": th, "elements_per_s": rate, "cg_iters_per_s": cg, "cg_gb_per_s": cg * bytes_per_iter / 1e9,
                      "stream_triad_gb_per_s": triad, "assemblies_timed": reps, "pcg_iterations_timed": info["iterations"]})
        del vals, F
    best_asm = max(sweep, key=lambda r: r["elements_per_s"])
    best_cg = max(sweep, key=lambda r: r["cg_iters_per_s"])
    one = next(r for r in sweep if r["threads"] == 1)
    best_triad = max(r["stream_triad_gb_per_s"] for r in sweep)
    print(json.dumps({
        "value": best_asm["elements_per_s"], "unit": "elements/s", "cores": best_asm["threads"], "kind": "port",
        "cg_iters_per_s": best_cg["cg_iters_per_s"], "cg_cores": best_cg["threads"],
        "cg_gb_per_s": best_cg["cg_gb_per_s"], "cg_bytes_per_iteration": bytes_per_iter,
        "cg_frac_of_host_triad": best_cg["cg_gb_per_s"] / best_triad if best_triad > 0 else None,
        "host_stream_triad_gb_per_s": best_triad,
        "thread_scaling": {"assembly": best_asm["elements_per_s"] / one["elements_per_s"],
                           "cg": best_cg["cg_iters_per_s"] / one["cg_iters_per_s"]},
        "single_thread": {"value": one["elements_per_s"], "unit": "elements/s", "cores": 1, "cg_iters_per_s": one["cg_iters_per_s"]},
        "thread_sweep": sweep,
        "cpu_model": model, "physical_cores": physical, "logical_cpus": logical, "cgroup_cpu_quota": cpu_quota(),
        "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")},
        "build": "gcc -O3 -march=native -fopenmp (oracle/Makefile target `fast`, compiled on this host)",
        "sample": "panel %dx%d (%d tri3): per thread count <=10 assemblies + <=200 PCG its; best of the sweep" % (nx, nx, len(m.tri)),
        "sample_detail": "the benchmark's own mesh: panel %dx%d squares (%d tri3, %d dofs); per thread count 1 first-touch "
                         "assembly + up to 10 timed ones, 4-200 PCG iterations (6x6 block-Jacobi, the oracle's method), a "
                         "STREAM triad of 3 x 1 GiB; `value` / `cg_iters_per_s` are the best of the sweep; "
                         "oracle/femshell_oracle.c" % (nx, nx, len(m.tri), n_dof),
    }))


def cpu_baseline(nx):
    """the CPU restatement of the path (oracle/femshell_oracle.c: same element arithmetic, BSR
    scatter, 6x6 block-Jacobi PCG) built -O3 -march=native with OpenMP on this host and timed on this host's cores, on
    the mesh the metric is quoted on.  A child process does it (cpu_baseline_worker): thread binding has to be in the
    environment before the OpenMP runtime of the process starts, and this one has long loaded torch's."""
    import subprocess

    env = dict(os.environ)
    env.update({"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"})
    env.pop("OMP_NUM_THREADS", None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--nx", str(nx)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if p.returncode != 0:
        return {"error": "cpu baseline worker failed (rc %d): %s" % (p.returncode, p.stderr[-400:])}
    return json.loads(p.stdout.strip().splitlines()[-1])


def parity_small(pkg, device):
    """Displacements of the HIP path vs the CPU oracle on a mesh the oracle finishes in seconds, with both
    preconditioners: solver term (against the refined direct solve of the matrix the GPU assembled), total (against the
    oracle's own assembly) and the kappa-sensitivity that separates them."""
    from tests.helpers import oracle

    m = panel_mesh(64)
    fs = pkg.FemShell(0.3, 1e7, 0.5, device=device)
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    u, info = fs.solve(rtol=1e-12, max_it=20000)
    fs.set_preconditioner("amg")
    ua, infoa = fs.solve(rtol=1e-12, max_it=2000)
    rg, cg, vg, Fg = fs.export_bsr()
    ug = oracle.refined_solve(rg, cg, vg, Fg)
    mat = oracle.material(0.3, 1e7, 0.5)
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u0 = oracle.refined_solve(r0, c0, v0, F0)
    fs.close()
    nrm = np.linalg.norm(u0)
    return {"mesh": "64x64 panel (8192 tri3)",
            "block_jacobi": {"cg_iterations": info["iterations"],
                             "rel_err_solver_term_vs_direct_same_matrix": float(np.linalg.norm(u.ravel() - ug) / nrm),
                             "rel_displacement_error_vs_cpu": float(np.linalg.norm(u.ravel() - u0) / nrm)},
            "multigrid": {"cg_iterations": infoa["iterations"],
                          "rel_err_solver_term_vs_direct_same_matrix": float(np.linalg.norm(ua.ravel() - ug) / nrm),
                          "rel_displacement_error_vs_cpu": float(np.linalg.norm(ua.ravel() - u0) / nrm)},
            "kappa_sensitivity_two_fp64_assemblies": float(np.linalg.norm(ug - u0) / nrm),
            "matrix_rel_diff_vs_oracle": float(np.abs(vg - v0).max() / np.abs(v0).max())}


def parity_config1(pkg, device):
    """BASELINE.json configs[1] at full size, converged, against the oracle: solver term = against
    the refined direct solve of the matrix the GPU assembled; total = against the oracle's own assembly; the
    difference of the two direct solves is the sensitivity kappa * (rounding difference of two FP64 assemblies)."""
    from tests.helpers import oracle

    m, mat = workload_mesh("roof", 354)
    fs = pkg.FemShell(*mat, device=device)
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.set_preconditioner("amg")
    t0 = time.perf_counter()
    u, info = fs.solve(rtol=1e-12, max_it=2000)
    wall = time.perf_counter() - t0
    rg, cg, vg, Fg = fs.export_bsr()
    fs.close()
    t0 = time.perf_counter()
    ug = oracle.refined_solve(rg, cg, vg, Fg, sweeps=4)
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), m.dirichlet_mask(), m.loads)
    u0 = oracle.refined_solve(r0, c0, v0, F0, sweeps=4)
    cpu_s = time.perf_counter() - t0
    nrm = np.linalg.norm(u0)
    return {
        "mesh": "Scordelis-Lo roof 354x354 squares (%d tri3, %d dofs), rtol 1e-12" % (len(m.tri), 6 * m.n_nodes),
        "preconditioner": "smoothed-aggregation multigrid, K cycle, 1 refinement pass",
        "iterations": info["iterations"], "converged": info["converged"], "solve_seconds": info["solve_seconds"],
        "pc_setup_seconds": info["pc_setup_seconds"], "wall_seconds_incl_setup": wall,
        "true_rel_residual": info["true_rel_residual"],
        "rel_err_solver_term_vs_direct_same_matrix": float(np.linalg.norm(u.ravel() - ug) / np.linalg.norm(ug)),
        "rel_err_total_vs_oracle": float(np.linalg.norm(u.ravel() - u0) / nrm),
        "kappa_sensitivity_two_fp64_assemblies": float(np.linalg.norm(ug - u0) / nrm),
        "matrix_rel_diff_vs_oracle": float(np.abs(vg - v0).max() / np.abs(v0).max()),
        "cpu_direct_solves_seconds": cpu_s,
    }


def fullsize_parity(fs, m, mat, kind):
    """BASELINE configs[2] / [3] at their own size: the HIP-assembled K against the oracle's, all 14M
    blocks, and the solver term of the multigrid solve against a manufactured solution (tests/helpers/manufactured.py:
    b = K u* in double-double on the device), with 0 and 1 refinement passes, beside the error estimate the C ABI
    returns.  Replaces the loads of the context."""
    from tests.helpers import fullsize

    # (products=True: femshell_spmv and the double-double femshell_residual of the device against the ORACLE's blocks as well)
    out = {"matrix_vs_oracle": fullsize.matrix_parity(fs, m, mat, products=True, kind=kind)}
    man = fullsize.manufactured_solve(fs, m, kind, rtol=1e-10, passes=(0, 1))
    r0, r1 = man["runs"][0], man["runs"][1]
    out["manufactured_solution"] = {
        "rtol": 1e-10, "rel_err_manufactured": r1["rel_err_vs_manufactured"], "iterations": r1["iterations"],
        "solve_seconds": r1["solve_seconds"], "error_estimate_from_solve_info": r1["error_estimate"],
        "refine_correction_rel": r1["refine_correction_rel"], "refine_residual_reduction": r1["refine_residual_reduction"],
        "true_rel_residual_double_double": r1["true_rel_residual_double_double"],
        "without_refinement_pass": {"rel_err_manufactured": r0["rel_err_vs_manufactured"], "iterations": r0["iterations"],
                                    "true_rel_residual_double_double": r0["true_rel_residual_double_double"]},
        "first_phase_alone_to_100_rtol": {"rel_err_manufactured": man["runs"]["first_phase"]["rel_err_vs_manufactured"],
                                          "iterations": man["runs"]["first_phase"]["iterations"]},
        "rounding_of_b": man.get("rounding_of_b"),
        "note": "u* smooth, zero on the fixed dofs; b = K u* evaluated in double-double on the device and rounded to double; "
                "the reference is u* + K^-1 (fl(b) - K u*)"}
    # the same with u* = the converged solution of the load case itself (the spectrum of the real right-hand side)
    fs.set_loads(m.loads)
    fs.assemble()
    fs.set_preconditioner("amg", refine_passes=1)
    u_load, il = fs.solve(rtol=1e-10, max_it=3000)
    man2 = fullsize.manufactured_solve(fs, m, kind, rtol=1e-10, passes=(1,), u_star=u_load)
    out["manufactured_solution_with_the_load_case_spectrum"] = {
        "rel_err_manufactured": man2["runs"][1]["rel_err_vs_manufactured"], "iterations": man2["runs"][1]["iterations"],
        "error_estimate_from_solve_info": man2["runs"][1]["error_estimate"],
        "note": "u* = the converged solution of the load case (uniform pressure / the two pinch loads), b = K u* in double-double"}
    fs.set_loads(m.loads)
    return out


def headline_load_case_witness():
    """w at the centre of the headline panel (the plate of the thesis' tests D / G, doc/validation.tex:283-295, 518) against
    Navier's series over mesh sizes, and at the headline size once more with the mesh translated -- the same matrix in exact
    arithmetic: what is left of the deviation at 4M triangles is kappa(K) x the rounding of K's FP64 entries (tests/test_gpu_fullsize.py
    test_headline_load_case_against_the_reference_held_plate_answer, profiles/r05_headline_load_case_vs_navier.txt)."""
    from tests.helpers import fullsize

    navier = fullsize.navier_centre_deflection(300.0, 10.0, 1e7, 0.3, 0.5)
    out = {"navier_series_w_centre": navier, "thesis_timoshenko_alpha_0.00406": 0.1064045, "thesis_tri3_64x64": 0.106413, "by_mesh": []}
    for n in (64, 256, 512, 1414):
        r = fullsize.panel_centre_deflection(n)
        out["by_mesh"].append({"squares_per_side": n, "w_centre": r["w_centre"], "rel_dev_from_navier": (r["w_centre"] - navier) / navier,
                               "iterations": r["iterations"], "solver_error_estimate": r["error_estimate"]})
    moved = fullsize.panel_centre_deflection(1414, shift=(3.0, 7.0, 0.0))
    out["headline_mesh_translated_by_3_7_0"] = {"w_centre": moved["w_centre"], "rel_dev_from_navier": (moved["w_centre"] - navier) / navier,
                                                "rel_change_against_the_untranslated_mesh": abs(moved["w_centre"] - out["by_mesh"][-1]["w_centre"]) / navier}
    out["note"] = ("second-order convergence up to 512^2 (-5.0e-4, -3.1e-5, -8.9e-6); at 1414^2 the discretisation error would be 1e-6, the answer is "
                   "1e-4 off and moves by as much under a translation of the mesh: the sensitivity of the solution to the rounding of K's own "
                   "entries, kappa x eps (the solver term is the error estimate); the reference's FP64 assembly rounds the same entries")
    return out


def config4_coupled_flap():
    """BASELINE configs[4] at its own size on one GPU: the coupled program (FEM-shell-precice, in-process dummy fluid) on the
    1M-triangle flap, three time steps; K against the oracle, tip series against the unit-load solution, manufactured
    solution (tests/helpers/fullsize.py: the checks of tests/test_host_tools.py::test_coupled_flap_config5_full_size)."""
    import tempfile

    from tests.helpers import fullsize
    host = os.path.join(ROOT, "fem-shell_amd", "host")
    config = os.path.join(ROOT, "tests", "golden", "coupling", "inprocess_config.xml")
    try:
        with tempfile.TemporaryDirectory() as td:
            o = fullsize.coupled_flap_full_size(os.path.join(host, "FEM-shell-precice"), os.path.join(host, "meshGen"), td, config, steps=3)
    except Exception as ex:  # noqa: BLE001 -- the headline line must still be printed
        return {"error": str(ex)[-300:]}
    if o.get("returncode") != 0:
        return {"error": o.get("stderr_tail")}
    man = o["manufactured"]["runs"][1]
    return {"workload": "flap 0.1 x 1 in the x-z plane, 500 x 1000 squares = %d tri3, E 1e6 nu 0.3 t 0.1, dummy fluid f_x = 1 + sin(t/25.01) on "
                        "the %d left-edge interface nodes, 3 time steps (the reference's run: 400), 1 GPU (the config's 2 GPUs: no such lease)"
                        % (o["triangles"], o["left_edge_nodes"]),
            "linear_solver": o["linear_solver"], "time_steps": o["time_steps"], "coupling_iterations": o["coupling_iterations"],
            "cg_iterations": o["cg_iterations"], "solve_seconds": o["solve_seconds"], "assemblies_of_K": o["assemblies_of_K"],
            "assembly_ms": o["assembly_ms"], "wall_seconds_program": o["wall_seconds_program"],
            "wall_note": "mesh reading (30 MB of XDA text), symbolic phase, assembly, multigrid setup, coupling loop, process start",
            "program_phase_seconds": o.get("program_phase_seconds"),
            "tip_displacements": o["tips"], "tip_series_max_rel_diff_vs_unit_load_solution": o["tip_series_max_rel_diff"],
            "matrix_vs_oracle": o["matrix_vs_oracle"],
            "manufactured_solution_rel_err": man["rel_err_vs_manufactured"], "manufactured_solution_iterations": man["iterations"],
            "manufactured_solution_error_estimate": man["error_estimate"]}


def config2_cylinder(pkg, device, steps, warmup, nx, roof):
    """BASELINE configs[2]: pinched cylinder, 4M tri3, one GPU, 'assembly HBM-GB/s vs roofline reported' -- the same
    measurements as the headline panel (timed assembly steps, k_assemble from HIP events, a short CG run), the multigrid
    solve of the pinched load case, and the full-size parity of this mesh."""
    m, mat = workload_mesh("cylinder", nx)
    fs = pkg.FemShell(*mat, device=device)
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    for _ in range(max(warmup, 25)):  # (the first twenty launches of a kernel run some 10 % slow: clocks, TLBs)
        fs.assemble()
    fs.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fs.assemble()  # (as in the headline leg: one full femshell_assemble per step)
    fs.sync()
    t_asm = time.perf_counter() - t0
    asm_ms, asm_bytes = fs.time_kernel(pkg.KERNEL_ASSEMBLE, max(5, steps))  # (right behind the timed steps, as in the headline leg)
    fs.solve(rtol=0.0, max_it=50, fetch=False)
    fs.sync()
    t0 = time.perf_counter()
    _, info = fs.solve(rtol=0.0, max_it=steps * 50, fetch=False)
    fs.sync()
    t_cg = time.perf_counter() - t0
    spmv_ms, spmv_bytes = fs.time_kernel(pkg.KERNEL_SPMV, max(5, steps))
    fs.set_preconditioner("amg")
    _, ia = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
    out = {"workload": "pinched cylinder R=300 L=600 t=3, E=3e6 nu=0.3: %dx%d squares -> %d tri3, %d nodes" % (nx, nx, len(m.tri), m.n_nodes),
           "elements_per_s": len(m.tri) * steps / t_asm, "ms_per_step": 1e3 * t_asm / steps,
           "cg_iters_per_s": info["iterations"] / t_cg,
           "roofline_assembly": dict(roof(asm_ms, asm_bytes), kernel=fs.assembly_kernel()),
           "roofline_cg_spmv": dict(roof(spmv_ms, spmv_bytes), kernel="k_spmv_sym"),
           "time_to_solution": {"rtol": 1e-10, "iterations": ia["iterations"], "converged": ia["converged"],
                                "solve_seconds": ia["solve_seconds"], "pc_setup_seconds": ia["pc_setup_seconds"],
                                "error_estimate": ia["error_estimate"], "refine_passes_done": ia["refine_passes_done"]}}
    out["parity"] = fullsize_parity(fs, m, mat, "cylinder")
    fs.close()
    return out


def amg_iteration_roofline(fs, info, n_nodes, args):
    """Roofline of one multigrid-preconditioned iteration: the bytes the cycle streams as it is built (femshell_amg_cycle_bytes:
    single-precision copies, increments, real product counts -- csrc/amg_solve.cpp) per level, over the solve's time per
    iteration; per-level milliseconds from the committed kernel trace of the same solve (tools/amg_level_times.py) while the
    kernel sources are the ones it was taken with."""
    per_level = [float(b) for b in fs.amg_cycle_bytes()]
    its, secs = max(info["iterations"], 1), info["solve_seconds"]
    krylov = info["bytes_per_iteration"] - sum(per_level)
    out = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
           "achieved": info["bytes_per_iteration"] * its / secs / 1e9,
           "ms_per_iteration": 1e3 * secs / its,
           "algorithmic_gb_per_iteration": info["bytes_per_iteration"] / 1e9,
           "algorithmic_gb_krylov_passes_on_level_0": krylov / 1e9,
           "algorithmic_gb_of_the_cycle_by_level": [b / 1e9 for b in per_level],
           "nodes_by_level": [l["n_nodes"] for l in fs.amg_levels()]}
    out["frac"] = out["achieved"] / HBM_PEAK_GBS
    path = os.path.join(ROOT, "profiles", "r05_amg_by_level.json")
    if args.workload == "panel" and args.nx == 1414 and os.path.exists(path):
        with open(path) as f:
            prof = json.load(f)
        if prof.get("kernel_source_digest") == kernel_source_digest():
            out["by_level_ms_from_committed_profile"] = prof["ms_per_iteration_by_level"]
            out["by_level_ms_source"] = "profiles/r05_amg_by_level.json (rocprofv3 --kernel-trace of tools/amg_probe.py panel 1414)"
            tot = sum(prof["ms_per_iteration_by_level"]) + prof.get("ms_per_iteration_unassigned", 0.0)
            out["frac_from_committed_profile"] = info["bytes_per_iteration"] / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            out["by_level_ms_from_committed_profile"] = None
            out["by_level_ms_source"] = "stale: profiles/r05_amg_by_level.json was taken with other kernel sources"
    return out


def mfma_counter_summary():
    path = os.path.join(ROOT, "profiles", "r05_pmc_mfma.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    out = dict(d.get("all_kernels_of_the_inverse", {}), source="profiles/r05_pmc_mfma.json")
    upd = d.get("kernels", {}).get("k_dense_update") or d.get("kernels", {}).get("k_dense_update<true>")
    if upd:
        out["k_dense_update"] = {"tflops_by_counter": upd["tflops_by_counter"], "mfma_busy_fraction": upd["mfma_busy_fraction"],
                                 "launches": upd["launches"]}
    return out


def jacobi_extrapolation(hist, target=1e-10):
    """Residual history of the fixed-count block-Jacobi run: decades per 1000 iterations over its second half and the
    iteration count that slope implies for `target` (the solve that block-Jacobi alone would need)."""
    h = np.asarray(hist)
    if len(h) < 40 or not np.all(h > 0):
        return None
    a, b = len(h) // 2, len(h) - 1
    slope = (np.log10(h[b]) - np.log10(h[a])) / (b - a)
    out = {"rel_residual_after": float(h[b]), "iterations_run": int(len(h)), "decades_per_1000_iterations": float(1000.0 * slope)}
    if slope < 0:
        out["extrapolated_iterations_to_1e-10"] = float(len(h) + (np.log10(target) - np.log10(h[b])) / slope)
    return out


METRIC = "elements assembled/s + CG iters/s, 4M-tri shell, 1/2/4/8 MI355X"
DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 4096  # bytes of the final stdout line (the driver's record keeps a tail of stdout; round 5's 20.7 KB line was not parsed)


def _walk(d, path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def parity_max_rel(detail):
    """Worst relative figure of the parity sections of the detail record: assembled K against the oracle's (every mesh the run
    checked) and the solver term of the displacements (against the refined direct solve of the same matrix on the small meshes,
    against manufactured solutions at full size).  None when the run carried no parity section (N > 1, --profile)."""
    paths = [
        ("parity", "small", "matrix_rel_diff_vs_oracle"),
        ("parity", "small", "block_jacobi", "rel_err_solver_term_vs_direct_same_matrix"),
        ("parity", "small", "multigrid", "rel_err_solver_term_vs_direct_same_matrix"),
        ("parity", "config1_scordelis_lo_250k", "matrix_rel_diff_vs_oracle"),
        ("parity", "config1_scordelis_lo_250k", "rel_err_solver_term_vs_direct_same_matrix"),
        ("parity", "config3_flat_panel_4M", "matrix_vs_oracle", "max_entry_diff_over_max_entry"),
        ("parity", "config3_flat_panel_4M", "manufactured_solution", "rel_err_manufactured"),
        ("parity", "config3_flat_panel_4M", "manufactured_solution_with_the_load_case_spectrum", "rel_err_manufactured"),
        ("config2_pinched_cylinder_4M", "parity", "matrix_vs_oracle", "max_entry_diff_over_max_entry"),
        ("config2_pinched_cylinder_4M", "parity", "manufactured_solution", "rel_err_manufactured"),
        ("config2_pinched_cylinder_4M", "parity", "manufactured_solution_with_the_load_case_spectrum", "rel_err_manufactured"),
        ("config4_coupled_flap_1M", "matrix_vs_oracle", "max_entry_diff_over_max_entry"),
        ("config4_coupled_flap_1M", "manufactured_solution_rel_err"),
    ]
    seen = [v for v in (_walk(detail, p) for p in paths) if isinstance(v, (int, float))]
    return max(seen) if seen else None


def compact_line(detail):
    """The ONE stdout line of the run: what the driver records and nothing else.  Built from the detail
    record (which goes to bench_detail.json); tests/test_bench_guard.py holds it below LINE_LIMIT bytes on a canned record."""
    cfg = detail.get("config", {})
    roof = detail.get("roofline") or {}
    cpu = detail.get("cpu_baseline") or {}
    tts = detail.get("time_to_solution") or {}
    line = {
        "metric": detail.get("metric", METRIC), "value": detail.get("value"), "unit": detail.get("unit", "elements/s"),
        "n_gpus": detail.get("n_gpus"), "steps": detail.get("steps"), "warmup": detail.get("warmup"),
        "ms_per_step": detail.get("ms_per_step"), "higher_is_better": True, "scaling": detail.get("scaling", "strong"),
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {k: cfg.get(k) for k in ("workload", "parallelism", "rccl_ranks_seen", "preconditioner", "assembly_step",
                                            "warmup_step", "cg_iters_per_step", "time_to_solution_preconditioner") if k in cfg},
        "roofline": {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch",
                                              "algorithmic_bytes_per_launch")} if roof else None,
        "cpu_baseline": {k: cpu.get(k) for k in ("value", "unit", "cores", "cpu_model", "cg_iters_per_s", "kind", "sample", "error")
                         if k in cpu} if cpu else None,
        "cg_iters_per_s": detail.get("cg_iters_per_s"),
        "time_to_solution_s": tts.get("solve_seconds"),
        "time_to_solution_iterations": tts.get("iterations"),
        # (the one-time multigrid setup of the first solve -- PETSc's KSPSetUp -- is not in time_to_solution_s; the two beside it)
        "pc_setup_s": tts.get("pc_setup_seconds"),
        "first_solve_wall_s": tts.get("wall_seconds_first_solve"),
        "parity_max_rel": parity_max_rel(detail),
        "detail_file": DETAIL_FILE,
    }
    if detail.get("error"):
        line["error"] = str(detail["error"])[:300]
    s = json.dumps(line)
    if len(s) > LINE_LIMIT:  # (prose that grew: the numbers stay, the strings are cut)
        for sect in ("config", "roofline", "cpu_baseline"):
            for k, v in (line.get(sect) or {}).items():
                if isinstance(v, str) and len(v) > 60:
                    line[sect][k] = v[:57] + "..."
        s = json.dumps(line)
    if len(s) > LINE_LIMIT:
        raise RuntimeError("bench line of %d bytes" % len(s))
    return s


def write_detail(detail):
    """The full record -- witnesses, per-level bytes, counters, the other configs, notes -- next to the script and, where the
    GPU box collects files (gpurun_out/), there as well.  Never on stdout."""
    txt = json.dumps(detail, indent=1)
    where = os.environ.get("FEMSHELL_BENCH_DETAIL_DIR")  # (tests)
    for d in ((where,) if where else (ROOT, os.path.join(ROOT, "gpurun_out"))):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, DETAIL_FILE), "w") as f:
                f.write(txt + "\n")
        except OSError as e:
            sys.stderr.write("bench: %s not written in %s: %s\n" % (DETAIL_FILE, d, e))


def emit(detail):
    """Rank 0's output: the detail record to its file, then the compact line as the LAST (and only) line on stdout."""
    write_detail(detail)
    sys.stdout.flush()
    print(compact_line(detail), flush=True)


def guarded_multi_rank_run():
    """N > 1: the rank's work runs in a CHILD process started before anything has touched the GPU; this parent only waits.
    A first run on real RCCL that stalls (a rank that never joins, a collective a peer never enters) then ends with a
    message instead of a hang: the library's own watchdog (csrc/comm.hpp CommWatch, FEMSHELL_COMM_TIMEOUT, default 120 s)
    ends the stuck child with status 86, and should even that not fire the parent kills the child's process group after
    FEMSHELL_BENCH_TIMEOUT seconds (default 1500).  Either way rank 0's parent prints one JSON line with "error", the
    phase, the tail of the child's stderr and of RCCL's own log (NCCL_DEBUG=WARN into a per-rank file) and every parent exits
    non-zero, which makes the launcher end the rank group.  Nothing is re-executed in a process that has used the GPU."""
    import collections
    import signal
    import subprocess
    import tempfile
    import threading

    rank = int(os.environ.get("RANK", "0"))
    limit = float(os.environ.get("FEMSHELL_BENCH_TIMEOUT", "1500"))
    logdir = tempfile.mkdtemp(prefix="femshell_bench_")
    env = dict(os.environ, FEMSHELL_BENCH_CHILD="1")
    env.setdefault("NCCL_DEBUG", "WARN")
    env.setdefault("NCCL_DEBUG_FILE", os.path.join(logdir, "rccl_rank%d.log" % rank))
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=subprocess.PIPE,
                             start_new_session=True)
    tail = collections.deque(maxlen=40)

    def pump():
        for raw in child.stderr:
            line = raw.decode(errors="replace")
            sys.stderr.write(line)
            tail.append(line.rstrip())

    th = threading.Thread(target=pump, daemon=True)
    th.start()

    def end_group(*_):
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except OSError:
            pass

    signal.signal(signal.SIGTERM, lambda *_: (end_group(), sys.exit(143)))
    why = None
    try:
        rc = child.wait(timeout=limit)
        if rc != 0:
            why = "rank %d: the bench process ended with status %d%s" % (rank, rc, " (the library's watchdog: a blocking phase made no "
                                                                         "progress, see stderr_tail)" if rc == 86 else "")
    except subprocess.TimeoutExpired:
        end_group()
        rc = 124
        why = "rank %d: no result after %.0f s (FEMSHELL_BENCH_TIMEOUT); process group killed" % (rank, limit)
    th.join(timeout=2.0)
    if why is None:
        return 0
    rccl = []
    try:
        with open(env["NCCL_DEBUG_FILE"], errors="replace") as f:
            rccl = f.read().splitlines()[-20:]
    except OSError:
        pass
    phase = next((ln for ln in reversed(tail) if "[femshell watchdog]" in ln or "femshell error" in ln), None)
    line = {"metric": METRIC, "value": None, "unit": "elements/s",
            "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": why[:300], "phase": (phase or "")[:300] or None,
            "retry_hint": "FEMSHELL_HALO_OVERLAP=0 (halo group off its second stream); FEMSHELL_COMM_TIMEOUT=<s> widens the watchdog",
            "detail_file": DETAIL_FILE}
    sys.stderr.write(why + "\n")
    if rank == 0:
        write_detail(dict(line, error=why, phase=phase, stderr_tail=list(tail)[-15:], rccl_debug_tail=rccl))
        print(json.dumps(line), flush=True)
    return rc if rc != 0 else 1


def main():
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("FEMSHELL_BENCH_CHILD") != "1" and "--cpu-baseline-worker" not in sys.argv:
        sys.exit(guarded_multi_rank_run())
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=None, help="squares per side (default 1414 -> 3,998,792 tri3; roof: 354)")
    ap.add_argument("--workload", default="panel", choices=["panel", "cylinder", "roof"])
    ap.add_argument("--cg-iters", type=int, default=50, help="CG iterations per step in the CG phase")
    ap.add_argument("--jacobi-probe-iters", type=int, default=5000,
                    help="untimed block-Jacobi run whose residual history goes into time_to_solution.block_jacobi_alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-full-parity", action="store_true", help="skip the 250k-element converged parity (two CPU direct solves)")
    ap.add_argument("--no-fullsize-parity", action="store_true",
                    help="skip the 4M-triangle parity sections (oracle assembly of 14M blocks + manufactured solutions, panel and cylinder)")
    ap.add_argument("--profile", action="store_true",
                    help="for rocprofv3 runs: only the timed 4M-tri phases (no parity probes, no CPU baseline, no multigrid solve)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:  # child process of cpu_baseline(): host only, never touches the GPU
        cpu_baseline_worker(args.nx or 1414)
        return

    import torch  # first: its HIP runtime then serves libfemshell too (same SONAME)
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("FEMSHELL_BENCH_SAME_DEVICE") == "1":
        local_rank = 0  # test hook: several ranks on the one GPU of a test box (with the fake RCCL of tests/helpers)
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane only (unique-id broadcast, barriers, timing max); the data path
        # (halo exchange, CG all-reduce) is RCCL inside libfemshell
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    pkg = importlib.import_module("fem-shell_amd")
    if args.nx is None:
        args.nx = 354 if args.workload == "roof" else 1414
    m, (nu, E, thick) = workload_mesh(args.workload, args.nx)
    fs = pkg.FemShell(nu, E, thick, device=local_rank, rank=rank, world_size=world)
    if world > 1:
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.from_numpy(pkg.comm_unique_id().copy())
        dist.broadcast(uid, src=0)
        fs.comm_init(uid.numpy())
    rccl_ranks = fs.comm_ranks()

    # the boxes of the pool differ by about 10 % on bandwidth-bound kernels: a streaming copy of 2 x 1 GiB on this box,
    # so that the roofline fractions of different runs can be compared
    def device_copy_rate():
        n = 1 << 27  # doubles
        a = torch.empty(n, dtype=torch.float64, device="cuda")
        b = torch.ones(n, dtype=torch.float64, device="cuda")
        for _ in range(3):
            a.copy_(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            a.copy_(b)
        e1.record()
        torch.cuda.synchronize()
        rate = 10 * 2 * n * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        torch.cuda.empty_cache()
        return rate
    copy_gbs = device_copy_rate()

    def box_stream_rates():
        """Streaming write / read / copy rates of this box by tools/lab/hbm_rw (plain 16-byte-per-lane kernels over 4.3 GB, a child
        process; built by __graft_entry__.build()): the practical roofs a write-heavy kernel (the assembly: 82 % stores) and a
        read-heavy one (the products) are to be held against -- HBM3E's 8 TB/s is reached by neither direction on these boxes."""
        exe = os.path.join(ROOT, "tools", "lab", "hbm_rw")
        if rank != 0 or not os.path.exists(exe):
            return None
        try:
            r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120)
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001 (a yardstick, not a measurement the line depends on)
            return {"error": repr(e)}
    stream = box_stream_rates() if not args.profile else None

    def device_alloc_ms():
        """Wall time of hipMalloc + hipFree of 4 GiB: a few ms on a fresh box, a hundred and more right after a run that
        churned the card's memory (the GPU test-suite: 138 contexts) -- the multigrid setup allocates and frees about 10 GB
        of transient buffers, and `pc_setup_seconds` was 0.5 s instead of 0.18 s in bench runs that followed the test-suite
        on the same box (the lap "uploads" of FEMSHELL_AMG_VERBOSE=1: 0.20 s instead of 0.005 s)."""
        torch.cuda.synchronize()
        t = time.perf_counter()
        x = torch.empty(1 << 32, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
        return 1e3 * (time.perf_counter() - t)
    alloc_ms = device_alloc_ms()
    t0 = time.perf_counter()
    fs.set_mesh(m.xyz, m.tri)
    setup_s = time.perf_counter() - t0
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    n_elem, n_nodes = len(m.tri), m.n_nodes

    # ---- cold figure: the very first assembly of the process
    fs.sync()
    t0 = time.perf_counter()
    fs.assemble()
    fs.sync()
    cold_ms = 1e3 * (time.perf_counter() - t0)
    # ---- the W warm-up steps of the contract, each the whole of what a timed step is: one femshell_assemble and cg_iters CG
    # iterations (the first launches of a fresh process run ~12 % slow -- clocks, TLBs, first touches; five such steps are 0.2 s of
    # device work, and profiles/r06_warmup_ab.txt shows the timed steps behind them at the rate 30 + 200 extra launches gave).
    # FEMSHELL_BENCH_EXTRA_WARMUP=1 adds those extra launches for that A/B; the default runs nothing but the W steps.
    if os.environ.get("FEMSHELL_BENCH_EXTRA_WARMUP") == "1":
        for _ in range(30):
            fs.assemble()
        fs.solve(rtol=0.0, max_it=200, fetch=False)
    for _ in range(args.warmup):
        fs.assemble()
        fs.solve(rtol=0.0, max_it=args.cg_iters, fetch=False)

    # ---- timed phase 1: K assembly steps, each one full femshell_assemble: launch, the status word's
    # round trip to the host (a degenerate element on any rank; on N ranks an all-reduce), return.  `value` / `ms_per_step`.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fs.assemble()
    fs.sync()
    barrier()
    t_asm = max_over_ranks(time.perf_counter() - t0)
    # beside it: the same K steps enqueued with femshell_assemble_async, the status of all of them collected by one
    # femshell_sync (`ms_per_step_async`: what a caller that assembles ahead of the device pays)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fs.assemble(wait=False)
    fs.sync()
    barrier()
    t_asm_async = max_over_ranks(time.perf_counter() - t0)
    # the kernel of these steps by HIP events on the library's stream, in the state the timed steps ran in (measured after
    # the CG kernels' event pairs, whose synchronisations let the clocks sag, its first launches run 3-4 % slow)
    asm_ms, asm_bytes = fs.time_kernel(pkg.KERNEL_ASSEMBLE, max(5, args.steps))

    # ---- timed phase 2: K * cg_iters CG iterations (one solve call, no host round trip inside)
    n_it = args.steps * args.cg_iters
    barrier()
    t0 = time.perf_counter()
    _, info = fs.solve(rtol=0.0, max_it=n_it, fetch=False)
    fs.sync()
    barrier()
    t_cg = max_over_ranks(time.perf_counter() - t0)
    jacobi_hist = fs.residual_history()
    if world == 1 and not args.profile and args.jacobi_probe_iters > n_it:
        # where block-Jacobi alone stands on this system after a few thousand iterations (time_to_solution)
        fs.solve(rtol=0.0, max_it=args.jacobi_probe_iters, fetch=False)
        jacobi_hist = fs.residual_history()

    # ---- per-kernel durations with HIP events on the library's stream
    reps = max(5, args.steps)
    spmv_ms, spmv_bytes = fs.time_kernel(pkg.KERNEL_SPMV, reps)
    upd_ms, upd_bytes = fs.time_kernel(pkg.KERNEL_CG_UPDATE, reps)
    dir_ms, dir_bytes = fs.time_kernel(pkg.KERNEL_CG_DIRECTION, reps)

    # HBM traffic per launch cannot be read inside this process (rocprofv3 --pmc has to own the run): the numbers
    # come from the committed summary of separate FETCH_SIZE / WRITE_SIZE passes over `bench.py --profile`
    # (profiles/*_pmc_hbm_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  They are
    # attached only for the workload they were taken on and only while the kernel sources are the ones they were
    # taken with (digest recorded by tools/pmc_summary.py); otherwise null.
    traffic, traffic_note = {}, None
    if world == 1 and args.workload == "panel" and args.nx == 1414:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm_traffic.json"))
        if cands:
            with open(os.path.join(ROOT, "profiles", cands[-1])) as f:
                pm = json.load(f)
            meta = pm.pop("_meta", {})
            if meta.get("kernel_source_digest") == kernel_source_digest():
                for kname, v in pm.items():
                    traffic[kname.split("::")[-1].split("<")[0]] = v["read_bytes_x2_gfx950"] + v["write_bytes"]
                    traffic[kname.split("::")[-1].split("<")[0] + ":read"] = v["read_bytes_x2_gfx950"]
                    traffic[kname.split("::")[-1].split("<")[0] + ":write"] = v["write_bytes"]
                traffic["_source"] = "profiles/" + cands[-1]
            else:
                traffic_note = "stale: profiles/%s was taken with other kernel sources" % cands[-1]
    # FP64 work per launch from the committed instruction-counter passes (tools/pmc_flops.py), same rule
    fp64 = {}
    if world == 1 and args.workload == "panel" and args.nx == 1414:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fp64.json"))
        if cands and traffic.get("_source"):
            with open(os.path.join(ROOT, "profiles", cands[-1])) as f:
                for kname, v in json.load(f).items():
                    fp64[kname.split("::")[-1].split("<")[0]] = v["flops_per_launch"]

    def roof(ms, nbytes, kernel=None):
        gbs = nbytes / (ms * 1e-3) / 1e9
        out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "traffic": traffic.get(kernel), "traffic_from_committed_profile": traffic.get("_source") if kernel in traffic else traffic_note,
               "ms_per_launch": ms, "algorithmic_bytes_per_launch": nbytes}
        if kernel in traffic:
            out["traffic_gb_per_s"] = traffic[kernel] / (ms * 1e-3) / 1e9
            # (how fast the kernel moves the bytes it really moves, against what this box reaches with a plain device-to-device
            #  copy of 1 GiB -- the practical roof of a mixed read / write stream, 4.8-5.2 TB/s on the boxes of this pool)
            out["traffic_rate_over_box_streaming_copy_rate"] = out["traffic_gb_per_s"] / copy_gbs
            if stream and "write_gb_per_s" in stream:
                # the same against the box's plain streaming kernels (tools/lab/hbm_rw): the mix of this kernel's reads and writes
                # at the box's read and write rates is the time a perfect stream of the same bytes would take
                rd = traffic.get(kernel + ":read"), traffic.get(kernel + ":write")
                if rd[0] is not None and rd[1] is not None:
                    t_stream = rd[0] / (stream["read_gb_per_s"] * 1e9) + rd[1] / (stream["write_gb_per_s"] * 1e9)
                    out["time_of_a_plain_stream_of_the_same_bytes_ms"] = 1e3 * t_stream
                    out["frac_of_the_box_streaming_rates"] = 1e3 * t_stream / ms
        if kernel in fp64:  # FP64 vector peak 78.6 TFLOP/s (MI355X_MICROARCH.md)
            out["fp64_gflop_per_launch"] = fp64[kernel] / 1e9
            out["fp64_tflops"] = fp64[kernel] / (ms * 1e-3) / 1e12
            out["fp64_frac_of_78.6_tflops"] = out["fp64_tflops"] / 78.6
        return out

    tts = None
    if not args.profile:
        # ---- time to solution on the same system: multigrid-preconditioned flexible CG to rtol 1e-10.  Row-partitioned
        # runs: the hierarchy is row-partitioned too (csrc/amg_dist.cpp: rank-local aggregates, levels above 60,000 nodes split
        # over the ranks, the rest all-gathered); a failure on one rank reaches all of them as an error
        fs.set_preconditioner("amg")
        fs.sync()
        t0 = time.perf_counter()
        _, ia = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        fs.sync()
        wall = time.perf_counter() - t0
        sst = fs.amg_setup_stats()
        dst = fs.amg_dense_stats()
        _, ib = fs.solve(rtol=1e-10, max_it=3000, fetch=False)  # hierarchy reused (the coupled program re-solves)
        fs.set_preconditioner("amg", refine_passes=0)            # same hierarchy, no refinement pass
        _, ic = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        tts = {"preconditioner": "smoothed-aggregation multigrid (rigid-body modes, Chebyshev/block-Jacobi smoothing, K cycle), "
                                 "flexible CG, 1 refinement pass with a double-double residual",
               "precision": "FP64 arithmetic throughout; the Krylov method, the K cycle's products, the Galerkin operators, residuals and "
                            "iterates are FP64 data; what the cycle only smooths or transfers with is read from single-precision copies "
                            "(level operators, D^-1, P, R on levels of >= 4096 nodes; products, transposed products and directions of "
                            "the smoother), the cycle's two residuals are increments on those copies (DESIGN section 5)",
               "rtol": 1e-10, "iterations": ia["iterations"], "converged": ia["converged"],
               "solve_seconds": ia["solve_seconds"], "pc_setup_seconds": ia["pc_setup_seconds"], "wall_seconds_first_solve": wall,
               "solve_seconds_hierarchy_reused": ib["solve_seconds"], "levels": ia["amg_levels"],
               "without_refinement_pass": {"iterations": ic["iterations"], "solve_seconds": ic["solve_seconds"],
                                           "note": "recurrence residual 1e-10 reached; the displacement error then stalls at kappa*eps (DESIGN section 5)"},
               "operator_complexity": ia["operator_complexity"], "true_rel_residual_double_double": ia["true_rel_residual"],
               "error_estimate": ia["error_estimate"], "refine_passes_done": ia["refine_passes_done"],
               "refine_correction_rel": ia["refine_correction_rel"], "refine_residual_reduction": ia["refine_residual_reduction"],
               "algorithmic_gb_per_iteration": ia["bytes_per_iteration"] / 1e9,
               "achieved_gb_per_s": ia["bytes_per_iteration"] * ia["iterations"] / ia["solve_seconds"] / 1e9,
               "roofline_amg_iteration": amg_iteration_roofline(fs, ia, n_nodes, args),
               "setup_first_coarsening_on_device": {
                   "prolongator_ms": sst["prolongator_ms"], "ap_ms": sst["ap_ms"], "restriction_ms": sst["restriction_ms"],
                   "galerkin_ms": sst["galerkin_ms"],
                   "galerkin_kernel": "k_amg_galerkin_mfma (v_mfma_f64_16x16x4_f64, one wave per coarse row)" if sst["galerkin_on_matrix_cores"] else "k_amg_galerkin (vector ALUs, one lane per result block)",
                   "galerkin_useful_gflop": sst["galerkin_useful_flops"] / 1e9,
                   "galerkin_useful_tflops": sst["galerkin_useful_flops"] / max(sst["galerkin_ms"], 1e-9) / 1e9},
               # the dense, GEMM-shaped contraction of the preconditioner: inverse of the coarsest operator by symmetric block
               # sweeps on the matrix cores (csrc/amg_dense.hip); `achieved` counts the flops issued as v_mfma_f64_16x16x4_f64
               "setup_coarsest_inverse_on_matrix_cores": None if not dst["n"] else {
                   "dofs": dst["n"], "ms": dst["ms"], "dropped_directions": dst["dropped_directions"],
                   "mfma_gflop_issued": dst["mfma_flops_issued"] / 1e9, "useful_gflop_n_cubed": dst["useful_flops"] / 1e9,
                   "lower_triangle_traffic_gb": dst["bytes"] / 1e9, "traffic_gb_per_s": dst["bytes"] / max(dst["ms"], 1e-9) / 1e6,
                   "roofline": {"bound": "mfma", "achieved": dst["mfma_flops_issued"] / max(dst["ms"], 1e-9) / 1e9, "peak": 78.6,
                                "unit": "TFLOP/s", "frac": dst["mfma_flops_issued"] / max(dst["ms"], 1e-9) / 1e9 / 78.6,
                                "note": "FP64 matrix peak 78.6 TFLOP/s; 64x64 tiles, block sweeps of width 128: the lower triangle is read and "
                                        "written once per sweep (58 sweeps at 7386 dofs); the pivot block of the next sweep is inverted by a launch of "
                                        "its own on a second stream beside the trailing update, the two meeting through device-side counters "
                                        "(FEMSHELL_AMG_DENSE_LOOKAHEAD=0: in front of every sweep instead, 1.4 times the time)"},
                   # the same by rocprofv3 counters (tools/pmc_mfma.py: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops over the kernel times of an
                   # unprofiled trace; SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x GRBM_GUI_ACTIVE / 8), committed summary of the same solve
                   "mfma_counters_from_committed_profile": mfma_counter_summary()},
               "block_jacobi_alone": jacobi_extrapolation(jacobi_hist)}
        free_b, total_b = torch.cuda.mem_get_info()
        tts["hbm_in_use_gb_max_over_ranks"] = max_over_ranks((total_b - free_b) / 1e9)
        tts["pc_setup_seconds_max_over_ranks"] = max_over_ranks(ia["pc_setup_seconds"])
        if world > 1:
            pi = fs.amg_partition_info()
            tts["row_partitioned_hierarchy"] = {
                "levels_split_over_the_ranks": pi["partitioned_levels"],
                "operator_bytes_of_those_levels_max_over_ranks_gb": max_over_ranks(pi["bytes_partitioned"] / 1e9),
                "operator_bytes_of_those_levels_this_rank_gb": pi["bytes_partitioned"] / 1e9,
                "operator_bytes_replicated_on_every_rank_gb": pi["bytes_replicated"] / 1e9,
                "rows_of_the_last_split_level_on_rank_0": pi["rows_on_last_partitioned_level"],
                "ghost_rows_there": pi["ghost_rows_on_last_partitioned_level"],
                "nodes_of_the_first_replicated_level": pi["nodes_of_first_replicated_level"],
                "note": "aggregates never span ranks; rows of Q, P and A P along the cuts are exchanged once at setup; per cycle every "
                        "product of a split level is preceded by a halo exchange, the K cycle's sums are all-reduced, the first "
                        "replicated level's right-hand side is all-gathered (DESIGN section 6)"}
        if world == 1:
            # the same solve with every copy the cycle reads in FP64 (FEMSHELL_AMG_SMOOTH_F32=0: no single-precision operators,
            # D^-1, transfers, vectors; coarsest inverse FP64): what the mixed-precision preconditioner buys
            os.environ["FEMSHELL_AMG_SMOOTH_F32"] = "0"
            os.environ["FEMSHELL_AMG_DENSE_F32"] = "0"
            fs.assemble()
            fs.set_preconditioner("amg")
            _, i64 = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
            _, i64b = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
            del os.environ["FEMSHELL_AMG_SMOOTH_F32"], os.environ["FEMSHELL_AMG_DENSE_F32"]
            tts["all_fp64_preconditioner"] = {"iterations": i64["iterations"], "converged": i64["converged"],
                                              "solve_seconds": i64b["solve_seconds"], "pc_setup_seconds": i64["pc_setup_seconds"],
                                              "algorithmic_gb_per_iteration": i64["bytes_per_iteration"] / 1e9,
                                              "error_estimate": i64["error_estimate"]}
            # the Galerkin product on the matrix cores, the measured alternative to the default vector-ALU kernel: one more setup
            os.environ["FEMSHELL_AMG_GALERKIN"] = "mfma"
            fs.assemble()
            fs.set_preconditioner("amg")
            _, iq = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
            sq = fs.amg_setup_stats()
            del os.environ["FEMSHELL_AMG_GALERKIN"]
            tts["setup_first_coarsening_on_device"]["galerkin_on_matrix_cores_alternative"] = {
                "kernel": "k_amg_galerkin_mfma (v_mfma_f64_16x16x4_f64, one wave per coarse row; FEMSHELL_AMG_GALERKIN=mfma)",
                "galerkin_ms": sq["galerkin_ms"], "mfma_gflop_issued": sq["galerkin_mfma_flops_issued"] / 1e9,
                "mfma_tflops_issued": sq["galerkin_mfma_flops_issued"] / max(sq["galerkin_ms"], 1e-9) / 1e9,
                "frac_of_78.6_tflops": sq["galerkin_mfma_flops_issued"] / max(sq["galerkin_ms"], 1e-9) / 1e9 / 78.6,
                "iterations_with_it": iq["iterations"],
                "note": "6-row panels leave 10 of 16 tile rows idle and the operands are gathered 8 bytes at a time: slower than the "
                        "vector-ALU kernel, hence not the default"}
            fs.assemble()  # the hierarchy of the default kernels again for what follows
            fs.set_preconditioner("amg")
        if tts["block_jacobi_alone"] and "extrapolated_iterations_to_1e-10" in tts["block_jacobi_alone"]:
            tts["block_jacobi_alone"]["extrapolated_seconds"] = tts["block_jacobi_alone"]["extrapolated_iterations_to_1e-10"] * t_cg / max(info["iterations"], 1)

    symmetric = os.environ.get("FEMSHELL_SYMMETRIC", "1") != "0"
    spmv_kernel = "k_spmv_sym" if symmetric else "k_spmv"
    asm_kernel = fs.assembly_kernel()
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": n_elem * args.steps / t_asm,
            "unit": "elements/s",
            "cg_iters_per_s": info["iterations"] / t_cg,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_asm / args.steps,
            "ms_per_step_async": 1e3 * t_asm_async / args.steps,
            "elements_per_s_async": n_elem * args.steps / t_asm_async,
            "cg_ms_per_iter": 1e3 * t_cg / max(info["iterations"], 1),
            "ms_first_assembly_cold": cold_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": {"panel": "configs[3] flat panel", "cylinder": "configs[2] pinched cylinder",
                                    "roof": "configs[1] Scordelis-Lo roof"}[args.workload]
                                   + " %dx%d squares: %d tri3, %d dofs" % (args.nx, args.nx, n_elem, 6 * n_nodes),
                       "workload_parameters": {"panel": "10x10, simply supported, uniform pressure 300, E=1e7 nu=0.3 t=0.5 (configs[2] has the same size)",
                                               "cylinder": "R=300 L=600 t=3, E=3e6 nu=0.3",
                                               "roof": "R=25 L=50 80deg t=0.25, E=4.32e8 nu=0"}[args.workload],
                       "nodes": n_nodes,
                       "parallelism": "row-partition x%d" % world, "cg_iters_per_step": args.cg_iters,
                       "assembly_step": "one femshell_assemble: launch + status round trip",
                       "warmup_step": "one femshell_assemble + cg_iters CG iterations",
                       "untimed_before_the_timed_steps": "1 cold assembly (ms_first_assembly_cold), then the --warmup steps"
                                                         + (" + 30 assemblies + 200 CG iterations (FEMSHELL_BENCH_EXTRA_WARMUP=1)"
                                                            if os.environ.get("FEMSHELL_BENCH_EXTRA_WARMUP") == "1" else ""),
                       "preconditioner": "6x6 block-Jacobi (cg_iters_per_s)",
                       "time_to_solution_preconditioner": "SA multigrid, K cycle, mixed precision",
                       "symbolic_setup_s": setup_s,
                       "matrix_storage": "symmetric (upper triangles of the diagonal blocks + the blocks of the lower-numbered row)" if symmetric else "full",
                       "rccl_ranks_seen": rccl_ranks, "rccl_selftest_us": fs.comm_selftest(), "box_streaming_copy_gb_per_s": copy_gbs,
                       "box_stream_write_gb_per_s": stream.get("write_gb_per_s") if stream else None,
                       "box_stream_read_gb_per_s": stream.get("read_gb_per_s") if stream else None,
                       "box_stream_copy_gb_per_s": stream.get("copy_gb_per_s") if stream else None,
                       "box_hipmalloc_plus_free_of_4GiB_ms": alloc_ms,
                       "kernel_source_digest": kernel_source_digest()},
            # `roofline` belongs to `value`: the kernel the timed assembly steps consist of
            "roofline": dict(roof(asm_ms, asm_bytes, asm_kernel), kernel=asm_kernel,
                             kernel_note="element records -> block slots -> K and F; the kernel `value` / `ms_per_step` time; "
                                         "not HBM-bound alone: see fp64_* and DESIGN.md section 4"),
            "roofline_cg_spmv": dict(roof(spmv_ms, spmv_bytes, spmv_kernel), kernel=spmv_kernel + " (q = K p, fused p.q" +
                                     ("; symmetric storage: first phase, the update kernel collects the transposed products)" if symmetric else ")")),
            "roofline_cg_update": dict(roof(upd_ms, upd_bytes, "k_cg_update"), kernel="k_cg_update"),
            "roofline_cg_direction": dict(roof(dir_ms, dir_bytes, "k_cg_direction"), kernel="k_cg_direction"),
            "roofline_cg_iteration": roof(1e3 * t_cg / max(info["iterations"], 1), info["bytes_per_iteration"]),
        }
        if tts is not None:
            out["time_to_solution"] = tts
        if world == 1 and not args.profile:
            out["parity"] = {"small": parity_small(pkg, local_rank)}
            if not args.no_full_parity:
                out["parity"]["config1_scordelis_lo_250k"] = parity_config1(pkg, local_rank)
            if not args.no_fullsize_parity and args.workload == "panel" and args.nx == 1414:
                out["parity"]["config3_flat_panel_4M"] = fullsize_parity(fs, m, (nu, E, thick), "panel")
                out["parity"]["headline_load_case_vs_plate_theory"] = headline_load_case_witness()
                out["config2_pinched_cylinder_4M"] = config2_cylinder(pkg, local_rank, args.steps, args.warmup, 1414, roof)
                out["config4_coupled_flap_1M"] = config4_coupled_flap()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.nx if args.workload == "panel" else 1414)
        emit(out)
    fs.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
