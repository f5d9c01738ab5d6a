"""Galerkin product of the first coarsening step on the 4M-triangle panel: matrix cores against vector ALUs (setup only).
usage: galerkin_ab.py [n=1414]"""
import importlib, os, subprocess, sys, json
sys.path.insert(0, ".")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from tests.helpers import fullsize
    pkg = importlib.import_module("fem-shell_amd")
    n = int(sys.argv[2])
    m, mat = fullsize.workload("panel", n)
    fs = pkg.FemShell(*mat, device=0)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    fs.set_preconditioner("amg")
    out = []
    for rep in range(2):
        fs.assemble()          # K changes -> the hierarchy is rebuilt
        _, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        st = fs.amg_setup_stats()
        out.append({"galerkin_ms": st["galerkin_ms"], "ap_ms": st["ap_ms"], "mfma": st["galerkin_on_matrix_cores"], "iterations": info["iterations"],
                    "setup_s": info["pc_setup_seconds"], "useful_gflop": st["galerkin_useful_flops"] / 1e9, "issued_gflop": st["galerkin_mfma_flops_issued"] / 1e9})
    print(json.dumps(out[-1]))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "1414"
    for mode in ("mfma", "valu", "mfma", "valu"):
        env = dict(os.environ, FEMSHELL_AMG_GALERKIN=mode)
        r = subprocess.run([sys.executable, __file__, "--child", n], env=env, capture_output=True, text=True)
        print(mode, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-500:], flush=True)
