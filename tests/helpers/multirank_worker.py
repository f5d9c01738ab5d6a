"""One rank of a multi-rank solve on a shared GPU (tests/test_multirank_gpu.py launches several of these
with FEMSHELL_RCCL_LIB pointing at the fake RCCL).  argv: rank world uid_file out_file mesh_kind"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import meshes  # noqa: E402


def build_problem(kind):
    if kind == "panel":
        m = meshes.structured(40, 56, 0, 0, 4, 5.6, kind="t", ul_lr=True, bcids=(0, 0, 1, -1), factor=3.0, loading=2)
        m.xyz[:, 2] = 0.2 * np.sin(1.1 * m.xyz[:, 0]) * np.cos(0.6 * m.xyz[:, 1])
        return m, (0.3, 2.0e5, 0.05)
    if kind == "panel_bad":
        # one zero-area triangle among the last rows: only the rank that owns it sees the failure locally
        m, mat = build_problem("panel")
        a, b, c = m.tri[-3]
        m.xyz[c] = m.xyz[b]  # coincident nodes: exactly zero area
        return m, mat
    if kind == "cylinder":
        m = meshes.pinched_cylinder(48, 40)
        return m, m.material
    if kind in ("jittered_random", "delaunay_hard_random"):
        # the same shells in the numbering the generator gives them -- random: a rank's range of the caller's ids is scattered
        # over the whole shell (half of all nodes are ghosts of the other rank).  For FEMSHELL_REORDER=morton|rcm, which
        # partitions the renumbered rows.
        from tests.test_gpu_parity import delaunay_shell

        class M:
            pass
        m = M()
        xyz, tri = delaunay_shell(20000, 3, jittered=(kind == "jittered_random"))
        m.xyz, m.tri, m.quad = np.ascontiguousarray(xyz), tri.astype(np.int32), None
        mask = np.zeros(len(m.xyz), dtype=np.uint8)
        mask[m.xyz[:, 0] < 0.15] = 0x3F
        m.loads = np.zeros((len(m.xyz), 6))
        m.loads[:, 2] = 1.0
        m.dirichlet_mask = lambda: mask
        m.n_nodes = len(m.xyz)
        return m, (0.3, 7.0e4, 0.03)
    if kind == "jittered":
        # an unstructured mesh of good element quality (Delaunay triangulation of a jittered grid on a curved shell: irregular
        # valence, no slivers), numbered along x as a mesh prepared for several ranks would be
        from tests.test_gpu_parity import delaunay_shell

        class M:
            pass
        m = M()
        xyz, tri = delaunay_shell(4000, 3, jittered=True)
        order = np.argsort(xyz[:, 0], kind="stable")
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        m.xyz, m.tri = np.ascontiguousarray(xyz[order]), inv[tri].astype(np.int32)
        m.quad = None
        mask = np.zeros(len(m.xyz), dtype=np.uint8)
        mask[m.xyz[:, 0] < 0.15] = 0x3F
        m.loads = np.zeros((len(m.xyz), 6))
        m.loads[:, 2] = 1.0
        m.dirichlet_mask = lambda: mask
        return m, (0.3, 7.0e4, 0.03)
    if kind == "delaunay_hard":
        # the Delaunay shell of poor element quality of tests/test_gpu_parity.py: the flexible CG breaks down under the
        # single-precision copies of the multigrid hierarchy
        from tests.test_gpu_parity import delaunay_shell

        class M:
            pass
        m = M()
        xyz, tri = delaunay_shell(20000, 3)
        # (numbered along x, as a mesh that was partitioned for several ranks would be: with the generator's random numbering
        #  half of all nodes are ghosts of the other rank and the rows of A P exchanged at setup exceed the 64 MB message slot
        #  of the test transport)
        order = np.argsort(xyz[:, 0], kind="stable")
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        m.xyz, m.tri = np.ascontiguousarray(xyz[order]), inv[tri].astype(np.int32)
        m.quad = None
        mask = np.zeros(len(m.xyz), dtype=np.uint8)
        mask[m.xyz[:, 0] < 0.15] = 0x3F
        m.loads = np.zeros((len(m.xyz), 6))
        m.loads[:, 2] = 1.0
        m.dirichlet_mask = lambda: mask
        return m, (0.3, 7.0e4, 0.03)
    raise SystemExit("unknown mesh kind")


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    uid_file, out_file, kind = sys.argv[3], sys.argv[4], sys.argv[5]
    pkg = importlib.import_module("fem-shell_amd")
    m, (nu, E, t) = build_problem(kind)
    fs = pkg.FemShell(nu, E, t, device=0, rank=rank, world_size=world)
    if world > 1:
        if rank == 0:
            uid = pkg.comm_unique_id()
            np.save(uid_file + ".tmp.npy", uid)
            os.replace(uid_file + ".tmp.npy", uid_file)
        else:
            t0 = time.time()
            while not os.path.exists(uid_file):
                if time.time() - t0 > 60:
                    raise SystemExit("timeout waiting for the unique id")
                time.sleep(0.01)
            uid = np.load(uid_file)
        fs.comm_init(uid)
        # first contact (femshell_comm_init): the grouped send/recv ring beside an all-reduce, the grouped broadcasts and a lone
        # all-reduce came back with their known answers on every rank, or comm_init would have raised
        st = fs.comm_selftest()
        assert st is not None and all(v > 0.0 for v in st.values()), st
    fs.set_mesh(m.xyz, m.tri, m.quad)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    pc = os.environ.get("FEMSHELL_TEST_PC", "")
    if pc == "amg":
        fs.set_preconditioner("amg", coarsest_nodes=60)
    if os.environ.get("FEMSHELL_TEST_ASYNC") == "1":
        fs.assemble(wait=False)  # twice: one status word collects both
        fs.assemble(wait=False)
    if kind == "panel_bad":
        # every rank must come back with an error (none may hang in a collective of the CG loop)
        try:
            fs.solve(rtol=1e-11, max_it=1000)
            code, msg = 0, ""
        except pkg.FemShellError as ex:
            code, msg = ex.code, str(ex)
        np.savez(out_file, code=code, msg=msg)
        fs.close()
        return
    if kind == "delaunay_hard_random":
        # the multigrid setup alone (one iteration): what it sends, and which assembly kernel the renumbered mesh gets
        fs.set_preconditioner("amg")
        fs.comm_bytes(clear=True)
        fs.assemble()
        _, info = fs.solve(rtol=1e-10, max_it=1)
        np.savez(out_file, setup_bytes=np.array(fs.comm_bytes()), assembly_kernel=fs.assembly_kernel(), levels=info["amg_levels"],
                 own=fs.owned_nodes())
        fs.close()
        return
    if kind == "delaunay_hard":
        # FEMSHELL_AMG_PATCH_TAU=0 (the test sets it): every rank takes the fallback together (the rebuild of the hierarchy is
        # collective) and ends at the iteration limit; with the patch smoother (default) the solve converges (FEMSHELL_TEST_MAX_IT)
        fs.set_preconditioner("amg")
        u, info = fs.solve(rtol=1e-10, max_it=int(os.environ.get("FEMSHELL_TEST_MAX_IT", "120")))
        np.savez(out_file, fallback=info["pc_fp64_fallback"], iterations=info["iterations"], converged=info["converged"],
                 finite=bool(np.all(np.isfinite(u))), levels=info["amg_levels"], u=u, patch=np.array(list(fs.amg_patch_info().values()), dtype=np.float64))
        fs.close()
        return
    if world > 1:
        fs.comm_bytes(clear=True)
    u, info = fs.solve(rtol=1e-11, max_it=100000)
    b, e = fs.row_range()
    extra = {"own": fs.owned_nodes(), "assembly_kernel": fs.assembly_kernel()}
    if world > 1:
        extra["first_solve_bytes"] = np.array(fs.comm_bytes())
    if os.environ.get("FEMSHELL_TEST_EXPORT") == "1":
        # the rows of K and F this rank assembled (global column ids), for the comparison with the oracle's assembly
        rp, ci, vals, F = fs.export_bsr()
        extra.update(k_rowptr=rp[:e - b + 1], k_cols=ci, k_vals=vals, k_F=F[:6 * (e - b)])
    if os.environ.get("FEMSHELL_TEST_AMG_EXPORT") == "1":
        # the rank's part of the multigrid hierarchy (row-partitioned levels: its rows, global column ids)
        lv = fs.amg_levels()
        extra["amg_n_nodes"] = np.array([l["n_nodes"] for l in lv])
        extra["amg_lambda"] = np.array([l["lambda_max"] for l in lv])
        pi = fs.amg_partition_info() if world > 1 else {"partitioned_levels": 0, "bytes_partitioned": 0.0, "bytes_replicated": 0.0}
        extra["amg_partitioned_levels"] = pi["partitioned_levels"]
        extra["amg_bytes"] = np.array([pi["bytes_partitioned"], pi["bytes_replicated"]])
        for li in range(len(lv) - 1):
            for name, arr in fs.amg_export(li).items():
                if arr is not None:
                    extra["amg_L%d_%s" % (li, name)] = arr
        extra["residual_history"] = fs.residual_history()
    # a second solve on the same context with doubled loads (the coupled program re-solves every coupling iteration)
    fs.set_loads(2.0 * m.loads)
    if world > 1:
        fs.comm_counters(clear=True)  # (the second solve reuses the hierarchy: what it enqueues is the solve alone, no setup)
    u2, info2 = fs.solve(rtol=1e-11, max_it=100000)
    if world > 1:
        cc = fs.comm_counters()
        extra["comm_solve2"] = np.array([cc["halo_exchanges_on_the_halo_stream"], cc["halo_exchanges_on_the_main_stream"],
                                         cc["allreduces"], cc["row_gathers"]])
    if os.environ.get("FEMSHELL_TEST_WARM") == "1":
        # a third solve of the same loads from the second one's solution (femshell_set_initial_guess on a row-partitioned context)
        fs.set_initial_guess(None)
        u3, info3 = fs.solve(rtol=1e-11, max_it=100000)
        extra.update(u3=u3, iterations3=info3["iterations"], converged3=info3["converged"])
        fs.set_initial_guess(u)  # ... and one from a host vector: the first solve's solution, half the loads
        u4, info4 = fs.solve(rtol=1e-11, max_it=100000)
        extra.update(u4=u4, iterations4=info4["iterations"], converged4=info4["converged"])
    np.savez(out_file, u=u, iterations=info["iterations"], converged=info["converged"], begin=b, end=e,
             true_res=info["true_rel_residual"], u2=u2, converged2=info2["converged"], iterations2=info2["iterations"],
             levels=info["amg_levels"], **extra)
    fs.close()


if __name__ == "__main__":
    main()
