"""N>1 path on CPU: two (and three) gloo ranks run the row-partitioned block-Jacobi CG with
exactly the communication pattern of the device driver (cg_driver.cpp): halo exchange of the SpMV input
following the plan's peer lists before every SpMV, and either the classic recurrence (one all-reduce
for p.Ap, one for (r.z, r.r)) or the single-reduction recurrence multi-rank device solves default to
(one all-reduce of (r.z, r.r, z.Az)); the local matrices come from each rank's plan (gather lists)
with the oracle's element arithmetic.  The partitioned solve must reproduce the single-process
oracle solve."""
import os
import sys
import tempfile

import numpy as np
import pytest

from tests.helpers import meshes, oracle, sell
from tests.helpers.product import ROOT, ensure_built

pkg = ensure_built()


def problem():
    m = meshes.structured(13, 21, 0, 0, 2, 3, kind="t", ul_lr=True, bcids=(0, -1, 1, -1), factor=5.0, loading=2)
    m.xyz[:, 2] = 0.05 * np.sin(2.0 * m.xyz[:, 0]) * m.xyz[:, 1]
    return m, (0.3, 3.0e4, 0.1)


def _rank_main(rank, world, init_file, out_dir, single_reduction):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    m, (nu, E, t) = problem()
    mat = oracle.material(nu, E, t)
    dmask = m.dirichlet_mask()
    plan = pkg.build_plan(m.xyz, m.tri, m.quad, rank=rank, world_size=world)
    n_own, n_pad, n_ghost = plan["n_own"], plan["n_pad"], plan["n_ghost"]
    dm = np.zeros(n_pad + n_ghost, dtype=np.uint8)
    dm[:n_own] = dmask[plan["row_begin"]:plan["row_end"]]
    dm[n_pad:] = dmask[plan["ghost_global"]]
    blocks = sell.assemble_from_plan(plan, mat, oracle, dm)
    # local operator on the extended vector [owned | padding | ghosts]
    rows, cols, vals = [], [], []
    for _, (r, c, blk) in blocks.items():
        rows.append(r)
        cols.append(c)
        vals.append(blk)
    rows, cols, vals = np.array(rows), np.array(cols), np.array(vals)

    def spmv(x_ext):
        y = np.zeros(6 * n_own)
        xb = x_ext.reshape(-1, 6)
        contrib = np.einsum("bij,bj->bi", vals, xb[cols])
        np.add.at(y.reshape(-1, 6), rows, contrib)
        return y

    diag = {r: blk for _, (r, c, blk) in blocks.items() if r == c}
    minv = np.stack([np.linalg.inv(diag[a]) for a in range(n_own)])
    b = np.where((dm[:n_own, None] >> np.arange(6)) & 1, 0.0, m.loads[plan["row_begin"]:plan["row_end"]]).ravel()

    def halo(p_ext):
        reqs = []
        recv_bufs = []
        for i, q in enumerate(plan["peer_ranks"]):
            s0, s1 = plan["peer_send_ptr"][i], plan["peer_send_ptr"][i + 1]
            if s1 > s0:
                snd = torch.from_numpy(p_ext.reshape(-1, 6)[plan["peer_send_nodes"][s0:s1]].copy())
                reqs.append(dist.isend(snd, int(q)))
            cnt = int(plan["peer_recv_count"][i])
            if cnt:
                buf = torch.zeros(cnt, 6, dtype=torch.float64)
                recv_bufs.append((int(plan["peer_recv_offset"][i]), cnt, buf))
                reqs.append(dist.irecv(buf, int(q)))
        for r in reqs:
            r.wait()
        for off, cnt, buf in recv_bufs:
            p_ext.reshape(-1, 6)[n_pad + off:n_pad + off + cnt] = buf.numpy()

    def allsum(*vals_):
        tt = torch.tensor(vals_, dtype=torch.float64)
        dist.all_reduce(tt)
        return tt.tolist()

    def precond(v):
        return np.einsum("aij,aj->ai", minv, v.reshape(-1, 6)).ravel()

    x = np.zeros(6 * n_own)
    r = b.copy()
    its = 0
    if single_reduction:
        # cg_driver.cpp cg_single_reduction: z carries the ghost entries, s = A p by recurrence
        z_ext = np.zeros(6 * (n_pad + n_ghost))
        z_ext[:6 * n_own] = precond(r)
        p = np.zeros(6 * n_own)
        s = np.zeros(6 * n_own)
        halo(z_ext)
        w = spmv(z_ext)
        rz, bb, zaz = allsum(r @ z_ext[:6 * n_own], b @ b, z_ext[:6 * n_own] @ w)
        alpha, beta = rz / zaz, 0.0
        for its in range(1, 20001):
            p = z_ext[:6 * n_own] + beta * p
            s = w + beta * s
            x += alpha * p
            r -= alpha * s
            z_ext[:6 * n_own] = precond(r)
            halo(z_ext)
            w = spmv(z_ext)
            rz_new, rr, zaz = allsum(r @ z_ext[:6 * n_own], r @ r, z_ext[:6 * n_own] @ w)
            if rr <= 1e-24 * bb:
                break
            beta = rz_new / rz
            alpha = rz_new / (zaz - beta * rz_new / alpha)
            rz = rz_new
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, begin=plan["row_begin"], end=plan["row_end"], its=its)
        dist.barrier()
        dist.destroy_process_group()
        return
    z = precond(r)
    p_ext = np.zeros(6 * (n_pad + n_ghost))
    p_ext[:6 * n_own] = z
    rz, bb = allsum(r @ z, b @ b)
    for its in range(1, 20001):
        halo(p_ext)
        q = spmv(p_ext)
        (pq,) = allsum(p_ext[:6 * n_own] @ q)
        alpha = rz / pq
        x += alpha * p_ext[:6 * n_own]
        r -= alpha * q
        z = np.einsum("aij,aj->ai", minv, r.reshape(-1, 6)).ravel()
        rz_new, rr = allsum(r @ z, r @ r)
        if rr <= 1e-24 * bb:
            break
        p_ext[:6 * n_own] = z + (rz_new / rz) * p_ext[:6 * n_own]
        rz = rz_new
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, begin=plan["row_begin"], end=plan["row_end"], its=its)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,single_reduction", [(2, False), (3, False), (2, True), (3, True)])
def test_row_partitioned_cg_matches_single_process(world, single_reduction):
    import torch.multiprocessing as mp

    m, (nu, E, t) = problem()
    mat = oracle.material(nu, E, t)
    r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
    u_ref = oracle.direct_solve(r0, c0, v0, F0)
    _, info = oracle.pcg(r0, c0, v0, F0, rtol=1e-12, max_it=20000)
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rendezvous")
        mp.spawn(_rank_main, args=(world, init_file, d, single_reduction), nprocs=world, join=True)
        u = np.zeros(6 * m.n_nodes)
        its = []
        for r in range(world):
            z = np.load(os.path.join(d, "rank%d.npz" % r))
            u[6 * int(z["begin"]):6 * int(z["end"])] = z["x"]
            its.append(int(z["its"]))
    assert len(set(its)) == 1  # every rank takes the same decisions
    # the classic recurrence follows the single-process iteration count; the single-reduction recurrence is the more
    # delicate one in finite precision (Chronopoulos & Gear), and with symmetric storage the ranks' lower blocks are
    # exact transposes where the oracle computes them independently (1e-16 apart): near the end of an ill-conditioned
    # solve that moves its iteration count by tens of percent, not its answer
    assert abs(its[0] - info["iterations"]) <= (0.25 * info["iterations"] if single_reduction else 3)
    assert np.linalg.norm(u - u_ref) <= 1e-9 * np.linalg.norm(u_ref)


def test_bench_cpu_baseline_worker_runs_on_a_small_panel():
    """bench.py's cpu_baseline leg runs in a child process with bound OpenMP threads (OMP_PROC_BIND / OMP_PLACES have to be
    in the environment before the runtime starts); the child is exercised here on a 60 x 60 panel: thread sweep, STREAM
    triad, first touch inside the threaded loops -- and the threaded assembly gives the serial checker's matrix."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_PROC_BIND="spread", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-baseline-worker", "--nx", "60"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["kind"] == "port" and out["unit"] == "elements/s" and out["value"] > 0 and out["cg_iters_per_s"] > 0
    assert out["omp"]["OMP_PROC_BIND"] == "spread" and out["host_stream_triad_gb_per_s"] > 0
    assert any(s["threads"] == 1 for s in out["thread_sweep"]) and out["single_thread"]["value"] > 0
    assert "7200 tri3" in out["sample"]


def test_threaded_baseline_assembly_equals_the_serial_checker():
    """The CPU-baseline build of the oracle (-O3 -march=native -fopenmp) assembles by node ranges with per-thread element
    lists (oracle/femshell_oracle.c ownership_lists): on 1, 3 and 8 threads it gives the matrix of the serial checker
    build, also after the mesh behind the same pointers changed (the lists are keyed on a hash of the connectivity)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from tests.helpers import meshes, oracle
mat = oracle.material(0.3, 1e7, 0.5)
ms = [meshes.structured(40, 30, 0, 0, 10, 9, kind="t", ul_lr=u, bcids=(0, 1, 0, 1), factor=300.0, loading=2) for u in (True, False)]
ref = [oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads) for m in ms]
oracle.use_fast_build(1)
tri = np.array(ms[0].tri, dtype=np.int32)                  # one buffer for both meshes: same pointer, other connectivity
for th in (1, 3, 8, 3):
    oracle.set_threads(th)
    for m, (r0, c0, v0, F0) in zip(ms, ref):
        tri[:] = m.tri
        r1, c1, v1, F1 = oracle.assemble(m.xyz, tri, m.quad, mat, m.dirichlet_mask(), m.loads)
        assert np.array_equal(r1, r0) and np.array_equal(c1, c0) and np.array_equal(F1, F0)
        assert np.abs(v1 - v0).max() <= 1e-15 * np.abs(v0).max(), (th, np.abs(v1 - v0).max())
print("ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
