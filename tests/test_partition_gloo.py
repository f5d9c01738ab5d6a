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
