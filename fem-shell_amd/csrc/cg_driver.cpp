// cg_driver.cpp -- host side of the conjugate-gradient solve: halo exchange beside the SpMV, the reduction /
// all-reduce / scalar-step sequence, and the two recurrences (classic and single-reduction).  Replaces what
// equation_systems.solve() hands to PETSc's KSPSolve (fem-shell.cpp:138, fem-shell_precice.cpp:271).
#include "context.hpp"

#include <cstddef>
#include <cstdlib>
#include "trace.hpp"

namespace femshell {

int halo_exchange(femshell_ctx *c, double *p, hipStream_t st)
{
    if (!c->comm.active() || c->comm.world == 1) return FEMSHELL_OK; // (a rank without neighbours still joins the group)
    TraceRange trace("femshell halo exchange");
    const Plan &pl = c->plan;
    for (size_t i = 0; i < pl.peers.size(); i++)
        launch_pack(p, c->send_nodes.p + c->send_offsets[i], (int32_t)pl.peers[i].send_nodes.size(),
                    c->sendbuf.p + 6ll * c->send_offsets[i], st);
    std::string e;
    if (!comm_halo(c->comm, pl.peers, c->send_offsets, c->sendbuf.p, p + 6ll * pl.n_pad, st, &e))
        return set_err(FEMSHELL_ERR_COMM, e);
    return FEMSHELL_OK;
}

// q = K p with the fused p.q partial sums.  Multi-rank contexts: the ghost entries of p travel on
// halo_stream (pack, grouped send/recv) while the main stream multiplies the slices that read owned
// columns only; the slices with ghost columns follow once the halo has landed.  Returns the number of
// partial sums written through *n_partials (0 = slice_grid).
// (xin: input vector with ghost space, yout = K xin, partial sums of xin.yout from partials[0] on)
// (defer_gather: symmetric storage, the caller's next kernel collects the transposed products -- k_cg_update<true>)
int spmv_with_halo(femshell_ctx *c, const CgVectors &v, double *xin, double *yout, double *partials, int *n_partials,
                   bool defer_gather, const float *vals32, int vec32)
{
    hipStream_t st = c->stream;
    *n_partials = 0;
    defer_gather = defer_gather && c->dm.symmetric;
    DeviceMatrix dm = c->dm;
    dm.vals32 = (defer_gather && c->dm.symmetric) ? vals32 : nullptr; // (the single-precision copy serves the symmetric first phase)
    const bool lowp = dm.vals32 != nullptr;
    dm.vec32 = lowp ? vec32 : 0; // (what such a product keeps in single precision besides: DeviceMatrix::vec32)
    if (!c->halo_overlap) {
        int rc = halo_exchange(c, xin, st);
        if (rc) return rc;
        if (defer_gather) launch_spmv_direct(dm, xin, yout, partials, v.s, st, lowp);
        else launch_spmv(c->dm, xin, yout, partials, v.s, st);
        return FEMSHELL_OK;
    }
    const Plan &pl = c->plan;
    FS_HIP(hipEventRecord(c->ev_p_ready, st));
    FS_HIP(hipStreamWaitEvent(c->halo_stream, c->ev_p_ready, 0));
    int rc = halo_exchange(c, xin, c->halo_stream);
    if (rc) return rc;
    FS_HIP(hipEventRecord(c->ev_halo_done, c->halo_stream));
    const int ni = pl.n_interior_slices, nb = pl.n_slices - ni;
    const int gi = launch_spmv_span(dm, xin, yout, partials, v.s, c->spmv_order.p, 0, ni, 0, st);
    FS_HIP(hipStreamWaitEvent(st, c->ev_halo_done, 0));
    const int gb = launch_spmv_span(dm, xin, yout, partials, v.s, c->spmv_order.p, ni, nb, gi, st);
    if (c->dm.symmetric && !defer_gather) launch_sym_gather(c->dm, yout, nullptr, 1.0, v.s, st); // all transposed products are in place
    *n_partials = gi + gb;
    return FEMSHELL_OK;
}

// (len3: length of the third partial array when nsums == 3)
// (phase NONE: the sums are reduced -- and all-reduced -- into red[] and no scalar step is taken: the caller's next
//  kernel does it; gate_phase then names the step the sums belong to)
int scalar_step(femshell_ctx *c, const CgVectors &v, int nsums, CgPhase phase, double rtol, int n_partials, int len3,
                int gate_phase)
{
    const int gate = gate_phase >= 0 ? gate_phase : (int)phase;
    if (c->comm.active()) {
        launch_cg_scalar(c->dm, v, true, nsums, CG_PHASE_NONE, rtol, c->stream, n_partials, len3, gate);
        std::string e;
        double *red = reinterpret_cast<double *>(reinterpret_cast<char *>(v.s) + offsetof(CgScalars, red));
        if (!comm_allreduce_sum(c->comm, red, nsums, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
        if (phase != CG_PHASE_NONE) launch_cg_scalar(c->dm, v, false, nsums, phase, rtol, c->stream);
    } else {
        launch_cg_scalar(c->dm, v, true, nsums, phase, rtol, c->stream, n_partials, len3, gate);
    }
    return FEMSHELL_OK;
}

// host side of the stopping test: the done flag is fetched at an 8 -> 64 iteration cadence
struct DonePoll {
    int32_t next_check = 8, check_step = 8;
    // returns 1 when the solve has finished (every further kernel would be a no-op), 0 to go on, < 0 on error
    int operator()(femshell_ctx *c, const CgVectors &v, int32_t it, int32_t max_it, CgScalars *hs)
    {
        if (it + 1 != next_check || it + 1 >= max_it) return 0;
        FS_HIP(hipMemcpyAsync(hs, v.s, sizeof *hs, hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        CommWatch::heartbeat(); // (progress: the watchdog of multi-rank contexts counts from here again)
        if (hs->done != 0) return 1;
        if (check_step < 64) check_step *= 2;
        next_check += check_step;
        return 0;
    }
};

// classic preconditioned CG: two reductions per iteration (p.q before the update, r.z and r.r after it); the
// iterates are those of the oracle's fso_pcg_block_jacobi
int cg_classic(femshell_ctx *c, const CgVectors &v, double rtol, int32_t max_it, const double *x0)
{
    const DeviceMatrix &m = c->dm;
    hipStream_t st = c->stream;
    launch_cg_init(m, v, false, st);
    int rc = scalar_step(c, v, 2, CG_PHASE_INIT, rtol);
    if (rc) return rc;
    if (x0 != nullptr) {
        // a solve from an initial guess (femshell_set_initial_guess): b.b and the threshold stand as derived from b above; the
        // method starts from x0 with the explicit residual r = b - K x0 (q = K x through the product's own input vector p)
        FS_HIP(hipMemcpyAsync(v.x, x0, (size_t)m.n_pad * 6 * sizeof(double), hipMemcpyDeviceToDevice, st));
        launch_copy_x_to_p(m, v, st);
        rc = halo_exchange(c, v.p, st);
        if (rc) return rc;
        launch_spmv(m, v.p, v.q, nullptr, nullptr, st);
        launch_cg_init(m, v, true, st);
        rc = scalar_step(c, v, 2, CG_PHASE_RESTART, rtol);
        if (rc) return rc;
    }
    CgScalars hs{};
    DonePoll poll;
    for (int32_t it = 0; it < max_it; it++) {
        int n_partials = 0;
        rc = spmv_with_halo(c, v, v.p, v.q, v.partials, &n_partials, true);
        if (rc) return rc;
        rc = scalar_step(c, v, 1, CG_PHASE_ALPHA, rtol, n_partials);
        if (rc) return rc;
        launch_cg_update(m, v, st, m.symmetric != 0);
        rc = scalar_step(c, v, 2, CG_PHASE_BETA, rtol);
        if (rc) return rc;
        launch_cg_direction(m, v, st);
        rc = poll(c, v, it, max_it, &hs);
        if (rc < 0) return rc;
        if (rc == 1) break;
    }
    return FEMSHELL_OK;
}

// single-reduction preconditioned CG (Chronopoulos & Gear, SIAM J. Sci. Stat. Comput. 1989): the same Krylov
// iterates in exact arithmetic, with s = A p carried by recurrence so that r.z, r.r and z.Az are reduced together --
// one all-reduce of three doubles and one vector kernel per iteration (multi-rank solves)
int cg_single_reduction(femshell_ctx *c, const CgVectors &v, double rtol, int32_t max_it)
{
    const DeviceMatrix &m = c->dm;
    hipStream_t st = c->stream;
    const int G = slice_grid(m);
    double *spmv_partials = v.partials + 2 * (size_t)G; // third partial array
    // FEMSHELL_CG_FOLD=0: the unfolded sequence (k_sym_gather pass, scalar step as a launch of its own) for A/B runs
    const char *fold_env = getenv("FEMSHELL_CG_FOLD");
    const bool fold = !(fold_env && atoi(fold_env) == 0);
    const bool gather = fold && m.symmetric != 0; // the update kernel collects the transposed products of w = A z
    launch_cgcg_init(m, v, st);
    int len3 = 0;
    int rc = spmv_with_halo(c, v, v.z, v.q, spmv_partials, &len3, gather);
    if (rc) return rc;
    rc = scalar_step(c, v, 3, CG_PHASE_FUSED_INIT, rtol, G, len3 > 0 ? len3 : G);
    if (rc) return rc;
    CgScalars hs{};
    DonePoll poll;
    // One iteration on the main stream: update (with the scalar step of the previous iteration and the collection of
    // the transposed products folded in), interior product, boundary product, reduction of the partial sums, all-reduce
    // of three doubles -- four kernels and one collective; pack and send/recv run beside the interior product.
    bool open_step = false; // an all-reduce whose scalar step has not been taken yet
    for (int32_t it = 0; it < max_it; it++) {
        launch_cgcg_update(m, v, st, fold && it > 0 ? (int)((it - 1) & 1) : -1, gather);
        rc = spmv_with_halo(c, v, v.z, v.q, spmv_partials, &len3, gather);
        if (rc) return rc;
        if (fold) {
            rc = scalar_step(c, v, 3, CG_PHASE_NONE, rtol, G, len3 > 0 ? len3 : G, (int)CG_PHASE_FUSED_STEP);
            open_step = true;
        } else {
            rc = scalar_step(c, v, 3, CG_PHASE_FUSED_STEP, rtol, G, len3 > 0 ? len3 : G);
        }
        if (rc) return rc;
        rc = poll(c, v, it, max_it, &hs);
        if (rc < 0) return rc;
        if (rc == 1) {
            open_step = false; // done was set by the step inside an update kernel; the sums reduced since belong to no-ops
            break;
        }
    }
    if (open_step) launch_cg_scalar(m, v, false, 3, CG_PHASE_FUSED_STEP, rtol, st); // closes the last iteration
    return FEMSHELL_OK;
}

// multi-rank contexts use the single-reduction recurrence; FEMSHELL_CG_SINGLE_REDUCTION=0/1 overrides
bool use_single_reduction(const femshell_ctx *c)
{
    const char *e = getenv("FEMSHELL_CG_SINGLE_REDUCTION");
    if (e) return atoi(e) != 0;
    return c->comm.active();
}

} // namespace femshell
