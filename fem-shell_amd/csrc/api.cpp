// api.cpp -- the C ABI of libfemshell (include/femshell.h): context, host<->HBM plumbing, measurement hooks.
// The CG driver is cg_driver.cpp, the plan inspection entry points plan_api.cpp.  All arithmetic of the hot path
// runs in the kernels of kernels.hip; there is no CPU fallback anywhere in this library.
#include "amg_device.hpp"
#include "context.hpp"
#include "reorder.hpp"
#include "trace.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace femshell;

namespace femshell {

namespace {
thread_local std::string g_err;
}

int set_err(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

const std::string &last_err() { return g_err; }

} // namespace femshell

namespace {

int select_device(femshell_ctx *c)
{
    FS_HIP(hipSetDevice(c->device));
    return FEMSHELL_OK;
}

int64_t real_blocks(const femshell_ctx *c) { return c->plan.nnz_blocks; }

double bytes_assemble(const femshell_ctx *c)
{
    const Plan &p = c->plan;
    // (symmetric storage: only the stored blocks are computed and written -- the algorithmic bytes of that layout -- and of a
    //  diagonal block, symmetric itself, the 12 words of the upper triangle: 192 instead of 288 bytes per node)
    return 12.0 * p.n_ltri() + 16.0 * p.n_lquad() + 24.0 * (p.n_own + p.n_ghost) + 292.0 * (double)p.stored_blocks -
           (c->dm.diag_upper ? 96.0 * p.n_own : 0.0) + 4.0 * (p.n_own + 1) + 48.0 * p.n_own;
}
double bytes_spmv(const femshell_ctx *c)
{
    const Plan &p = c->plan;
    // (symmetric storage: every stored block is streamed once; the 48-byte transposed products written and read
    // beside them are overhead of the method, not algorithmic traffic)
    // ... and of a diagonal block the 12 words (192 bytes) that hold its upper triangle
    return 292.0 * (double)p.stored_blocks - (p.symmetric ? 96.0 * p.n_own : 0.0) + 4.0 * (p.n_own + 1) + 96.0 * p.n_own;
}
// (the inverse diagonal blocks are symmetric: 21 of their 36 words are stored and read)
double bytes_update(const femshell_ctx *c) { return (7.0 * 48.0 + 168.0) * c->plan.n_own; }
double bytes_direction(const femshell_ctx *c) { return 3.0 * 48.0 * c->plan.n_own; }
// single-reduction recurrence: z, w, p, s, x, r read, p, s, x, r, z written, Minv read
double bytes_update_single_reduction(const femshell_ctx *c) { return (11.0 * 48.0 + 168.0) * c->plan.n_own; }

int interpret_status(femshell_ctx *c, int32_t st, const char *what);

int check_status(femshell_ctx *c, const char *what)
{
    if (!c->status_mapped) FS_HIP(hipMemcpyAsync(c->status_host, c->status_word, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    return interpret_status(c, *(volatile int32_t *)c->status_host, what);
}

// the device status word of this rank as an error code + message (and the word cleared for the next launch)
int interpret_status(femshell_ctx *c, int32_t st, const char *what)
{
    if (st == 0) return FEMSHELL_OK;
    {
        const int rcc = clear_status_word(c, c->stream);
        if (rcc) return rcc;
    }
    char buf[160];
    if (st > 0) {
        const Plan &p = c->plan;
        // the kernel reports an index into the per-slice element lists
        const int32_t le = st >= kStatusDirect ? st - kStatusDirect
                           : ((st - 1 < (int32_t)p.slice_elems.size()) ? p.slice_elems[st - 1] : st - 1);
        if (le < p.n_ltri())
            snprintf(buf, sizeof buf, "%s: triangle %d is degenerate (zero area or zero-length first edge)", what,
                     p.tri_global_id[le]);
        else if (le - p.n_ltri() < p.n_lquad())
            snprintf(buf, sizeof buf, "%s: quad %d is degenerate or QUAD4 is unsupported by this kernel", what,
                     p.quad_global_id[le - p.n_ltri()]);
        else
            snprintf(buf, sizeof buf, "%s: element %d failed", what, le);
        return set_err(FEMSHELL_ERR_MESH, buf);
    }
    {
        const int32_t node = c->plan.row_begin + (-st - 1);
        snprintf(buf, sizeof buf, "%s: diagonal block of node %d is not positive definite", what,
                 (!c->perm.empty() && node >= 0 && node < (int32_t)c->perm.size()) ? c->perm[node] : node);
    }
    return set_err(FEMSHELL_ERR_BREAKDOWN, buf);
}

// Collective agreement on a rank-local outcome (multi-rank contexts): a degenerate element or a non-SPD diagonal
// block exists on the owning rank only; without this the healthy ranks would walk on into the halo exchange and
// the all-reduces of the CG loop and wait there forever.  Every rank calls this at the same points (after the
// assembly and after the block-Jacobi setup); all of them leave with an error when any of them has one.
// (neutral: the caller is neither the assembly nor the block-Jacobi setup -- the healthy ranks then report "failed on N
//  other rank(s)" with FEMSHELL_ERR_COMM instead of guessing at a degenerate element or a non-SPD block over there)
int agree_status(femshell_ctx *c, int local_rc, const char *what, bool neutral = false)
{
    if (!c->comm.active()) return local_rc;
    const std::string local_msg = last_err();
    FS_HIP(c->agree.alloc(2));
    // [ranks with a mesh error, ranks with any other error]: the healthy ranks leave with the same class of error
    c->agree_host[0] = local_rc == FEMSHELL_ERR_MESH ? 1.0 : 0.0;
    c->agree_host[1] = (local_rc && local_rc != FEMSHELL_ERR_MESH) ? 1.0 : 0.0;
    FS_HIP(hipMemcpyAsync(c->agree.p, c->agree_host, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    std::string e;
    if (!comm_allreduce_sum(c->comm, c->agree.p, 2, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    FS_HIP(hipMemcpyAsync(c->agree_host, c->agree.p, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    if (local_rc) return set_err(local_rc, local_msg);
    if (neutral && c->agree_host[0] + c->agree_host[1] > 0.0) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: failed on %d other rank(s) of the row partition (their own message says why)", what,
                 (int)(c->agree_host[0] + c->agree_host[1]));
        return set_err(FEMSHELL_ERR_COMM, buf);
    }
    if (c->agree_host[0] + c->agree_host[1] > 0.0) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: failed on %d other rank(s) of the row partition (%s there)", what,
                 (int)(c->agree_host[0] + c->agree_host[1]), c->agree_host[0] > 0.0 ? "degenerate element" : "non-SPD diagonal block");
        return set_err(c->agree_host[0] > 0.0 ? FEMSHELL_ERR_MESH : FEMSHELL_ERR_BREAKDOWN, buf);
    }
    return FEMSHELL_OK;
}

// After a kernel that reports through the device status word (assembly, block-Jacobi setup): the local check and the
// cross-rank agreement in ONE stream synchronisation -- a kernel turns the status word into the two agreement counters,
// the all-reduce runs on the device, status and counters come back together.  (check_status followed by agree_status is
// two synchronisations and three copies; on 8 ranks a 0.1 ms assembly step would spend as long agreeing as assembling.)
int check_and_agree(femshell_ctx *c, const char *what)
{
    if (!c->comm.active()) return check_status(c, what);
    FS_HIP(c->agree.alloc(2));
    launch_status_flags(c->status_word, c->agree.p, c->stream);
    std::string e;
    if (!comm_allreduce_sum(c->comm, c->agree.p, 2, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    if (!c->status_mapped) FS_HIP(hipMemcpyAsync(c->status_host, c->status_word, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipMemcpyAsync(c->agree_host, c->agree.p, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    const int local_rc = interpret_status(c, *c->status_host, what);
    if (local_rc) return local_rc;
    if (c->agree_host[0] + c->agree_host[1] > 0.0) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: failed on %d other rank(s) of the row partition (%s there)", what,
                 (int)(c->agree_host[0] + c->agree_host[1]), c->agree_host[0] > 0.0 ? "degenerate element" : "non-SPD diagonal block");
        return set_err(c->agree_host[0] > 0.0 ? FEMSHELL_ERR_MESH : FEMSHELL_ERR_BREAKDOWN, buf);
    }
    return FEMSHELL_OK;
}

// scatter the global per-node arrays into the rank-local numbering and upload
// the rank's part of the Dirichlet masks (kNodeMasks) and / or of the nodal loads (kNodeLoads) to HBM; kNodeCleared: both
// are known to be zero (a mesh was just set), no host data moves at all
enum : int { kNodeMasks = 1, kNodeLoads = 2, kNodeCleared = 4 };
int upload_node_data(femshell_ctx *c, int what)
{
    const Plan &p = c->plan;
    hipStream_t st = c->stream;
    if (what & kNodeCleared) {
        FS_HIP(c->dmask.alloc((size_t)p.n_local_nodes()));
        FS_HIP(c->loads.alloc((size_t)p.n_pad * 6));
        FS_HIP(c->dmask.zero(st));
        FS_HIP(c->loads.zero(st));
        what = kNodeMasks; // (the item flags below)
    } else {
        if (what & kNodeMasks) {
            std::vector<uint8_t> dm((size_t)p.n_local_nodes(), 0);
            for (int32_t a = 0; a < p.n_own; a++) dm[(size_t)a] = c->dmask_global[(size_t)(p.row_begin + a)];
            for (int32_t g = 0; g < p.n_ghost; g++) dm[(size_t)(p.n_pad + g)] = c->dmask_global[(size_t)p.ghost_global[(size_t)g]];
            FS_HIP(c->dmask.upload(dm, st));
            FS_HIP(hipStreamSynchronize(st)); // the host vector goes out of scope
        }
        if (what & kNodeLoads) {
            // the owned rows are one contiguous piece of the global array: no staging copy (96 MB at 4M triangles)
            FS_HIP(c->loads.alloc((size_t)p.n_pad * 6));
            if (p.n_own > 0)
                FS_HIP(hipMemcpyAsync(c->loads.p, c->loads_global.data() + 6ull * (size_t)p.row_begin, (size_t)p.n_own * 6 * sizeof(double),
                                      hipMemcpyHostToDevice, st));
            if (p.n_pad > p.n_own)
                FS_HIP(hipMemsetAsync(c->loads.p + 6ull * (size_t)p.n_own, 0, (size_t)(p.n_pad - p.n_own) * 6 * sizeof(double), st));
            FS_HIP(hipStreamSynchronize(st)); // (loads_global may be replaced by the next femshell_set_loads)
        }
    }
    if (what & kNodeMasks) {
        c->dm.dmask = c->dmask.p;
        launch_item_flags(c->dm, (int64_t)p.items.size(), st); // the Dirichlet set may have changed
        FS_HIP(hipGetLastError());
    }
    return FEMSHELL_OK;
}

// status and timing of an assembly that femshell_assemble_async enqueued: every entry point that reads results, changes
// inputs or synchronises calls this first
int finish_pending_assembly(femshell_ctx *c)
{
    if (!c->assembly_pending) return FEMSHELL_OK;
    c->assembly_pending = false;
    int rc = select_device(c);
    if (rc) return rc;
    rc = check_and_agree(c, "femshell_assemble_async");
    if (rc) {
        c->matrix_valid = false;
        c->rhs_valid = false;
        return rc;
    }
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_assemble_s = 1e-3 * ms;
    return FEMSHELL_OK;
}

int do_assemble(femshell_ctx *c, bool wait = true)
{
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_assemble: no mesh set");
    TraceRange trace("femshell_assemble");
    CommWatch watch(c->cfg.rank, c->comm.active() ? c->cfg.world_size : 1, "femshell_assemble (agreement of the ranks on the outcome)");
    int rc = select_device(c);
    if (rc) return rc;
    if (c->assembly_pending && wait) { // (async after async: one status word collects both)
        rc = finish_pending_assembly(c);
        if (rc) return rc;
    }
    // (the event pair costs the synchronous call its two records: without it assemble_seconds is the host's clock around launch
    //  and synchronisation; an enqueued assembly always carries the pair -- nobody else could time it)
    // (MEASURED, round 6, profiles/r06_sync_step_ab_raw.txt: neither the status word in mapped host memory, nor dropping the event
    //  pair, nor polling a word that a one-lane kernel writes behind the assembly instead of synchronising the stream, nor polled
    //  completion signals move the 40-50 us a synchronous step costs beyond its kernel: it is launch latency on an idle queue)
    const bool events = c->asm_events || !wait;
    const auto t_host = std::chrono::steady_clock::now();
    if (events) FS_HIP(hipEventRecord(c->ev0, c->stream));
    c->dm.rhs_loads = c->loads.p;
    c->dm.rhs_F = c->F.p;
    launch_assemble(c->dm, c->mc, c->stream); // K and F (k_rhs alone serves changes of the loads)
    if (events) FS_HIP(hipEventRecord(c->ev1, c->stream));
    FS_HIP(hipGetLastError());
    if (wait) {
        rc = check_and_agree(c, "femshell_assemble");
        if (rc) return rc;
        if (events) {
            float ms = 0.f;
            FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            c->last_assemble_s = 1e-3 * ms;
        } else {
            c->last_assemble_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host).count();
        }
    } else {
        c->assembly_pending = true; // (a failed element leaves its mark in the status word until someone looks)
    }
    c->matrix_valid = true;
    c->rhs_valid = true;
    c->jacobi_valid = false;
    c->amg.reset(); // the multigrid hierarchy belongs to the previous K
    return FEMSHELL_OK;
}

int do_rhs(femshell_ctx *c)
{
    launch_rhs(c->dm, c->loads.p, c->F.p, c->stream);
    FS_HIP(hipGetLastError());
    c->rhs_valid = true;
    return FEMSHELL_OK;
}

int do_jacobi(femshell_ctx *c)
{
    TraceRange trace("femshell block-Jacobi setup");
    FS_HIP(hipEventRecord(c->ev0, c->stream));
    launch_block_jacobi(c->dm, c->stream);
    FS_HIP(hipEventRecord(c->ev1, c->stream));
    FS_HIP(hipGetLastError());
    int rc = check_and_agree(c, "block-Jacobi setup");
    if (rc) return rc;
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_setup_s = 1e-3 * ms;
    c->jacobi_valid = true;
    return FEMSHELL_OK;
}

double wall_s()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

int clear_status_word(femshell_ctx *c, hipStream_t st)
{
    if (c->status_word == nullptr) return FEMSHELL_OK;
    if (c->status_mapped) {
        FS_HIP(hipStreamSynchronize(st));
        *(volatile int32_t *)c->status_host = 0;
    } else {
        FS_HIP(hipMemsetAsync(c->status_word, 0, sizeof(int32_t), st));
    }
    return FEMSHELL_OK;
}
int fetch_status_word(femshell_ctx *c, hipStream_t st, int32_t *out)
{
    if (!c->status_mapped) FS_HIP(hipMemcpyAsync(c->status_host, c->status_word, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    *out = *(volatile int32_t *)c->status_host;
    return FEMSHELL_OK;
}

namespace femshell {

// (col_out / val_out != nullptr: columns and values go straight into the caller's arrays -- femshell_export_bsr at the
//  4M-triangle sizes, where a second 4 GB copy of the blocks is not welcome -- and A keeps the row pointers only)
int download_matrix(femshell_ctx *c, Bsr *Aout, int32_t *col_out, double *val_out)
{
    const Plan &p = c->plan;
    // pinned landing buffer: a pageable std::vector would be zero-filled first and copied at a fraction of the PCIe rate
    struct Pinned {
        double *p = nullptr;
        ~Pinned() { if (p) (void)hipHostFree(p); }
        double *data() const { return p; }
    } h;
    const size_t h_size = (size_t)p.total_slots() * 36;
    FS_HIP(hipHostMalloc(reinterpret_cast<void **>(&h.p), h_size * sizeof(double), hipHostMallocDefault));
    FS_HIP(hipMemcpyAsync(h.data(), c->vals.p, h_size * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    Bsr &A = *Aout;
    A = Bsr();
    A.nr = p.n_own;   // this rank's node rows [row_begin, row_end)
    A.nc = p.n_nodes; // global column ids
    A.ptr.assign((size_t)p.n_own + 1, 0);
    auto global_col = [&](int32_t lc) { return lc < p.n_pad ? p.row_begin + lc : p.ghost_global[lc - p.n_pad]; };
    for (int32_t a = 0; a < p.n_own; a++) {
        const int s = a / kSliceNodes, n = a % kSliceNodes;
        int cnt = 0;
        for (int k = 0; k < p.slice_width[s]; k++) {
            const int64_t slot = Plan::slot_index(p.slice_base[s], k, n);
            if (p.pair_ptr[slot + 1] > p.pair_ptr[slot]) cnt++;
        }
        if (p.symmetric)
            for (int k = 0; k < p.in_width[s]; k++)
                if (p.in_slots[(size_t)(p.in_base[s] + (int64_t)k * kSliceNodes + n)] >= 0) cnt++;
        A.ptr[a + 1] = A.ptr[a] + cnt;
    }
    if (!col_out) {
        A.col.resize((size_t)A.ptr[p.n_own]);
        A.val.resize((size_t)A.ptr[p.n_own] * 36);
        col_out = A.col.data();
        val_out = A.val.data();
    }
    parallel_chunks(p.n_own, [&](int64_t a0, int64_t a1) {
        struct Src { int32_t col; int64_t slot; bool transposed; };
        std::vector<Src> order;
        for (int64_t a = a0; a < a1; a++) {
            const int s = (int)(a / kSliceNodes), n = (int)(a % kSliceNodes);
            const int64_t base = p.slice_base[s];
            order.clear();
            for (int k = 0; k < p.slice_width[s]; k++) {
                const int64_t slot = Plan::slot_index(base, k, n);
                if (p.pair_ptr[slot + 1] > p.pair_ptr[slot]) order.push_back({global_col(p.cols[slot]), slot, false});
            }
            if (p.symmetric) // blocks stored with the lower-numbered row: K(a, src) = K(src, a)^T
                for (int k = 0; k < p.in_width[s]; k++) {
                    const size_t e = (size_t)(p.in_base[s] + (int64_t)k * kSliceNodes + n);
                    if (p.in_slots[e] >= 0) order.push_back({p.row_begin + p.in_rows[e], (int64_t)p.in_slots[e], true});
                }
            std::sort(order.begin(), order.end(), [](const Src &x, const Src &y) { return x.col < y.col; });
            int64_t nb = A.ptr[a];
            for (const Src &sc : order) {
                col_out[nb] = sc.col;
                double *blk = val_out + (size_t)nb * 36;
                const int ns = (int)(sc.slot % kSliceNodes);
                const double *src = h.data() + (sc.slot - ns) * 36; // the (slice, k) group of 32 blocks
                // (a diagonal block of K holds its upper triangle only: DeviceMatrix::diag_upper)
                const bool upper_only = c->dm.diag_upper && !sc.transposed && sc.col == p.row_begin + (int32_t)a;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) {
                        const int r = (upper_only && j < i) ? j : i, cl = (upper_only && j < i) ? i : j;
                        const double v = src[(((int64_t)(cl / 2) * 6 + r) * kSliceNodes + ns) * 2 + (cl & 1)];
                        if (sc.transposed) blk[6 * j + i] = v; else blk[6 * i + j] = v;
                    }
                nb++;
            }
        }
    });
    return FEMSHELL_OK;
}

CgVectors cg_vectors(femshell_ctx *c)
{
    CgVectors v;
    v.x = c->x.p;
    v.r = c->r.p;
    v.z = c->z.p;
    v.p = c->p.p;
    v.q = c->q.p;
    v.sv = c->sv.p;
    v.b = c->F.p;
    v.partials = c->partials.p;
    v.s = c->scal.p;
    v.hist = c->hist.p;
    v.hist_cap = (int32_t)c->hist.n;
    return v;
}

} // namespace femshell

// =========================================================================================

extern "C" {

const char *femshell_last_error(void) { return last_err().c_str(); }

int femshell_create(const femshell_config *cfg, femshell_ctx **out)
{
    if (!cfg || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_create: null argument");
    *out = nullptr;
    if (!(cfg->nu > -1.0 && cfg->nu < 0.5 + 1e-12) || !(cfg->E > 0.0) || !(cfg->thickness > 0.0))
        return set_err(FEMSHELL_ERR_INVALID, "femshell_create: need -1 < nu <= 0.5, E > 0, thickness > 0");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_create: invalid rank/world_size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(FEMSHELL_ERR_NO_DEVICE, "femshell_create: no HIP device (this library has no CPU path)");
    int dev = cfg->device;
    if (dev < 0) FS_HIP(hipGetDevice(&dev));
    if (dev >= ndev) return set_err(FEMSHELL_ERR_NO_DEVICE, "femshell_create: device ordinal out of range");
    femshell_ctx *c = new femshell_ctx();
    DevPool::get().context_opened();
    c->cfg = *cfg;
    c->device = dev;
    plan_progress_hook = &CommWatch::heartbeat; // (the phases of the symbolic plan are progress in the eyes of the watchdog)
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    {
        const char *sm = getenv("FEMSHELL_STATUS_MAPPED"), *ae = getenv("FEMSHELL_ASM_EVENTS");
        c->status_mapped = !(sm && atoi(sm) == 0);
        c->asm_events = !(ae && atoi(ae) == 0);
    }
    if (e == hipSuccess) e = c->status.alloc(1);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->status_host), sizeof(int32_t), c->status_mapped ? hipHostMallocMapped : hipHostMallocDefault);
    if (e == hipSuccess) {
        *c->status_host = 0;
        c->status_word = c->status.p;
        if (c->status_mapped) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, c->status_host, 0) == hipSuccess && dp != nullptr) c->status_word = static_cast<int32_t *>(dp);
            else c->status_mapped = false;
        }
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->agree_host), 2 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) { // (optional: see context.hpp)
        c->stage_bytes = (size_t)32 << 20;
        if (hipHostMalloc(&c->stage_host, 2 * c->stage_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->stage_host = nullptr;
            c->stage_bytes = 0;
        }
    }
    if (e == hipSuccess) e = c->status.zero(c->stream);
    if (e == hipSuccess) e = c->scal.alloc(1);
    if (e == hipSuccess) e = c->scal.zero(c->stream);
    if (e == hipSuccess && c->stage_host != nullptr) {
        // the first copy from HBM into pinned memory on a stream takes 6 ms on these boxes (the runtime sets up the engine's queue
        // for that direction at first use; every later one 20 us): taken here, once per context, not inside the first setup
        // (64 KB: copies of a few bytes take another path and leave this one cold)
        femshell::DevBuf<char> warm;
        e = warm.alloc(65536);
        if (e == hipSuccess) e = hipMemsetAsync(warm.p, 0, 65536, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->stage_host, warm.p, 65536, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(warm.p, c->stage_host, 65536, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    // the second stream of the dense inverse's look-ahead (amg_dense.hip): made, and used once, here -- the first launch on a new
    // stream pays for its hardware queue (5 ms), which has no place inside the multigrid setup
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    // (the stream of the multigrid setup's helper thread -- context.hpp copy_stream -- for single-rank contexts only, and behind the
    //  other two: made lazily inside the first setup it cost the 250k-triangle roof 14 of 33 ms, made for every context it was one
    //  stream too many where two ranks share a card)
    if (e == hipSuccess && cfg->world_size <= 1) {
        e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemsetAsync(c->scal.p, 0, sizeof(*c->scal.p), c->copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->scal.p, 0, sizeof(*c->scal.p), c->aux_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->aux_stream);
    if (e == hipSuccess && amg_dense_probe_streams(c) != FEMSHELL_OK) e = hipErrorUnknown; // (do the two run side by side?  10 ms the first time)
    if (e != hipSuccess) {
        (void)femshell_destroy(c); // releases whatever was created
        return set_err(FEMSHELL_ERR_HIP, std::string("femshell_create: ") + hipGetErrorString(e));
    }
    const double nu = cfg->nu, E = cfg->E, t = cfg->thickness;
    c->mc.cm = E / (1.0 - nu * nu);
    c->mc.cp = E * t * t * t / (12.0 * (1.0 - nu * nu));
    c->mc.nu = nu;
    c->mc.g = (1.0 - nu) / 2.0;
    c->mc.t = t;
    if (!(c->cfg.flags & (FEMSHELL_REORDER_MORTON | FEMSHELL_REORDER_RCM)))
        if (const char *e = getenv("FEMSHELL_REORDER")) {
            if (std::strcmp(e, "morton") == 0) c->cfg.flags |= FEMSHELL_REORDER_MORTON;
            else if (std::strcmp(e, "rcm") == 0) c->cfg.flags |= FEMSHELL_REORDER_RCM;
        }
    c->mc.flags = cfg->flags;
    c->mc.pad = 0;
    if (cfg->world_size > 1) {
        // one process per GPU on one node (BASELINE: "the 8 GPUs of one node"): the ranks share the host's cores for the
        // plan and the multigrid setup; LOCAL_WORLD_SIZE (torchrun) / FEMSHELL_LOCAL_WORLD_SIZE say how many share this host
        const char *lw = getenv("FEMSHELL_LOCAL_WORLD_SIZE");
        if (!lw) lw = getenv("LOCAL_WORLD_SIZE");
        set_host_share(lw && atoi(lw) > 0 ? atoi(lw) : cfg->world_size);
    }
    {
        const char *e = getenv("FEMSHELL_PC");
        const bool amg = e && (std::strcmp(e, "amg") == 0 || std::strcmp(e, "gamg") == 0);
        (void)femshell_pc_defaults(amg ? FEMSHELL_PC_AMG : FEMSHELL_PC_BLOCK_JACOBI, &c->pc);
    }
    *out = c;
    return FEMSHELL_OK;
}

int femshell_destroy(femshell_ctx *c)
{
    if (!c) return FEMSHELL_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->amg.reset();
    comm_destroy(c->comm);
    if (c->halo_stream) (void)hipStreamSynchronize(c->halo_stream);
    if (c->ev_p_ready) (void)hipEventDestroy(c->ev_p_ready);
    if (c->ev_halo_done) (void)hipEventDestroy(c->ev_halo_done);
    if (c->halo_stream) (void)hipStreamDestroy(c->halo_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->agree_host) (void)hipHostFree(c->agree_host);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    DevPool::get().context_closed(); // (the last context of the process: the kept blocks go back to the driver)
    return FEMSHELL_OK;
}

int femshell_comm_unique_id(uint8_t id_out[128])
{
    std::string e;
    if (!id_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_comm_unique_id: null argument");
    if (!comm_unique_id(id_out, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    return FEMSHELL_OK;
}

int femshell_comm_init(femshell_ctx *c, const uint8_t id[128])
{
    if (!c || !id) return set_err(FEMSHELL_ERR_INVALID, "femshell_comm_init: null argument");
    // one rank needs no communicator; FEMSHELL_FORCE_COMM=1 creates a 1-rank RCCL communicator anyway so
    // that the RCCL code path (all-reduce on the stream, row gather) can be exercised on a single GPU
    if (c->cfg.world_size == 1 && !(getenv("FEMSHELL_FORCE_COMM") && atoi(getenv("FEMSHELL_FORCE_COMM")) == 1))
        return FEMSHELL_OK;
    int rc = select_device(c);
    if (rc) return rc;
    std::string e;
    if (!comm_init(c->comm, id, c->cfg.rank, c->cfg.world_size, &e)) return set_err(FEMSHELL_ERR_COMM, e);
    {
        // the first collective, under the watch: an all-reduce of one word that must come back as the rank count.  A
        // communicator that initialises but cannot move data (a fabric or IPC problem) shows here, not in the first solve.
        CommWatch watch(c->cfg.rank, c->cfg.world_size, "first all-reduce after ncclCommInitRank");
        FS_HIP(c->agree.alloc(2));
        const double one = 1.0;
        FS_HIP(hipMemcpyAsync(c->agree.p, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
        if (!comm_allreduce_sum(c->comm, c->agree.p, 1, c->stream, &e)) return set_err(FEMSHELL_ERR_COMM, e);
        double got = 0.0;
        FS_HIP(hipMemcpyAsync(&got, c->agree.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        if (got != (double)c->cfg.world_size)
            return set_err(FEMSHELL_ERR_COMM, "femshell_comm_init: the first all-reduce returned " + std::to_string(got) + " instead of the rank count " +
                                                  std::to_string(c->cfg.world_size));
    }
    {
        // ... and the patterns the solves use, once each with a known answer (comm.cpp comm_selftest): the halo exchange's grouped
        // send/recv on the second stream beside an all-reduce on the main one, the grouped broadcasts of the row gather.  The first
        // real multi-GPU run either passes here or names the pattern that failed -- or, stalled, is ended by the watchdog with the
        // pattern's name -- instead of hanging in its first solve.
        if (!c->halo_stream) FS_HIP(hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
        c->comm.second = c->halo_stream;
        DevBuf<double> scratch;
        FS_HIP(scratch.alloc((size_t)16 + 2 * (size_t)c->cfg.world_size));
        if (!comm_selftest(c->comm, c->stream, c->halo_stream, scratch.p, c->comm_selftest_us, &e)) return set_err(FEMSHELL_ERR_COMM, "femshell_comm_init: " + e);
        c->comm_selftest_done = true;
    }
    return FEMSHELL_OK;
}

int femshell_comm_counters(femshell_ctx *c, int64_t out[4], int32_t clear)
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_comm_counters: null argument");
    out[0] = c->comm.halo_groups_second;
    out[1] = c->comm.halo_groups_main;
    out[2] = c->comm.allreduces;
    out[3] = c->comm.gathers;
    if (clear) c->comm.halo_groups_second = c->comm.halo_groups_main = c->comm.allreduces = c->comm.gathers = 0;
    return c->comm.active() ? 1 : 0;
}

int femshell_comm_bytes(femshell_ctx *c, int64_t out[2], int32_t clear)
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_comm_bytes: null argument");
    out[0] = c->comm.halo_bytes_sent;
    out[1] = c->comm.collective_bytes;
    if (clear) c->comm.halo_bytes_sent = c->comm.collective_bytes = 0;
    return c->comm.active() ? 1 : 0;
}

int femshell_comm_selftest(femshell_ctx *c, double out_us[3])
{
    if (!c || !out_us) return set_err(FEMSHELL_ERR_INVALID, "femshell_comm_selftest: null argument");
    for (int i = 0; i < 3; i++) out_us[i] = c->comm_selftest_done ? c->comm_selftest_us[i] : -1.0;
    return c->comm_selftest_done ? 1 : 0;
}

int32_t femshell_comm_ranks(femshell_ctx *c) { return c ? comm_count(c->comm) : 0; }

static int set_mesh_on_this_rank(femshell_ctx *c, int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                                 int32_t n_quad, const int32_t *quad);

int femshell_set_mesh(femshell_ctx *c, int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                      int32_t n_quad, const int32_t *quad)
{
    if (!c || !xyz || (n_tri > 0 && !tri) || (n_quad > 0 && !quad))
        return set_err(FEMSHELL_ERR_INVALID, "femshell_set_mesh: null argument");
    if (c->cfg.world_size > 1 && !c->comm.active())
        return set_err(FEMSHELL_ERR_INVALID, "femshell_set_mesh: call femshell_comm_init first on a multi-rank context");
    CommWatch watch(c->cfg.rank, c->comm.active() ? c->cfg.world_size : 1, "femshell_set_mesh (agreement of the ranks on the outcome)");
    int rc = select_device(c);
    if (rc) return rc;
    if (c->assembly_pending) {
        // an assembly of the previous mesh whose status nobody asked for: its mark in the device status word (a degenerate
        // element of the OLD mesh) must not surface as a failure of the first assembly on the new one
        c->assembly_pending = false;
        {
            const int rcc = clear_status_word(c, c->stream);
            if (rcc) return rcc;
        }
    }
    // a failure that only this rank sees (its slices exceed the LDS staging, one of its nodes has too many neighbours, a
    // HIP allocation failed) must reach the others: they would wait in the next collective forever
    return agree_status(c, set_mesh_on_this_rank(c, n_nodes, xyz, n_tri, tri, n_quad, quad), "femshell_set_mesh", true);
}

static int set_mesh_on_this_rank(femshell_ctx *c, int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri,
                                 int32_t n_quad, const int32_t *quad)
{
    int rc = FEMSHELL_OK;
    TraceRange trace("femshell_set_mesh (plan + upload)");
    // FEMSHELL_PLAN_VERBOSE=1: wall time of the call's parts on stderr (the plan prints its own phases)
    static const bool verbose = getenv("FEMSHELL_PLAN_VERBOSE") && atoi(getenv("FEMSHELL_PLAN_VERBOSE")) != 0;
    double t_part = wall_s();
    auto part_done = [&](const char *what) {
        const double t = wall_s();
        if (verbose) fprintf(stderr, "[femshell set_mesh] %-40s %.3f s\n", what, t - t_part);
        t_part = t;
    };
    {
        std::atomic<int> bad{0};
        parallel_chunks(3ll * n_nodes, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; i++)
                if (!std::isfinite(xyz[i])) bad.store(1);
        }, 1 << 18);
        if (bad.load()) return set_err(FEMSHELL_ERR_MESH, "femshell_set_mesh: non-finite coordinate");
    }
    std::string e;
    c->have_mesh = false;
    c->warm_next = false;
    c->amg_fp64_only = false;
    c->perm.clear();
    c->iperm.clear();
    std::vector<double> xyz_r;
    std::vector<int32_t> tri_r, quad_r;
    // (row-partitioned contexts: every rank holds the whole mesh and computes the same permutation -- both orderings are
    //  serial and deterministic -- BEFORE the slice partition, so a rank's rows are a stretch of the curve / of the level
    //  structure: compact in space whatever the caller's numbering is, ghosts along its two ends only)
    if ((c->cfg.flags & (FEMSHELL_REORDER_MORTON | FEMSHELL_REORDER_RCM)) && n_nodes > 0) {
        for (int64_t q = 0; q < 3ll * n_tri; q++)
            if (tri[q] < 0 || tri[q] >= n_nodes) return set_err(FEMSHELL_ERR_MESH, "femshell_set_mesh: triangle " + std::to_string(q / 3) + " references a node out of range");
        for (int64_t q = 0; q < 4ll * n_quad; q++)
            if (quad[q] < 0 || quad[q] >= n_nodes) return set_err(FEMSHELL_ERR_MESH, "femshell_set_mesh: quad " + std::to_string(q / 4) + " references a node out of range");
        if (c->cfg.flags & FEMSHELL_REORDER_RCM) rcm_order(n_nodes, n_tri, tri, n_quad, quad, &c->perm);
        else morton_order(n_nodes, xyz, &c->perm);
        c->iperm.assign((size_t)n_nodes, 0);
        for (int32_t i = 0; i < n_nodes; i++) c->iperm[c->perm[i]] = i;
        xyz_r.resize((size_t)n_nodes * 3);
        for (int32_t i = 0; i < n_nodes; i++)
            for (int d = 0; d < 3; d++) xyz_r[3ull * i + d] = xyz[3ull * c->perm[i] + d];
        tri_r.resize((size_t)n_tri * 3);
        for (int64_t q = 0; q < 3ll * n_tri; q++) tri_r[(size_t)q] = c->iperm[tri[q]];
        quad_r.resize((size_t)n_quad * 4);
        for (int64_t q = 0; q < 4ll * n_quad; q++) quad_r[(size_t)q] = c->iperm[quad[q]];
        xyz = xyz_r.data();
        tri = tri_r.data();
        quad = quad_r.data();
    }
    if (!build_plan(n_nodes, xyz, n_tri, tri, n_quad, quad, c->cfg.rank, c->cfg.world_size, &c->plan, &e, default_symmetric_storage(),
                    /* geometric orientation of the symmetric storage when the library chose the numbering: */ !c->perm.empty()))
        return set_err(FEMSHELL_ERR_MESH, "femshell_set_mesh: " + e);
    part_done("coordinate check, renumbering, plan");
    c->have_mesh_centre = false;
    if (c->cfg.world_size > 1 || c->comm.active()) { // (the row-partitioned multigrid: rigid-body modes about the centre of the whole mesh)
        mesh_centre(n_nodes, xyz, c->mesh_centre);
        c->have_mesh_centre = true;
    }
    const Plan &p = c->plan;
    hipStream_t st = c->stream;
    FS_HIP(c->xyz.upload(p.xyz_local, st));
    FS_HIP(c->tri.upload(p.tri_local, st));
    FS_HIP(c->quad.upload(p.quad_local, st));
    FS_HIP(c->slice_width.upload(p.slice_width, st));
    FS_HIP(c->slice_base.upload(p.slice_base, st));
    FS_HIP(c->cols.upload(p.cols, st));
    FS_HIP(c->pair_ptr.upload(p.pair_ptr, st));
    FS_HIP(c->slice_elem_ptr.upload(p.slice_elem_ptr, st));
    FS_HIP(c->slice_elem_nodes.upload(p.slice_elem_nodes, st));
    FS_HIP(c->item_ptr.upload(p.item_ptr, st));
    FS_HIP(c->slice_desc.upload(p.slice_desc, st));
    FS_HIP(c->items.upload(p.items, st));
    FS_HIP(c->in_width.upload(p.in_width, st));
    FS_HIP(c->in_base.upload(p.in_base, st));
    FS_HIP(c->in_slots.upload(p.in_slots, st));
    FS_HIP(c->in_rows.upload(p.in_rows, st));
    FS_HIP(c->gat_slots.upload(p.gat_slots, st));
    FS_HIP(c->loc_index.upload(p.loc_index, st));
    FS_HIP(c->loc_list.upload(p.loc_list, st));
    part_done("uploads of the plan");
    if (p.symmetric) {
        FS_HIP(c->tbuf.alloc((size_t)p.total_slots() * 6)); // transposed products next to every slot
        FS_HIP(c->tbuf.zero(st));
    } else {
        c->tbuf.release();
    }
    const size_t nrow = (size_t)p.n_pad * 6, nrow_ext = (size_t)p.n_local_nodes() * 6;
    FS_HIP(c->vals.alloc((size_t)p.total_slots() * 36));
    FS_HIP(c->vals.zero(st)); // the padding slots of the ELL layout stay zero; assembly writes the real blocks only
    FS_HIP(c->minv.alloc((size_t)p.n_slices * 21 * kSliceNodes)); // upper triangles of the inverse diagonal blocks
    FS_HIP(c->F.alloc(nrow));
    FS_HIP(c->x.alloc(nrow));
    FS_HIP(c->r.alloc(nrow));
    FS_HIP(c->z.alloc(nrow_ext)); // ghost space: the single-reduction recurrence multiplies K by z
    FS_HIP(c->sv.alloc(nrow));
    FS_HIP(c->q.alloc(nrow));
    FS_HIP(c->p.alloc(nrow_ext));
    FS_HIP(c->x.zero(st));
    FS_HIP(c->p.zero(st));
    FS_HIP(c->q.zero(st));
    FS_HIP(c->z.zero(st));
    c->dm = DeviceMatrix();
    c->dm.n_own = p.n_own;
    c->dm.n_pad = p.n_pad;
    c->dm.n_ghost = p.n_ghost;
    c->dm.n_slices = p.n_slices;
    c->dm.n_ltri = p.n_ltri();
    c->dm.n_lquad = p.n_lquad();
    c->dm.xyz = c->xyz.p;
    c->dm.tri = c->tri.p;
    c->dm.quad = c->quad.p;
    c->dm.slice_width = c->slice_width.p;
    c->dm.slice_base = c->slice_base.p;
    c->dm.cols = c->cols.p;
    c->dm.pair_ptr = c->pair_ptr.p;
    c->dm.slice_elem_ptr = c->slice_elem_ptr.p;
    c->dm.slice_elem_nodes = reinterpret_cast<const int4 *>(c->slice_elem_nodes.p);
    c->dm.max_slice_elems = p.max_slice_elems;
    c->dm.max_slice_width = p.max_slice_width;
    c->dm.item_ptr = c->item_ptr.p;
    c->dm.slice_desc = reinterpret_cast<const int4 *>(c->slice_desc.p);
    c->dm.items = reinterpret_cast<const uint4 *>(c->items.p);
    FS_HIP(c->item_flags.alloc(p.pipe ? 0 : p.items.size())); // (pipelined layout: the word rides in the item, kernels.hip)
    c->dm.item_flags = c->item_flags.p;
    c->dm.max_stage_rows = p.max_stage_rows;
    c->dm.pipe = p.pipe ? 1 : 0;
    c->dm.slice_elem_ptr_last = (int32_t)(p.slice_elem_nodes.size() / 4);
    {
        // LDS of k_assemble: element records + partial-sum staging
        const size_t lds = assemble_lds_layout(c->dm, p.max_slice_elems, p.max_stage_rows, p.n_lquad() > 0);
        if (p.max_slice_width > 64) return set_err(FEMSHELL_ERR_UNSUPPORTED, "femshell_set_mesh: a node has more than 63 neighbours");
        if (lds > 160 * 1024) // one workgroup per CU at most; well-numbered meshes need about 50 KB (three per CU)
            return set_err(FEMSHELL_ERR_UNSUPPORTED, "femshell_set_mesh: a 32-node slice touches too many elements for the LDS staging");
    }
    c->dm.vals = c->vals.p;
    c->dm.symmetric = p.symmetric ? 1 : 0;
    // the lower words of the diagonal blocks are not written (kernels.hpp); FEMSHELL_DIAG_UPPER=0: all 18 words (A/B runs)
    c->dm.diag_upper = (p.symmetric && !(getenv("FEMSHELL_DIAG_UPPER") && atoi(getenv("FEMSHELL_DIAG_UPPER")) == 0)) ? 1 : 0;
    c->dm.max_in_width = p.max_in_width;
    c->dm.in_width = c->in_width.p;
    c->dm.in_base = c->in_base.p;
    c->dm.in_slots = c->in_slots.p;
    c->dm.in_rows = c->in_rows.p;
    // FEMSHELL_SPMV_LOCAL=0: every transposed product through HBM (A/B runs)
    const bool local_products = p.symmetric && p.max_loc > 0 && !(getenv("FEMSHELL_SPMV_LOCAL") && atoi(getenv("FEMSHELL_SPMV_LOCAL")) == 0);
    c->dm.gat_slots = local_products ? c->gat_slots.p : c->in_slots.p;
    c->dm.loc_index = local_products ? c->loc_index.p : nullptr;
    c->dm.loc_list = local_products ? c->loc_list.p : nullptr;
    c->dm.max_loc = local_products ? p.max_loc : 0;
    c->dm.tbuf = c->tbuf.p;
    c->dm.minv = c->minv.p;
    c->dm.status = c->status_word;
    FS_HIP(c->partials.alloc(4 * (size_t)slice_grid(c->dm))); // r.z | r.r | the SpMV's dot (up to 2 x grid when split)
    // halo lists
    c->send_offsets.clear();
    std::vector<int32_t> flat;
    for (const HaloPeer &h : p.peers) {
        c->send_offsets.push_back((int32_t)flat.size());
        flat.insert(flat.end(), h.send_nodes.begin(), h.send_nodes.end());
    }
    FS_HIP(c->send_nodes.upload(flat, st));
    FS_HIP(c->sendbuf.alloc(flat.size() * 6));
    // overlap of the halo exchange with the interior SpMV (FEMSHELL_HALO_OVERLAP=0 keeps everything on one stream)
    c->halo_overlap = false;
    if (c->comm.active() && !p.peers.empty() && p.n_interior_slices > 0 &&
        !(getenv("FEMSHELL_HALO_OVERLAP") && atoi(getenv("FEMSHELL_HALO_OVERLAP")) == 0)) {
        if (!c->halo_stream) { // (femshell_comm_init creates it for its self-test)
            FS_HIP(hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
            c->comm.second = c->halo_stream;
        }
        if (!c->ev_p_ready) FS_HIP(hipEventCreateWithFlags(&c->ev_p_ready, hipEventDisableTiming));
        if (!c->ev_halo_done) FS_HIP(hipEventCreateWithFlags(&c->ev_halo_done, hipEventDisableTiming));
        FS_HIP(c->spmv_order.upload(p.spmv_order, st));
        c->halo_overlap = true;
    }
    c->all_begin.resize(p.world);
    c->all_end.resize(p.world);
    for (int r = 0; r < p.world; r++) {
        c->all_begin[r] = p.part_bounds[(size_t)r];
        c->all_end[r] = p.part_bounds[(size_t)r + 1];
    }
    c->dmask_global.assign((size_t)n_nodes, 0);
    c->loads_global.resize((size_t)n_nodes * 6);
    parallel_chunks((int64_t)c->loads_global.size(), [&](int64_t b, int64_t e) { std::fill(c->loads_global.begin() + b, c->loads_global.begin() + e, 0.0); }, 1 << 18);
    rc = upload_node_data(c, kNodeCleared);
    if (rc) return rc;
    FS_HIP(hipStreamSynchronize(st));
    part_done("allocations, fills, node data");
    c->have_mesh = true;
    c->matrix_valid = c->rhs_valid = c->jacobi_valid = c->have_solution = false;
    return FEMSHELL_OK;
}

int femshell_set_dirichlet(femshell_ctx *c, int32_t n, const int32_t *node_ids, const uint8_t *mask6)
{
    if (!c || (n > 0 && !mask6)) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_dirichlet: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_dirichlet: call femshell_set_mesh first");
    const int32_t nn = c->plan.n_nodes;
    if (!node_ids && n != nn) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_dirichlet: dense form needs n == n_nodes");
    std::vector<uint8_t> m((size_t)nn, 0);
    for (int32_t i = 0; i < n; i++) {
        const int32_t a = node_ids ? node_ids[i] : i;
        if (a < 0 || a >= nn) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_dirichlet: node id out of range");
        if (mask6[i] & ~0x3Fu) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_dirichlet: mask has bits above dof 5");
        m[c->iperm.empty() ? a : c->iperm[a]] |= mask6[i];
    }
    int rc = select_device(c);
    if (rc) return rc;
    c->dmask_global.swap(m);
    rc = upload_node_data(c, kNodeMasks);
    if (rc) return rc;
    c->matrix_valid = c->rhs_valid = c->jacobi_valid = false;
    return FEMSHELL_OK;
}

int femshell_set_loads(femshell_ctx *c, int32_t n, const int32_t *node_ids, const double *f6)
{
    if (!c || (n > 0 && !f6)) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: call femshell_set_mesh first");
    const int32_t nn = c->plan.n_nodes;
    if (!node_ids && n != nn) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: dense form needs n == n_nodes");
    RawVec<double> l((size_t)nn * 6);
    if (!node_ids && c->iperm.empty()) {
        // the dense form in the caller's own numbering: checked and copied on the host threads
        std::atomic<int> bad{0};
        parallel_chunks(6ll * nn, [&](int64_t b, int64_t e) {
            for (int64_t q = b; q < e; q++) {
                if (!std::isfinite(f6[q])) bad.store(1);
                l[(size_t)q] = f6[q];
            }
        }, 1 << 18);
        if (bad.load()) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: non-finite load");
    } else {
        parallel_chunks((int64_t)l.size(), [&](int64_t b, int64_t e) { std::fill(l.begin() + b, l.begin() + e, 0.0); }, 1 << 18);
        for (int32_t i = 0; i < n; i++) {
            const int32_t a = node_ids ? node_ids[i] : i;
            if (a < 0 || a >= nn) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: node id out of range");
            for (int v = 0; v < 6; v++) {
                if (!std::isfinite(f6[6ll * i + v])) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_loads: non-finite load");
                l[6ull * (size_t)(c->iperm.empty() ? a : c->iperm[a]) + v] = f6[6ll * i + v];
            }
        }
    }
    int rc = select_device(c);
    if (rc) return rc;
    c->loads_global.swap(l);
    rc = upload_node_data(c, kNodeLoads);
    if (rc) return rc;
    c->rhs_valid = false;
    return FEMSHELL_OK;
}

int femshell_assemble(femshell_ctx *c)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_assemble: null context");
    return do_assemble(c);
}

int femshell_assemble_async(femshell_ctx *c)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_assemble_async: null context");
    return do_assemble(c, false);
}

int femshell_pc_defaults(int32_t type, femshell_pc_options *out)
{
    if (!out) return set_err(FEMSHELL_ERR_INVALID, "femshell_pc_defaults: null argument");
    if (type != FEMSHELL_PC_BLOCK_JACOBI && type != FEMSHELL_PC_AMG)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_pc_defaults: unknown preconditioner type");
    amg_default_options(out);
    out->type = type;
    if (type == FEMSHELL_PC_AMG) {
        const char *cy = getenv("FEMSHELL_AMG_CYCLE");
        if (cy && (cy[0] == 'V' || cy[0] == 'v')) out->cycle = FEMSHELL_CYCLE_V;
        const char *rp = getenv("FEMSHELL_REFINE_PASSES");
        if (rp && atoi(rp) >= 0 && atoi(rp) <= 4) out->refine_passes = atoi(rp);
    }
    return FEMSHELL_OK;
}

int femshell_set_preconditioner(femshell_ctx *c, const femshell_pc_options *opt)
{
    if (!c || !opt) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_preconditioner: null argument");
    if (opt->type != FEMSHELL_PC_BLOCK_JACOBI && opt->type != FEMSHELL_PC_AMG)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_set_preconditioner: unknown preconditioner type");
    if (opt->type == FEMSHELL_PC_AMG) {
        if ((opt->cycle != FEMSHELL_CYCLE_V && opt->cycle != FEMSHELL_CYCLE_K) || opt->smoother_degree < 1 || opt->smoother_degree > 16 ||
            opt->coarse_degree < 1 || opt->coarse_degree > 16 || opt->coarsest_nodes < 1 || opt->coarsest_nodes > 1400 ||
            opt->max_levels < 2 || opt->max_levels > 32 || !(opt->eig_ratio > 1.0) || opt->refine_passes < 0 || opt->refine_passes > 4)
            return set_err(FEMSHELL_ERR_INVALID, "femshell_set_preconditioner: option out of range");
    }
    // the hierarchy does not depend on the refinement passes: keep it when nothing else changes
    femshell_pc_options a = c->pc, b = *opt;
    a.refine_passes = b.refine_passes = 0;
    a.reserved = b.reserved = 0;
    const bool same_hierarchy = std::memcmp(&a, &b, sizeof a) == 0;
    c->pc = *opt;
    if (same_hierarchy && c->amg && c->amg->valid) {
        c->amg->opt = *opt;
    } else {
        c->amg.reset();
        c->amg_fp64_only = false;
    }
    return FEMSHELL_OK;
}

int32_t femshell_amg_levels(femshell_ctx *c) { return (c && c->amg && c->amg->valid) ? (int32_t)c->amg->levels.size() : 0; }

int femshell_amg_level(femshell_ctx *c, int32_t level, femshell_amg_level_info *info)
{
    if (!c || !info) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_level: null argument");
    if (level < 0 || level >= femshell_amg_levels(c)) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_level: no such level");
    const AmgLevel &L = *c->amg->levels[level];
    const bool last = level + 1 == (int32_t)c->amg->levels.size();
    // (row-partitioned levels: the level as a whole, not the rank's rows)
    auto nodes_of = [](const AmgLevel &lv) { return lv.dist || lv.n_global > 0 ? lv.n_global : lv.n; };
    info->n_nodes = nodes_of(L);
    info->n_coarse = last ? 0 : nodes_of(*c->amg->levels[level + 1]);
    info->nnz_blocks = L.nnzb;
    info->p_blocks = last ? 0 : L.P.nnzb;
    info->lambda_max = L.lam;
    return FEMSHELL_OK;
}

int femshell_amg_patch_info(femshell_ctx *c, double out[6])
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_patch_info: null argument");
    if (!c->amg || !c->amg->valid || c->amg->levels.empty()) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_patch_info: no multigrid hierarchy");
    for (int i = 0; i < 6; i++) out[i] = 0.0;
    const AmgLevel &L = *c->amg->levels[0];
    if (L.patches) {
        out[0] = (double)L.patches->edges;
        out[1] = L.patches->n_clusters;
        out[2] = L.patches->n_members;
        out[3] = L.patches->fell_back;
        out[4] = L.patches->tau;
        out[5] = L.patches->max_nodes;
    }
    return FEMSHELL_OK;
}

int femshell_amg_setup_stats(femshell_ctx *c, double out[7])
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_setup_stats: null argument");
    if (!c->amg || !c->amg->valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_setup_stats: no multigrid hierarchy");
    const AmgSetupStats &S = c->amg->stats;
    out[0] = S.prolongator_ms;
    out[1] = S.ap_ms;
    out[2] = S.restriction_ms;
    out[3] = S.galerkin_ms;
    out[4] = S.galerkin_useful_flops;
    out[5] = S.galerkin_mfma_flops_issued;
    out[6] = (double)S.galerkin_mfma;
    return FEMSHELL_OK;
}

int femshell_amg_symbolic_info(femshell_ctx *c, int32_t out[3])
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_symbolic_info: null argument");
    if (!c->amg || !c->amg->valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_symbolic_info: no multigrid hierarchy");
    const AmgSetupStats &S = c->amg->stats;
    out[0] = S.symbolic_device;
    out[1] = S.symbolic_fallback;
    out[2] = S.symbolic_host;
    return FEMSHELL_OK;
}

int femshell_amg_partition_info(femshell_ctx *c, double out[6])
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_partition_info: null argument");
    if (!c->amg || !c->amg->valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_partition_info: no multigrid hierarchy");
    const Amg &H = *c->amg;
    for (int i = 0; i < 6; i++) out[i] = 0.0;
    if (H.dist) {
        out[0] = (double)H.dist->dist_levels;
        out[1] = H.dist->hierarchy_bytes_partitioned;
        out[2] = H.dist->hierarchy_bytes_replicated;
        const AmgLevel &last = *H.levels[(size_t)H.dist->dist_levels - 1];
        out[3] = (double)last.n;        // the rank's rows of the last row-partitioned level
        out[4] = (double)last.n_ghost;  // ... and the ghost rows it reads
        out[5] = (double)H.levels[(size_t)H.dist->dist_levels]->n; // nodes of the first replicated level
    }
    return FEMSHELL_OK;
}

int32_t femshell_amg_cycle_bytes(femshell_ctx *c, double *per_level, int32_t cap)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_cycle_bytes: null context");
    if (!c->amg || !c->amg->valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_cycle_bytes: no multigrid hierarchy");
    std::vector<double> b;
    (void)amg_cycle_bytes(c, &b);
    for (int32_t l = 0; per_level != nullptr && l < cap && l < (int32_t)b.size(); l++) per_level[l] = b[(size_t)l];
    return (int32_t)b.size();
}

int femshell_assembly_kernel(femshell_ctx *c)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_assembly_kernel: null context");
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_assembly_kernel: no mesh");
    return c->dm.pipe ? 1 : 0;
}

int femshell_amg_dense_stats(femshell_ctx *c, double out[6])
{
    if (!c || !out) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_dense_stats: null argument");
    if (!c->amg || !c->amg->valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_amg_dense_stats: no multigrid hierarchy");
    const AmgDenseStats &S = c->amg->dense;
    out[0] = (double)S.n;
    out[1] = S.ms;
    out[2] = S.mfma_flops;
    out[3] = S.useful_flops;
    out[4] = (double)S.dropped;
    out[5] = S.bytes;
    return FEMSHELL_OK;
}

int64_t femshell_amg_export(femshell_ctx *c, int32_t level, int32_t which, void *out)
{
    if (!c || level < 0 || level >= femshell_amg_levels(c)) return -1;
    const AmgLevel &L = *c->amg->levels[level];
    auto give = [&](const void *src, size_t count, size_t elem) -> int64_t {
        if (out && count) std::memcpy(out, src, count * elem);
        return (int64_t)count;
    };
    if (which == FEMSHELL_AMG_COARSE_INVERSE) {
        const Amg &H = *c->amg;
        if (level + 1 != (int32_t)H.levels.size()) return -1;
        const int64_t n = 6ll * L.n;
        if (!out) return n * n;
        if (select_device(c)) return -1;
        double *o = static_cast<double *>(out);
        if (H.coarse_lda > 0 && H.coarse_inv.p) { // computed on the matrix cores: rows of coarse_lda doubles
            if (hipMemcpy2D(o, (size_t)n * 8, H.coarse_inv.p, (size_t)H.coarse_lda * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        } else if (H.coarse_lda > 0 && H.coarse_inv32.p) {
            std::vector<float> h((size_t)n * H.coarse_lda);
            if (hipMemcpy(h.data(), H.coarse_inv32.p, h.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            for (int64_t r = 0; r < n; r++)
                for (int64_t q = 0; q < n; q++) o[r * n + q] = (double)h[(size_t)(r * H.coarse_lda + q)];
        } else {
            if (hipMemcpy(o, H.coarse_inv.p, (size_t)n * n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        }
        return n * n;
    }
    if (which == FEMSHELL_AMG_PATCH_LABELS) return L.patches ? give(L.patches->label.data(), L.patches->label.size(), sizeof(int32_t)) : -1;
    const Bsr &M = (which >= FEMSHELL_AMG_P_ROWPTR) ? L.hP : L.hA;
    switch (which) {
    case FEMSHELL_AMG_AGGREGATES: return L.agg.empty() ? -1 : give(L.agg.data(), L.agg.size(), sizeof(int32_t));
    case FEMSHELL_AMG_A_ROWPTR:
    case FEMSHELL_AMG_P_ROWPTR: return M.ptr.empty() ? -1 : give(M.ptr.data(), M.ptr.size(), sizeof(int64_t));
    case FEMSHELL_AMG_A_COLS:
    case FEMSHELL_AMG_P_COLS: return M.ptr.empty() ? -1 : give(M.col.data(), M.col.size(), sizeof(int32_t));
    case FEMSHELL_AMG_A_VALS:
    case FEMSHELL_AMG_P_VALS: return M.ptr.empty() ? -1 : give(M.val.data(), M.val.size(), sizeof(double));
    default: return -1;
    }
}

int femshell_solve(femshell_ctx *c, double rtol, int32_t max_it, double *u_out, femshell_solve_info *info)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_solve: null context");
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_solve: no mesh set");
    if (max_it < 0) return set_err(FEMSHELL_ERR_INVALID, "femshell_solve: max_it < 0");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    TraceRange trace("femshell_solve");
    // (the CG loops report progress at every poll of the convergence flag: CommWatch::heartbeat)
    CommWatch watch(c->cfg.rank, c->comm.active() ? c->cfg.world_size : 1, "femshell_solve (halo exchanges and all-reduces of the solve)");
    int rc = select_device(c);
    if (rc) return rc;
    double asm_s = 0.0;
    if (!c->matrix_valid || (c->cfg.flags & FEMSHELL_REASSEMBLE_EACH_SOLVE)) {
        rc = do_assemble(c);
        if (rc) return rc;
        asm_s = c->last_assemble_s;
    } else if (!c->rhs_valid) {
        rc = do_rhs(c);
        if (rc) return rc;
    }
    double setup_s = 0.0;
    if (!c->jacobi_valid) {
        rc = do_jacobi(c);
        if (rc) return rc;
        setup_s = c->last_setup_s;
    }
    double pc_setup_s = 0.0;
    const bool use_amg = c->pc.type == FEMSHELL_PC_AMG;
    hipStream_t st = c->stream;
    TraceRange trace_cg(use_amg ? "femshell_solve: multigrid-preconditioned CG" : "femshell_solve: block-Jacobi CG");
    FS_HIP(c->hist.alloc((size_t)std::min<int64_t>(std::max(max_it, 1), 1 << 22))); // history of the first 4M iterations
    CgVectors v = cg_vectors(c);
    const DeviceMatrix &m = c->dm;
    const bool single_reduction = !use_amg && use_single_reduction(c);
    const double *x0 = c->warm_next ? c->x0.p : nullptr; // femshell_set_initial_guess: this solve's, and only this one's
    c->warm_next = false;
    double amg_true_rr = -1.0, amg_rec_rr = -1.0;
    CgScalars hs{};
    bool fp64_fallback = false;
    float ms_abandoned = 0.f; // device time of an attempt that ended in the breakdown below: part of solve_seconds
    for (int attempt = 0;; attempt++) {
        if (use_amg && (!c->amg || !c->amg->valid)) {
            if (c->comm.active()) {
                const double t0 = wall_s();
                // row-partitioned hierarchy (amg_dist.cpp); a failure only one rank sees must reach the others
                rc = agree_status(c, amg_setup_dist(c), "multigrid setup", true);
                if (rc) {
                    c->amg.reset();
                    return rc;
                }
                pc_setup_s += wall_s() - t0;
            } else {
                rc = amg_setup(c);
                if (rc) {
                    c->amg.reset();
                    return rc;
                }
                pc_setup_s += c->amg->setup_seconds;
            }
        }
        // a previous solve leaves done = 1 behind; the reduction launch in front of an all-reduce carries no phase and
        // would skip its work on it (multi-rank re-solves, e.g. every coupling iteration)
        FS_HIP(hipEventRecord(c->ev0, st)); // (behind the multigrid setup: solve_seconds is the time of the Krylov loop)
        FS_HIP(c->scal.zero(st));
        amg_true_rr = amg_rec_rr = -1.0;
        // (from an initial guess the block-Jacobi method runs its classic recurrence: the single-reduction form has no such start)
        rc = use_amg ? cg_amg(c, v, rtol, max_it, &amg_true_rr, &amg_rec_rr, x0)
                     : (single_reduction && !x0) ? cg_single_reduction(c, v, rtol, max_it) : cg_classic(c, v, rtol, max_it, x0);
        if (rc) return rc;
        FS_HIP(hipMemcpyAsync(&hs, v.s, sizeof hs, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        // A breakdown of the flexible CG (p.Ap <= 0, the same all-reduced number on every rank) under a hierarchy that keeps
        // single-precision copies: on very thin shells the rounded preconditioner is not positive definite any more.  Once:
        // the hierarchy again, everything FP64, and the solve from the start.  The context stays that way until a new mesh
        // or preconditioner is set; femshell_solve_info::pc_fp64_fallback says it happened.
        // (row-partitioned contexts decide by what every rank knows alike -- the rebuild is collective --, not by what this
        //  rank's part of the hierarchy happens to hold: a rank without rows on a split level keeps no copies)
        const bool had_copies = use_amg && (c->comm.active() || amg_uses_single_precision(*c->amg));
        if (use_amg && hs.done < 0 && attempt == 0 && !c->amg_fp64_only && had_copies) {
            FS_HIP(hipEventRecord(c->ev1, st));
            FS_HIP(hipEventSynchronize(c->ev1));
            FS_HIP(hipEventElapsedTime(&ms_abandoned, c->ev0, c->ev1));
            c->amg_fp64_only = true;
            c->amg.reset();
            fp64_fallback = true;
            continue;
        }
        break;
    }
    const bool recurrence_converged = hs.done == 1;
    const double recurrence_rr = (use_amg && amg_rec_rr >= 0.0) ? amg_rec_rr : hs.rr;
    double true_rel = -1.0;
    if (use_amg && amg_true_rr >= 0.0 && hs.bb > 0.0) {
        true_rel = std::sqrt(amg_true_rr / hs.bb); // computed by the residual replacement of cg_amg
    } else if (recurrence_converged && rtol > 0.0 && hs.bb > 0.0) {
        // explicit residual r = b - K x (reported, not enforced): q = K x through the SpMV kernel, whose
        // input vector carries the ghost entries; the CG state is dead at this point
        launch_copy_x_to_p(m, v, st);
        rc = halo_exchange(c, v.p, st);
        if (rc) return rc;
        launch_spmv(m, v.p, v.q, nullptr, nullptr, st);
        launch_cg_init(m, v, true, st);
        rc = scalar_step(c, v, 2, CG_PHASE_RESTART, rtol);
        if (rc) return rc;
        CgScalars hv{};
        FS_HIP(hipMemcpyAsync(&hv, v.s, sizeof hv, hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        true_rel = std::sqrt(hv.rr / hv.bb);
    }
    FS_HIP(hipEventRecord(c->ev1, st));
    FS_HIP(hipMemcpyAsync(&hs, v.s, sizeof hs, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    FS_HIP(hipGetLastError());
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    ms += ms_abandoned;
    c->last_iters = hs.iters;
    c->hist_host.assign((size_t)std::min<int64_t>(hs.iters, (int64_t)c->hist.n), 0.0);
    if (!c->hist_host.empty())
        FS_HIP(hipMemcpy(c->hist_host.data(), c->hist.p, c->hist_host.size() * sizeof(double), hipMemcpyDeviceToHost));
    c->have_solution = true;
    if (info) {
        info->iterations = hs.iters;
        info->converged = recurrence_converged ? 1 : 0;
        info->rel_residual = hs.bb > 0.0 ? std::sqrt(recurrence_rr / hs.bb) : 0.0;
        info->true_rel_residual = true_rel;
        info->assemble_seconds = asm_s;
        info->setup_seconds = setup_s;
        info->solve_seconds = 1e-3 * ms;
        // (multigrid: the Krylov product, the update with its start of the smoothing, the two dot products and the new direction
        //  around the cycle: x, p, r, q read, x, r, q, z written, D^-1; z, r, q; z, p read, p written)
        info->bytes_per_iteration = use_amg ? bytes_spmv(c) + (8.0 + 3.0 + 3.0) * 48.0 * c->plan.n_own + amg_cycle_bytes(c)
                                    : single_reduction ? bytes_spmv(c) + bytes_update_single_reduction(c)
                                                       : bytes_spmv(c) + bytes_update(c) + bytes_direction(c);
        info->pc_type = c->pc.type;
        info->amg_levels = use_amg ? (int32_t)c->amg->levels.size() : 0;
        info->pc_setup_seconds = pc_setup_s;
        info->operator_complexity = 0.0;
        info->refine_passes_done = use_amg ? c->refine.passes : 0;
        info->pc_fp64_fallback = fp64_fallback ? 1 : 0;
        info->refine_correction_rel = use_amg ? c->refine.correction_rel : -1.0;
        info->refine_residual_reduction = use_amg ? c->refine.residual_reduction : 0.0;
        info->error_estimate = (use_amg && c->refine.passes > 0) ? c->refine.correction_rel * c->refine.residual_reduction : -1.0;
        if (use_amg) {
            double tot = 0.0;
            for (auto &L : c->amg->levels) tot += (double)L->nnzb;
            info->operator_complexity = tot / (double)c->amg->levels[0]->nnzb;
        }
    }
    if (hs.done < 0)
        return set_err(FEMSHELL_ERR_BREAKDOWN, "femshell_solve: CG breakdown, p.Ap <= 0 (matrix not positive definite)");
    if (u_out) return femshell_get_solution(c, u_out);
    return FEMSHELL_OK;
}

int femshell_set_initial_guess(femshell_ctx *c, const double *u0)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_initial_guess: null context");
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_initial_guess: no mesh set");
    int rc = select_device(c);
    if (rc) return rc;
    const Plan &p = c->plan;
    const size_t n6 = (size_t)p.n_pad * 6;
    if (u0 == nullptr) {
        if (!c->have_solution) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_initial_guess: no previous solve to start from");
        FS_HIP(c->x0.alloc(n6));
        FS_HIP(hipMemcpyAsync(c->x0.p, c->x.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else {
        // the rank's own rows, internal numbering
        std::vector<double> h(n6, 0.0);
        for (int32_t i = 0; i < p.n_own; i++) {
            const int32_t row = p.row_begin + i, node = c->perm.empty() ? row : c->perm[(size_t)row];
            for (int v = 0; v < 6; v++) {
                const double val = u0[6ull * (size_t)node + v];
                if (!std::isfinite(val)) return set_err(FEMSHELL_ERR_INVALID, "femshell_set_initial_guess: non-finite entry");
                h[6ull * (size_t)i + v] = val;
            }
        }
        FS_HIP(c->x0.alloc(n6));
        FS_HIP(hipMemcpyAsync(c->x0.p, h.data(), n6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream)); // (h goes out of scope)
    }
    c->warm_next = true;
    return FEMSHELL_OK;
}

int femshell_get_solution(femshell_ctx *c, double *u_out)
{
    if (!c || !u_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_get_solution: null argument");
    if (!c->have_solution) return set_err(FEMSHELL_ERR_INVALID, "femshell_get_solution: no solve has run");
    int rc = select_device(c);
    if (rc) return rc;
    const Plan &p = c->plan;
    if (!c->comm.active()) {
        if (c->perm.empty()) {
            FS_HIP(hipMemcpyAsync(u_out, c->x.p, (size_t)p.n_own * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            FS_HIP(hipStreamSynchronize(c->stream));
            return FEMSHELL_OK;
        }
        std::vector<double> h((size_t)p.n_own * 6); // internal numbering -> the caller's
        FS_HIP(hipMemcpyAsync(h.data(), c->x.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        for (int32_t i = 0; i < p.n_own; i++) std::memcpy(u_out + 6ull * c->perm[i], &h[6ull * i], 6 * sizeof(double));
        return FEMSHELL_OK;
    }
    FS_HIP(c->ufull.alloc((size_t)p.n_nodes * 6));
    std::string e;
    if (!comm_gather_rows(c->comm, c->x.p, c->ufull.p, c->all_begin, c->all_end, c->stream, &e))
        return set_err(FEMSHELL_ERR_COMM, e);
    if (c->perm.empty()) {
        FS_HIP(hipMemcpyAsync(u_out, c->ufull.p, (size_t)p.n_nodes * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FS_HIP(hipStreamSynchronize(c->stream));
        return FEMSHELL_OK;
    }
    std::vector<double> h((size_t)p.n_nodes * 6); // internal numbering -> the caller's
    FS_HIP(hipMemcpyAsync(h.data(), c->ufull.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    parallel_chunks(p.n_nodes, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) std::memcpy(u_out + 6ull * c->perm[(size_t)i], &h[6ull * (size_t)i], 6 * sizeof(double));
    }, 1 << 16);
    return FEMSHELL_OK;
}

int32_t femshell_residual_history(femshell_ctx *c, double *hist, int32_t cap)
{
    if (!c || !hist || cap < 0) return 0;
    const int32_t n = std::min<int32_t>(cap, (int32_t)c->hist_host.size());
    for (int32_t i = 0; i < n; i++) hist[i] = std::sqrt(c->hist_host[i]);
    return n;
}

int femshell_element_matrices(femshell_ctx *c, int32_t first, int32_t count, double *Ke_out)
{
    if (!c || !Ke_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_element_matrices: null argument");
    if (!c->have_mesh) return set_err(FEMSHELL_ERR_INVALID, "femshell_element_matrices: no mesh set");
    if (c->cfg.world_size != 1) return set_err(FEMSHELL_ERR_UNSUPPORTED, "femshell_element_matrices: single-rank contexts only");
    const Plan &p = c->plan;
    if (first < 0 || count < 0 || (int64_t)first + count > (int64_t)p.n_tri + p.n_quad)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_element_matrices: range out of bounds");
    if (count == 0) return FEMSHELL_OK;
    const bool quads = first >= p.n_tri;
    if (!quads && first + count > p.n_tri)
        return set_err(FEMSHELL_ERR_INVALID, "femshell_element_matrices: range mixes triangles and quads");
    int rc = select_device(c);
    if (rc) return rc;
    if (c->assembly_pending) { // (its status word is the one this call checks below)
        rc = finish_pending_assembly(c);
        if (rc) return rc;
    }
    DevBuf<double> out;
    const size_t per = quads ? 576 : 324;
    FS_HIP(out.alloc((size_t)count * per));
    launch_element_matrices(c->dm, c->mc, first, count, out.p, c->stream);
    FS_HIP(hipGetLastError());
    rc = check_status(c, "femshell_element_matrices");
    if (rc) return rc;
    FS_HIP(hipMemcpy(Ke_out, out.p, (size_t)count * per * sizeof(double), hipMemcpyDeviceToHost));
    return FEMSHELL_OK;
}

int64_t femshell_nnz_blocks(femshell_ctx *c) { return (c && c->have_mesh) ? real_blocks(c) : 0; }

int femshell_export_bsr(femshell_ctx *c, int32_t *rowptr, int32_t *colidx, double *vals, double *F)
{
    if (!c || !rowptr || !colidx || !vals) return set_err(FEMSHELL_ERR_INVALID, "femshell_export_bsr: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->matrix_valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_export_bsr: call femshell_assemble first");
    int rc = select_device(c);
    if (rc) return rc;
    const Plan &p = c->plan;
    if (F) {
        if (!c->rhs_valid) {
            rc = do_rhs(c);
            if (rc) return rc;
        }
        FS_HIP(hipMemcpyAsync(F, c->F.p, (size_t)p.n_own * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    Bsr A;
    if (c->perm.empty()) {
        rc = download_matrix(c, &A, colidx, vals);
        if (rc) return rc;
        for (int32_t a = 0; a <= p.n_own; a++) rowptr[a] = (int32_t)A.ptr[a];
        return FEMSHELL_OK;
    }
    rc = download_matrix(c, &A);
    if (rc) return rc;
    if (c->cfg.world_size > 1) {
        // a renumbered row partition: the rank's rows in the order of femshell_owned_nodes (the stretch of the internal
        // numbering it owns), columns ascending in the caller's ids; F already is in that order
        FS_HIP(hipStreamSynchronize(c->stream));
        for (int32_t a = 0; a <= p.n_own; a++) rowptr[a] = (int32_t)A.ptr[a];
        parallel_chunks(p.n_own, [&](int64_t a0, int64_t a1) {
            std::vector<std::pair<int32_t, int64_t>> order;
            for (int64_t a = a0; a < a1; a++) {
                order.clear();
                for (int64_t q = A.ptr[a]; q < A.ptr[a + 1]; q++) order.push_back({c->perm[A.col[(size_t)q]], q});
                std::sort(order.begin(), order.end());
                int64_t w = A.ptr[a];
                for (auto &o : order) {
                    colidx[w] = o.first;
                    std::memcpy(vals + 36 * w, &A.val[(size_t)o.second * 36], 36 * sizeof(double));
                    w++;
                }
            }
        });
        return FEMSHELL_OK;
    }
    // internal numbering -> the caller's: rows in the caller's order, columns ascending in the caller's ids
    if (F) {
        std::vector<double> Fi(F, F + (size_t)p.n_own * 6);
        for (int32_t i = 0; i < p.n_own; i++) std::memcpy(F + 6ull * c->perm[i], &Fi[6ull * i], 6 * sizeof(double));
    }
    rowptr[0] = 0;
    for (int32_t u = 0; u < p.n_own; u++) {
        const int32_t i = c->iperm[u];
        rowptr[u + 1] = rowptr[u] + (int32_t)(A.ptr[i + 1] - A.ptr[i]);
    }
    parallel_chunks(p.n_own, [&](int64_t u0, int64_t u1) {
        std::vector<std::pair<int32_t, int64_t>> order;
        for (int64_t u = u0; u < u1; u++) {
            const int32_t i = c->iperm[(size_t)u];
            order.clear();
            for (int64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) order.push_back({c->perm[A.col[(size_t)q]], q});
            std::sort(order.begin(), order.end());
            int64_t w = rowptr[u];
            for (auto &o : order) {
                colidx[w] = o.first;
                std::memcpy(vals + 36 * w, &A.val[(size_t)o.second * 36], 36 * sizeof(double));
                w++;
            }
        }
    });
    return FEMSHELL_OK;
}

int femshell_spmv(femshell_ctx *c, const double *x, double *y)
{
    if (!c || !x || !y) return set_err(FEMSHELL_ERR_INVALID, "femshell_spmv: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->matrix_valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_spmv: call femshell_assemble first");
    if (c->cfg.world_size != 1) return set_err(FEMSHELL_ERR_UNSUPPORTED, "femshell_spmv: single-rank contexts only");
    int rc = select_device(c);
    if (rc) return rc;
    const Plan &p = c->plan;
    DevBuf<double> dx, dy;
    FS_HIP(dx.alloc((size_t)p.n_local_nodes() * 6));
    FS_HIP(dy.alloc((size_t)p.n_pad * 6));
    FS_HIP(dx.zero(c->stream));
    std::vector<double> xi, yi; // internal numbering when the library renumbered the nodes
    if (!c->perm.empty()) {
        xi.resize((size_t)p.n_own * 6);
        yi.resize((size_t)p.n_own * 6);
        for (int32_t i = 0; i < p.n_own; i++) std::memcpy(&xi[6ull * i], x + 6ull * c->perm[i], 6 * sizeof(double));
    }
    FS_HIP(hipMemcpyAsync(dx.p, xi.empty() ? x : xi.data(), (size_t)p.n_own * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_spmv(c->dm, dx.p, dy.p, nullptr, nullptr, c->stream);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(yi.empty() ? y : yi.data(), dy.p, (size_t)p.n_own * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    for (int32_t i = 0; i < p.n_own && !yi.empty(); i++) std::memcpy(y + 6ull * c->perm[i], &yi[6ull * i], 6 * sizeof(double));
    return FEMSHELL_OK;
}

int femshell_residual(femshell_ctx *c, const double *x, double *r)
{
    if (!c || !x || !r) return set_err(FEMSHELL_ERR_INVALID, "femshell_residual: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->matrix_valid) return set_err(FEMSHELL_ERR_INVALID, "femshell_residual: call femshell_assemble first");
    if (c->cfg.world_size != 1) return set_err(FEMSHELL_ERR_UNSUPPORTED, "femshell_residual: single-rank contexts only");
    int rc = select_device(c);
    if (rc) return rc;
    if (!c->rhs_valid) {
        rc = do_rhs(c);
        if (rc) return rc;
    }
    const Plan &p = c->plan;
    DevBuf<double> dx, dr;
    FS_HIP(dx.alloc((size_t)p.n_local_nodes() * 6));
    FS_HIP(dr.alloc((size_t)p.n_pad * 6));
    FS_HIP(dx.zero(c->stream));
    std::vector<double> xi, ri;
    if (!c->perm.empty()) {
        xi.resize((size_t)p.n_own * 6);
        ri.resize((size_t)p.n_own * 6);
        for (int32_t i = 0; i < p.n_own; i++) std::memcpy(&xi[6ull * i], x + 6ull * c->perm[i], 6 * sizeof(double));
    }
    FS_HIP(hipMemcpyAsync(dx.p, xi.empty() ? x : xi.data(), (size_t)p.n_own * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_residual_dd(c->dm, dx.p, c->F.p, dr.p, c->stream);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(ri.empty() ? r : ri.data(), dr.p, (size_t)p.n_own * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    for (int32_t i = 0; i < p.n_own && !ri.empty(); i++) std::memcpy(r + 6ull * c->perm[i], &ri[6ull * i], 6 * sizeof(double));
    return FEMSHELL_OK;
}

int32_t femshell_owned_nodes(femshell_ctx *c, int32_t *ids_out)
{
    if (!c || !c->have_mesh) return 0;
    const Plan &p = c->plan;
    if (ids_out)
        for (int32_t i = 0; i < p.n_own; i++) {
            // (one rank with a renumbering: femshell_export_bsr gives the rows in the caller's order)
            const int32_t row = p.row_begin + i;
            ids_out[i] = (c->perm.empty() || c->cfg.world_size == 1) ? row : c->perm[(size_t)row];
        }
    return p.n_own;
}

int32_t femshell_row_begin(femshell_ctx *c) { return (c && c->have_mesh) ? c->plan.row_begin : 0; }
int32_t femshell_row_end(femshell_ctx *c) { return (c && c->have_mesh) ? c->plan.row_end : 0; }

int femshell_time_kernel(femshell_ctx *c, femshell_kernel which, int32_t reps, double *mean_ms_out, double *bytes_out)
{
    if (!c || !mean_ms_out) return set_err(FEMSHELL_ERR_INVALID, "femshell_time_kernel: null argument");
    if (c->assembly_pending) {
        const int prc = finish_pending_assembly(c);
        if (prc) return prc;
    }
    if (!c->have_mesh || reps <= 0) return set_err(FEMSHELL_ERR_INVALID, "femshell_time_kernel: no mesh or reps <= 0");
    int rc = select_device(c);
    if (rc) return rc;
    if (which != FEMSHELL_KERNEL_ASSEMBLE && !c->matrix_valid) {
        rc = do_assemble(c);
        if (rc) return rc;
    }
    if (which != FEMSHELL_KERNEL_ASSEMBLE && !c->rhs_valid) {
        rc = do_rhs(c);
        if (rc) return rc;
    }
    if (which != FEMSHELL_KERNEL_ASSEMBLE && !c->jacobi_valid) {
        rc = do_jacobi(c);
        if (rc) return rc;
    }
    hipStream_t st = c->stream;
    const Plan &p = c->plan;
    const size_t nrow = (size_t)p.n_pad * 6, nrow_ext = (size_t)p.n_local_nodes() * 6;
    CgVectors v;
    if (which != FEMSHELL_KERNEL_ASSEMBLE) {
        FS_HIP(c->bx.alloc(nrow));
        FS_HIP(c->br.alloc(nrow));
        FS_HIP(c->bz.alloc(nrow));
        FS_HIP(c->bq.alloc(nrow));
        FS_HIP(c->bp.alloc(nrow_ext));
        FS_HIP(c->bpart.alloc(2 * (size_t)slice_grid(c->dm)));
        FS_HIP(c->bscal.alloc(1));
        FS_HIP(c->bscal.zero(st)); // the reduction's ticket counter must start at 0 (recycled memory is not)
        v.x = c->bx.p; v.r = c->br.p; v.z = c->bz.p; v.p = c->bp.p; v.q = c->bq.p;
        v.b = c->F.p; v.partials = c->bpart.p; v.s = c->bscal.p; v.hist = nullptr; v.hist_cap = 0;
        FS_HIP(c->bp.zero(st));
        launch_cg_init(c->dm, v, false, st);                // x=0, r=b, z=M^-1 b, p=z
        launch_cg_scalar(c->dm, v, true, 2, CG_PHASE_INIT, 0.0, st);
        launch_spmv(c->dm, v.p, v.q, v.partials, v.s, st);  // q = A p
        launch_cg_scalar(c->dm, v, true, 1, CG_PHASE_ALPHA, 0.0, st);
    }
    FS_HIP(hipStreamSynchronize(st));
    double bytes = 0.0;
    if (which == FEMSHELL_KERNEL_ASSEMBLE) {
        FS_HIP(hipEventRecord(c->ev0, st));
        for (int32_t i = 0; i < reps; i++) launch_assemble(c->dm, c->mc, st);
        FS_HIP(hipEventRecord(c->ev1, st));
        FS_HIP(hipStreamSynchronize(st));
        FS_HIP(hipGetLastError());
        float ms = 0.f;
        FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *mean_ms_out = (double)ms / reps;
        bytes = bytes_assemble(c);
    } else {
        // the CG kernels are timed where they run: `reps` iterations of the local recurrence (this rank's rows, no
        // communication, no stopping test), an event pair around the chosen kernel of every iteration.  A kernel
        // launched back to back with itself finds different cache contents and measured up to 13 % faster.
        if (which != FEMSHELL_KERNEL_SPMV && which != FEMSHELL_KERNEL_CG_UPDATE && which != FEMSHELL_KERNEL_CG_DIRECTION)
            return set_err(FEMSHELL_ERR_INVALID, "femshell_time_kernel: unknown kernel");
        double sum_ms = 0.0;
        for (int32_t i = 0; i < reps; i++) {
            // (as cg_classic runs them: with symmetric storage the SpMV is its first phase and the update kernel
            // collects the transposed products)
            if (which == FEMSHELL_KERNEL_SPMV) FS_HIP(hipEventRecord(c->ev0, st));
            if (c->dm.symmetric) launch_spmv_direct(c->dm, v.p, v.q, v.partials, v.s, st);
            else launch_spmv(c->dm, v.p, v.q, v.partials, v.s, st);
            if (which == FEMSHELL_KERNEL_SPMV) FS_HIP(hipEventRecord(c->ev1, st));
            launch_cg_scalar(c->dm, v, true, 1, CG_PHASE_ALPHA, 0.0, st);
            if (which == FEMSHELL_KERNEL_CG_UPDATE) FS_HIP(hipEventRecord(c->ev0, st));
            launch_cg_update(c->dm, v, st, c->dm.symmetric != 0);
            if (which == FEMSHELL_KERNEL_CG_UPDATE) FS_HIP(hipEventRecord(c->ev1, st));
            launch_cg_scalar(c->dm, v, true, 2, CG_PHASE_BETA, 0.0, st);
            if (which == FEMSHELL_KERNEL_CG_DIRECTION) FS_HIP(hipEventRecord(c->ev0, st));
            launch_cg_direction(c->dm, v, st);
            if (which == FEMSHELL_KERNEL_CG_DIRECTION) FS_HIP(hipEventRecord(c->ev1, st));
            FS_HIP(hipStreamSynchronize(st));
            float ms = 0.f;
            FS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            sum_ms += ms;
        }
        FS_HIP(hipGetLastError());
        *mean_ms_out = sum_ms / reps;
        bytes = which == FEMSHELL_KERNEL_SPMV ? bytes_spmv(c) : which == FEMSHELL_KERNEL_CG_UPDATE ? bytes_update(c) : bytes_direction(c);
    }
    if (bytes_out) *bytes_out = bytes;
    if (which == FEMSHELL_KERNEL_ASSEMBLE) {
        rc = check_status(c, "femshell_time_kernel");
        if (rc) return rc;
        c->matrix_valid = true;
        c->jacobi_valid = false;
    }
    return FEMSHELL_OK;
}

int femshell_sync(femshell_ctx *c)
{
    if (!c) return set_err(FEMSHELL_ERR_INVALID, "femshell_sync: null context");
    int rc = select_device(c);
    if (rc) return rc;
    if (c->assembly_pending) return finish_pending_assembly(c); // (synchronises)
    FS_HIP(hipStreamSynchronize(c->stream));
    return FEMSHELL_OK;
}

} // extern "C"
