// context.hpp -- what the translation units of libfemshell share: error reporting, device buffers, the context
// behind a femshell_ctx handle and the CG driver's entry points (cg_driver.cpp).
#pragma once

#include "femshell.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "comm.hpp"
#include "errors.hpp"
#include "kernels.hpp"
#include "plan.hpp"

namespace femshell {

struct Amg; // multigrid hierarchy (amg_device.hpp)

inline const char *fs_basename(const char *path)
{
    const char *b = path;
    for (const char *q = path; *q; q++)
        if (*q == '/') b = q + 1;
    return b;
}

#define FS_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_err(FEMSHELL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_) +      \
                                                 " (" + fs_basename(__FILE__) + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// Device memory of the library's buffers.  Blocks of 1 MiB and more that a buffer gives back are kept (per device, at most
// FEMSHELL_POOL_GB gigabytes, default 24; 0 = no pool) and handed to the next request of about their size instead of going
// through hipFree / hipMalloc again: a multigrid setup at 4M triangles allocates and frees about 10 GB of transient operators
// (P, A P, R, A_c, Q), every setup after the first of a process -- a changed K in a coupled run, the next context of a test
// process -- then finds them here, and the card is not left churned for whoever allocates next (on a churned
// card hipMalloc made a 0.18 s setup take 0.5 s).  A block is reused only after the device has gone idle once since it came
// back (hipFree synchronises too); the pool empties when the last context of the process is destroyed.
class DevPool {
  public:
    static DevPool &get()
    {
        static DevPool pool;
        return pool;
    }
    hipError_t alloc(void **out, size_t bytes)
    {
        *out = nullptr;
        if (bytes >= kMinBytes && limit_ > 0) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lock(m_);
            // best fit: the smallest kept block of this device that holds the request without wasting more than a quarter
            auto it = free_.lower_bound(Key{dev, bytes});
            if (it != free_.end() && it->first.dev == dev && it->first.bytes <= bytes + bytes / 4) {
                *out = it->second;
                live_[*out] = it->first.bytes;
                cached_ -= it->first.bytes;
                free_.erase(it);
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess && trim()) { // out of memory with blocks kept: give them back and try once more
            (void)hipGetLastError();
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess && bytes >= kMinBytes && limit_ > 0) {
            std::lock_guard<std::mutex> lock(m_);
            live_[*out] = bytes;
        }
        return e;
    }
    void free(void *p)
    {
        if (p == nullptr) return;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> lock(m_);
            auto it = live_.find(p);
            if (it != live_.end()) {
                bytes = it->second;
                live_.erase(it);
            }
        }
        if (bytes == 0 || bytes > limit_) {
            (void)hipFree(p);
            return;
        }
        (void)hipDeviceSynchronize(); // (what hipFree does: nothing in flight reads or writes the block any more)
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(m_);
        free_.emplace(Key{dev, bytes}, p);
        cached_ += bytes;
        while (cached_ > limit_ && !free_.empty()) { // over the limit: the largest kept block goes back to the driver
            auto big = std::prev(free_.end());
            cached_ -= big->first.bytes;
            (void)hipFree(big->second);
            free_.erase(big);
        }
    }
    // every kept block back to the driver; true when there was one
    bool trim()
    {
        std::lock_guard<std::mutex> lock(m_);
        const bool any = !free_.empty();
        for (auto &kv : free_) (void)hipFree(kv.second);
        free_.clear();
        cached_ = 0;
        return any;
    }
    void context_opened() { contexts_.fetch_add(1); }
    void context_closed()
    {
        if (contexts_.fetch_sub(1) == 1) (void)trim();
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> lock(m_);
        return cached_;
    }

  private:
    struct Key {
        int dev;
        size_t bytes;
        bool operator<(const Key &o) const { return dev != o.dev ? dev < o.dev : bytes < o.bytes; }
    };
    static constexpr size_t kMinBytes = 1u << 20;
    DevPool()
    {
        const char *e = getenv("FEMSHELL_POOL_GB");
        const double gb = e ? atof(e) : 24.0;
        limit_ = gb > 0.0 ? (size_t)(gb * 1073741824.0) : 0;
    }
    std::mutex m_;
    std::multimap<Key, void *> free_;
    std::unordered_map<void *, size_t> live_;
    size_t cached_ = 0, limit_ = 0;
    std::atomic<int> contexts_{0};
};

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) DevPool::get().free(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count)
    {
        if (count == n && p) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = DevPool::get().alloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    template <class Vec> hipError_t upload(const Vec &h, hipStream_t st) // any contiguous container of T
    {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
    }
    hipError_t zero(hipStream_t st) { return n ? hipMemsetAsync(p, 0, n * sizeof(T), st) : hipSuccess; }
};

} // namespace femshell

struct femshell_ctx;
// zero the status word after a failure was read (mapped: by the host, behind a stream synchronisation)
int clear_status_word(femshell_ctx *c, hipStream_t st);
// the status word on the host after everything enqueued on st (mapped: no copy)
int fetch_status_word(femshell_ctx *c, hipStream_t st, int32_t *out);

struct femshell_ctx {
    // (global namespace: the opaque type of include/femshell.h)
    femshell_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int32_t *status_host = nullptr; // pinned landing place of the device status word
    // the status word as the kernels see it: c->status in HBM, or -- FEMSHELL_STATUS_MAPPED, the default -- the device address of
    // status_host itself (kernels write the word on failure only, with a system-scope compare-and-swap; the host reads it after the
    // stream synchronisation without a copy: 12 us of every femshell_assemble)
    int32_t *status_word = nullptr;
    bool status_mapped = false;
    bool asm_events = true; // FEMSHELL_ASM_EVENTS=0: no event pair around the assembly kernel (assemble_seconds by the host's clock)
    double *agree_host = nullptr;   // pinned word of the cross-rank agreement on a rank-local failure (api.cpp)
    femshell::DevBuf<double> agree;
    // halo exchange beside the interior SpMV (multi-rank contexts): second stream + hand-off events
    hipStream_t halo_stream = nullptr;
    // femshell_set_initial_guess: the iterate the NEXT femshell_solve starts from (owned rows, internal numbering), consumed by it
    femshell::DevBuf<double> x0;
    bool warm_next = false;
    hipStream_t aux_stream = nullptr; // the look-ahead of the dense inverse (amg_dense.hip)
    int aux_streams_side_by_side = 0; // 0: not asked yet, 1: stream and aux_stream run concurrently, -1: they share a hardware queue
    // first-contact self-test of femshell_comm_init (comm.cpp comm_selftest): microseconds of its three patterns
    double comm_selftest_us[3] = {-1.0, -1.0, -1.0};
    bool comm_selftest_done = false;
    hipEvent_t ev_p_ready = nullptr, ev_halo_done = nullptr;
    bool halo_overlap = false;
    femshell::MatConst mc{};
    femshell::Plan plan;
    bool have_mesh = false, matrix_valid = false, rhs_valid = false, jacobi_valid = false, have_solution = false;

    // optional renumbering (reorder.cpp): internal node i is the caller's node perm[i]; iperm is the inverse; both empty
    // when the caller's numbering is kept.  dmask_global / loads_global and everything below are in internal numbering.
    std::vector<int32_t> perm, iperm;
    std::vector<uint8_t> dmask_global;  // n_nodes
    femshell::RawVec<double> loads_global; // n_nodes*6

    femshell::DevBuf<double> xyz, vals, minv, loads, F;
    femshell::DevBuf<int32_t> tri, quad, slice_width, cols, pair_ptr, status;
    femshell::DevBuf<int64_t> slice_base, in_base;
    femshell::DevBuf<int32_t> in_width, in_slots, in_rows; // symmetric storage: per-row lists of transposed blocks
    femshell::DevBuf<int32_t> gat_slots;                    // ... without the blocks whose product stays inside the slice
    femshell::DevBuf<uint8_t> loc_index, loc_list;
    femshell::DevBuf<double> tbuf;                          // ... and their products K_ac^T x_a (6 doubles per slot)
    femshell::DevBuf<int32_t> slice_elem_ptr, slice_elem_nodes, item_ptr, slice_desc;
    femshell::DevBuf<femshell::Plan::Item> items;
    femshell::DevBuf<uint32_t> item_flags;
    femshell::DevBuf<uint8_t> dmask;
    // CG state
    femshell::DevBuf<double> x, r, z, p, q, sv, partials, hist, sendbuf, ufull;
    femshell::DevBuf<double> xacc, rres; // iterative refinement of the multigrid-preconditioned solve: accumulated solution, residual
    femshell::DevBuf<femshell::CgScalars> scal;
    femshell::DevBuf<int32_t> send_nodes, spmv_order;
    std::vector<int32_t> send_offsets; // per peer, in nodes
    // scratch for femshell_time_kernel
    femshell::DevBuf<double> bx, br, bz, bp, bq, bpart;
    femshell::DevBuf<femshell::CgScalars> bscal;

    femshell::DeviceMatrix dm{};
    femshell::Comm comm;
    std::vector<int32_t> all_begin, all_end;

    femshell_pc_options pc{};           // preconditioner of femshell_solve (block-Jacobi unless set otherwise)
    // the multigrid hierarchy keeps everything FP64 (no single-precision copies, vectors or coarsest inverse): set by
    // femshell_solve after a breakdown of the flexible CG with a hierarchy that used them -- the rounded preconditioner of a
    // very thin shell is not positive definite any more --, cleared by femshell_set_mesh / femshell_set_preconditioner
    bool amg_fp64_only = false;
    std::shared_ptr<femshell::Amg> amg; // hierarchy of the multigrid preconditioner, rebuilt when K changes
    // contexts with a communicator: the centre of the WHOLE mesh, about which the rigid-body modes of the row-partitioned
    // multigrid turn on every rank alike (amg_dist.cpp; a rank's plan holds its own rows and their ghosts only)
    double mesh_centre[3] = {0.0, 0.0, 0.0};
    bool have_mesh_centre = false;

    // error estimate of the last multigrid-preconditioned solve (femshell_solve_info, cg_amg)
    struct RefineStats {
        int32_t passes = 0;
        double correction_rel = -1.0, residual_reduction = 0.0;
    } refine;
    femshell::DevBuf<double> dots_scratch;

    double last_assemble_s = 0.0, last_setup_s = 0.0;
    bool assembly_pending = false; // femshell_assemble_async: status word and timing of the last assembly not collected yet
    std::vector<double> hist_host;
    int32_t last_iters = 0;
};

namespace femshell {

CgVectors cg_vectors(femshell_ctx *c);
// pack + grouped send/recv of the ghost entries of vec (owned | padding | ghosts) on stream st
int halo_exchange(femshell_ctx *c, double *vec, hipStream_t st);
// yout = K xin with the halo exchange beside the interior slices (xin carries ghost space); partial sums of xin.yout when
// partials != nullptr (*n_partials of them, 0 = slice_grid); defer_gather: symmetric storage, the caller's next kernel
// collects the transposed products
// (vals32: the products read this single-precision copy of K's values -- smoothing products of the multigrid cycle)
int spmv_with_halo(femshell_ctx *c, const CgVectors &v, double *xin, double *yout, double *partials, int *n_partials,
                   bool defer_gather = false, const float *vals32 = nullptr, int vec32 = 0);
// the two recurrences (cg_driver.cpp); the CG state is left in the context's vectors and scalars
int cg_classic(femshell_ctx *c, const CgVectors &v, double rtol, int32_t max_it, const double *x0 = nullptr);
int cg_single_reduction(femshell_ctx *c, const CgVectors &v, double rtol, int32_t max_it);
bool use_single_reduction(const femshell_ctx *c);
// reduction of the partial sums [+ all-reduce on contexts with a communicator] + scalar step
int scalar_step(femshell_ctx *c, const CgVectors &v, int nsums, CgPhase phase, double rtol, int n_partials = 0, int len3 = 0,
                int gate_phase = -1);

} // namespace femshell
