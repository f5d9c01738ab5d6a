// context.hpp -- what the translation units of libfemshell share: error reporting, device buffers, the context
// behind a femshell_ctx handle and the CG driver's entry points (cg_driver.cpp).
#pragma once
#include <cstdio>

#include "femshell.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <iterator>
#include <list>
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

// Device memory of the library's buffers.
// (1) ARENAS (round 6).  A multigrid setup at 4M triangles makes about 150 allocations, and hipMalloc costs 0.5 - 2.5 ms a call on
//     the boxes of this pool whatever the size: 60 to 370 ms of a 0.18 - 0.5 s setup were the driver's allocator
//     (FEMSHELL_AMG_VERBOSE=1 prints the count).  Requests below kArenaMax are therefore carved out of arenas of kArenaBytes (one
//     hipMalloc each, bump pointer, 256-byte granules); larger ones -- K's values, the big operators -- go to the driver as before.
// (2) REUSE.  Blocks a buffer gives back are kept (per device) and handed to the next request of about their size: a setup
//     allocates and frees about 10 GB of transient operators (P, A P, R, A_c, Q), every setup after the first of a process -- a
//     changed K in a coupled run, the next context of a test process -- finds them here.  Directly allocated blocks of 1 MiB and
//     more are kept up to FEMSHELL_POOL_GB gigabytes (default 24; 0 = no pool and no arenas), pieces of arenas always (they cannot go
//     back to the driver one by one).  A block is reused only after the device has gone idle once since it came back (hipFree
//     synchronises too); everything goes back to the driver when the last context of the process is destroyed.
class DevPool {
  public:
    static DevPool &get()
    {
        static DevPool pool;
        return pool;
    }
    // FEMSHELL_POOL_POISON=1 (tests, debugging): every block is filled with 0xFF bytes -- NaNs as doubles and floats, -1 as
    // integers -- before it is handed out.  Fresh memory from the driver is zero and recycled memory is not: whoever relies on
    // zeros it never wrote works until the pool hands it a used block (found that way: round 6, the carved blocks).
    hipError_t alloc(void **out, size_t bytes)
    {
        const hipError_t e = alloc_raw(out, bytes);
        if (e == hipSuccess && poison_ && *out != nullptr && bytes > 0) {
            (void)hipMemset(*out, 0xFF, bytes);
            (void)hipDeviceSynchronize();
        }
        return e;
    }
    hipError_t alloc_raw(void **out, size_t bytes)
    {
        *out = nullptr;
        int dev = 0;
        if (limit_ > 0) (void)hipGetDevice(&dev);
        const bool small = limit_ > 0 && arena_bytes_ > 0 && bytes < kArenaMax;
        const size_t need = small ? (bytes + kGranule - 1) / kGranule * kGranule : bytes;
        if (limit_ > 0 && (small || bytes >= kMinBytes)) {
            std::lock_guard<std::mutex> lock(m_);
            // best fit: the smallest kept block of this device that holds the request without wasting more than a quarter
            auto it = free_.lower_bound(Key{dev, need});
            if (it != free_.end() && it->first.dev == dev && it->first.bytes <= need + need / 4 + (small ? kGranule : 0)) {
                *out = it->second;
                live_[*out] = it->first.bytes;
                if (!in_arena(*out)) cached_ -= it->first.bytes;
                else arena_of(*out)->live++;
                free_.erase(it);
                return hipSuccess;
            }
            if (small) {
                Arena *a = nullptr;
                for (auto &ar : arenas_)
                    if (ar.dev == dev && ar.used + need <= ar.bytes) {
                        a = &ar;
                        break;
                    }
                if (a == nullptr && carve_) // (a kept block of the driver's that holds the request becomes the next arena: alloc (3) below)
                    for (auto jt = free_.lower_bound(Key{dev, need}); jt != free_.end() && jt->first.dev == dev; ++jt) {
                        if (in_arena(jt->second)) continue;
                        arenas_.push_back(Arena{dev, static_cast<char *>(jt->second), jt->first.bytes, 0, 0, true});
                        cached_ -= jt->first.bytes;
                        free_.erase(jt);
                        a = &arenas_.back();
                        break;
                    }
                if (a == nullptr) {
                    void *base = nullptr;
                    const size_t ab = arena_bytes_;
                    const auto t0 = std::chrono::steady_clock::now();
                    const hipError_t e = hipMalloc(&base, ab);
                    stat_malloc_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                    stat_mallocs_++;
                    if (getenv("FEMSHELL_POOL_VERBOSE")) fprintf(stderr, "[femshell pool] hipMalloc of an arena, %zu MB (for a request of %zu bytes)\n", ab >> 20, bytes);
                    if (e == hipSuccess) {
                        arenas_.push_back(Arena{dev, static_cast<char *>(base), ab, 0, 0, false});
                        a = &arenas_.back();
                    } else {
                        (void)hipGetLastError(); // (no room for another arena: the request goes to the driver on its own, below)
                    }
                }
                if (a != nullptr) {
                    *out = a->base + a->used;
                    a->used += need;
                    a->live++;
                    live_[*out] = need;
                    return hipSuccess;
                }
            } else if (arena_bytes_ > 0 && carve_) {
                // (3) CARVING (round 6).  A large request that no kept block fits within a quarter: (a) room in a block that was
                // carved before, else (b) the smallest kept block that holds it at all becomes an arena and the request its first
                // piece; the rest serves the requests that follow.  A multigrid setup ends with the single-precision copies -- 1.1,
                // 0.8, 0.7 GB at 4M triangles -- right after A P (3.1 GB) has come back: five hipMalloc calls and 2.6 GB of fresh
                // memory less, which is 35-50 ms on a box whose allocator is in its slow state (profiles/r06_hipmalloc_probe.txt).
                const size_t need_g = (bytes + kGranule - 1) / kGranule * kGranule;
                Arena *a = nullptr;
                for (auto &ar : arenas_)
                    if (ar.carved && ar.dev == dev && ar.used + need_g <= ar.bytes) {
                        a = &ar;
                        break;
                    }
                if (a == nullptr)
                    for (auto jt = free_.lower_bound(Key{dev, need_g}); jt != free_.end() && jt->first.dev == dev; ++jt) {
                        if (in_arena(jt->second)) continue;
                        arenas_.push_back(Arena{dev, static_cast<char *>(jt->second), jt->first.bytes, 0, 0, true});
                        cached_ -= jt->first.bytes;
                        free_.erase(jt);
                        a = &arenas_.back();
                        break;
                    }
                if (a != nullptr) {
                    *out = a->base + a->used;
                    a->used += need_g;
                    a->live++;
                    live_[*out] = need_g;
                    return hipSuccess;
                }
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(out, bytes);
        stat_malloc_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        stat_mallocs_++;
        if (getenv("FEMSHELL_POOL_VERBOSE"))
            fprintf(stderr, "[femshell pool] hipMalloc %.1f MB: %.2f ms\n", (double)bytes / 1048576.0,
                    1e-6 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
        if (e != hipSuccess) { // out of memory with blocks kept (or waiting for the end of a setup): give them back and try once more
            const bool waited = flush_pending(), kept = trim();
            if (waited || kept) {
                (void)hipGetLastError();
                e = hipMalloc(out, bytes);
            }
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
        bool arena = false;
        {
            std::lock_guard<std::mutex> lock(m_);
            auto it = live_.find(p);
            if (it != live_.end()) {
                bytes = it->second;
                live_.erase(it);
            }
            arena = in_arena(p);
        }
        const auto t0 = std::chrono::steady_clock::now();
        // (a block too large to keep, released inside a multigrid setup -- A P of a 32M-triangle mesh, 26 GB -- waits with the others:
        //  freed at once, the setup's NEXT requests -- the single-precision copies -- waited for the driver to wipe it, 0.65 s of
        //  hipMalloc in a 1.05 s setup; kept until the setup is over it is what those requests are carved from)
        const bool deferred = defer_depth_.load() > 0 && bytes > 0;
        if (!deferred && !arena && (bytes == 0 || bytes > limit_)) {
            (void)hipFree(p);
            stat_free_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            stat_frees_++;
            return;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (deferred) { // (a multigrid setup in progress: see Defer below)
            std::lock_guard<std::mutex> lock(m_);
            pending_.push_back(Pending{p, bytes, arena, dev});
            return;
        }
        (void)hipDeviceSynchronize(); // (what hipFree does: nothing in flight reads or writes the block any more)
        stat_sync_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        stat_syncs_++;
        std::lock_guard<std::mutex> lock(m_);
        keep_block(p, bytes, arena, dev);
    }
    // While a Defer object lives, blocks that come back to the pool wait in a list instead of synchronising the device one by one
    // (the power iteration of a multigrid setup runs on the second stream from the setup's first moment; a temporary buffer of the
    // search for clusters, released 1 ms in, used to wait 22 ms for it).  They are not handed out again before the last Defer object
    // is gone: one device synchronisation then, and they join the kept blocks.
    struct Defer {
        Defer() { DevPool::get().defer_depth_.fetch_add(1); }
        ~Defer()
        {
            if (DevPool::get().defer_depth_.fetch_sub(1) == 1) {
                (void)DevPool::get().flush_pending();
                DevPool::get().enforce_limit();
            }
        }
        Defer(const Defer &) = delete;
        Defer &operator=(const Defer &) = delete;
    };
    // (when the last Defer object is gone: kept blocks beyond the pool's limit go back to the driver, largest first)
    void enforce_limit()
    {
        std::lock_guard<std::mutex> lock(m_);
        trim_to_limit();
    }
    bool flush_pending()
    {
        std::vector<Pending> list;
        {
            std::lock_guard<std::mutex> lock(m_);
            list.swap(pending_);
        }
        if (list.empty()) return false;
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipDeviceSynchronize();
        stat_sync_ns_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        stat_syncs_ += list.size();
        std::lock_guard<std::mutex> lock(m_);
        for (const Pending &b : list) keep_block(b.p, b.bytes, b.arena, b.dev);
        return true;
    }

  private:
    struct Pending {
        void *p;
        size_t bytes;
        bool arena;
        int dev;
    };
    // (callers hold m_; the device has been idle since the block came back)
    void keep_block(void *p, size_t bytes, bool arena, int dev)
    {
        if (arena) {
            Arena *a = arena_of(p);
            a->live--;
            if (a->live == 0) { // nothing of it in use: the arena starts over (its kept pieces leave the free list)
                for (auto it = free_.begin(); it != free_.end();)
                    if (arena_of(it->second) == a) it = free_.erase(it);
                    else ++it;
                a->used = 0;
                return;
            }
            free_.emplace(Key{a->dev, bytes}, p);
            return;
        }
        free_.emplace(Key{dev, bytes}, p);
        cached_ += bytes;
        if (defer_depth_.load() > 0) return; // (inside a setup the kept blocks may exceed the limit: enforce_limit() when it is over)
        trim_to_limit();
    }
    void trim_to_limit()
    {
        while (cached_ > limit_) { // over the limit: the largest kept block that is the driver's goes back to it
            auto big = free_.end();
            for (auto it = free_.end(); it != free_.begin();) {
                --it;
                if (!in_arena(it->second)) {
                    big = it;
                    break;
                }
            }
            if (big == free_.end()) break;
            cached_ -= big->first.bytes;
            (void)hipFree(big->second);
            free_.erase(big);
        }
    }

  public:
    // every kept block back to the driver, and every arena nothing of which is in use; true when there was one
    bool trim()
    {
        std::lock_guard<std::mutex> lock(m_);
        bool any = false;
        for (auto it = free_.begin(); it != free_.end();) {
            if (in_arena(it->second)) {
                ++it;
                continue;
            }
            (void)hipFree(it->second);
            it = free_.erase(it);
            any = true;
        }
        cached_ = 0;
        for (auto ar = arenas_.begin(); ar != arenas_.end();) {
            if (ar->live != 0) {
                ++ar;
                continue;
            }
            for (auto it = free_.begin(); it != free_.end();)
                if (it->second >= ar->base && it->second < ar->base + ar->bytes) it = free_.erase(it);
                else ++it;
            (void)hipFree(ar->base);
            ar = arenas_.erase(ar);
            any = true;
        }
        return any;
    }
    void context_opened() { contexts_.fetch_add(1); }
    void context_closed()
    {
        if (contexts_.fetch_sub(1) == 1) (void)trim();
    }
    // (FEMSHELL_AMG_VERBOSE: what the driver's allocator cost since the last call -- hipMalloc, hipFree, the synchronisations of
    //  blocks that went back to the pool)
    void stats(double out[6])
    {
        out[0] = (double)stat_mallocs_.exchange(0);
        out[1] = 1e-6 * (double)stat_malloc_ns_.exchange(0);
        out[2] = (double)stat_frees_.exchange(0);
        out[3] = 1e-6 * (double)stat_free_ns_.exchange(0);
        out[4] = (double)stat_syncs_.exchange(0);
        out[5] = 1e-6 * (double)stat_sync_ns_.exchange(0);
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
    struct Arena {
        int dev;
        char *base;
        size_t bytes, used;
        int64_t live; // pieces in use
        bool carved = false; // a kept block turned into an arena (alloc (3)); large requests are served from these only
    };
    static constexpr size_t kMinBytes = 1u << 20;
    static constexpr size_t kGranule = 256;
    static constexpr size_t kArenaMax = 96u << 20; // requests below this come out of an arena
    DevPool()
    {
        const char *e = getenv("FEMSHELL_POOL_GB");
        const double gb = e ? atof(e) : 24.0;
        limit_ = gb > 0.0 ? (size_t)(gb * 1073741824.0) : 0;
        poison_ = getenv("FEMSHELL_POOL_POISON") && atoi(getenv("FEMSHELL_POOL_POISON")) != 0;
        carve_ = !(getenv("FEMSHELL_POOL_CARVE") && atoi(getenv("FEMSHELL_POOL_CARVE")) == 0); // (0: no carving of kept blocks, A/B runs)
        const char *a = getenv("FEMSHELL_POOL_ARENA_MB"); // 0: no arenas (every request to the driver, as in round 5)
        const double mb = a ? atof(a) : 512.0;
        arena_bytes_ = mb > 0.0 ? std::max((size_t)(mb * 1048576.0), 2 * kArenaMax) : 0;
    }
    // (callers hold m_)
    Arena *arena_of(const void *p)
    {
        const char *q = static_cast<const char *>(p);
        for (auto &a : arenas_)
            if (q >= a.base && q < a.base + a.bytes) return &a;
        return nullptr;
    }
    bool in_arena(const void *p) { return arena_of(p) != nullptr; }
    std::mutex m_;
    std::multimap<Key, void *> free_;
    std::unordered_map<void *, size_t> live_;
    std::list<Arena> arenas_; // (a list: the pointers into it stay valid)
    size_t cached_ = 0, limit_ = 0, arena_bytes_ = 0;
    bool poison_ = false, carve_ = true;
    std::atomic<int> contexts_{0}, defer_depth_{0};
    std::vector<Pending> pending_;
    std::atomic<uint64_t> stat_mallocs_{0}, stat_malloc_ns_{0}, stat_frees_{0}, stat_free_ns_{0}, stat_syncs_{0}, stat_sync_ns_{0};
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
    // Two regions of 32 MB of pinned host memory through which the multigrid setup moves its arrays (staged_download / staged_upload
    // in amg_device_setup.cpp; region 0: the calling thread, region 1: the setup's helper thread).  Why not plain copies from and to
    // the arrays themselves: the runtime registers the pages of a large pageable copy with the driver, and a host array that is
    // FREED soon after -- the 48 MB of node normals, released when the first coarsening step is through -- takes that registration
    // down through the MMU notifier, which evicts the process's queues: the next launch on the main stream waited 24-31 ms (found
    // with FEMSHELL_AMG_VERBOSE laps around the block-Jacobi kernel of level 1; gone with the array kept, or copied through here).
    // nullptr when the allocation failed: the copies then go the pageable way.
    void *stage_host = nullptr;
    size_t stage_bytes = 0; // of ONE region
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
    // uploads a helper thread of the multigrid setup makes beside the main stream (the node normals, amg_solve.cpp): made by that
    // thread at its first use in a single-rank context and kept.  (Not at femshell_create: HIP maps streams onto four hardware queues,
    // and one stream more per context was enough, with two ranks sharing a card in the tests, for the look-ahead of the dense
    // inverse -- two streams that must run side by side -- to run into its bounded wait one time in four.)
    hipStream_t copy_stream = nullptr;
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
