// comm.cpp -- RCCL wiring (see comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

#include <unistd.h>

namespace femshell {

namespace {

struct Api {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Api g_api;

std::mutex g_api_mutex; // distinct contexts are thread-safe (include/femshell.h): two threads may get here together

bool load_api(std::string *err)
{
    std::lock_guard<std::mutex> lock(g_api_mutex);
    if (g_api.lib) return true;
    // FEMSHELL_RCCL_LIB: tests point this at tests/helpers/fake_rccl (several ranks on one GPU)
    const char *override_path = getenv("FEMSHELL_RCCL_LIB");
    void *lib = override_path ? dlopen(override_path, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib && override_path) {
        if (err) *err = std::string("cannot open FEMSHELL_RCCL_LIB=") + override_path + ": " + dlerror();
        return false;
    }
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        if (err) *err = std::string("cannot open librccl: ") + dlerror();
        return false;
    }
    Api a;
    a.lib = lib;
#define FS_SYM(name)                                                              \
    a.name = reinterpret_cast<decltype(a.name)>(dlsym(lib, "nccl" #name));         \
    if (!a.name) {                                                                \
        if (err) *err = "librccl lacks nccl" #name;                               \
        return false;                                                             \
    }
    FS_SYM(GetUniqueId)
    FS_SYM(CommInitRank)
    FS_SYM(CommDestroy)
    FS_SYM(CommCount)
    FS_SYM(AllReduce)
    FS_SYM(Broadcast)
    FS_SYM(Send)
    FS_SYM(Recv)
    FS_SYM(GroupStart)
    FS_SYM(GroupEnd)
    FS_SYM(GetErrorString)
#undef FS_SYM
    g_api = a;
    return true;
}

bool check(ncclResult_t r, const char *what, std::string *err)
{
    if (r == ncclSuccess) return true;
    if (err) *err = std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
    return false;
}

} // namespace

// ---- watchdog (comm.hpp) ---------------------------------------------------------------------------------------
// Every live CommWatch is an entry of a registry the monitor thread walks: contexts on different threads have their own
// phases and timers (distinct contexts are thread-safe, include/femshell.h).  The time limit is read once, by the thread that
// opens the watch -- the monitor thread never calls getenv, which would race with a host that changes its environment.
namespace {

int64_t now_ms()
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct WatchEntry {
    const char *phase = nullptr;
    int rank = 0, world = 1;
    double limit_s = 0.0;
    std::atomic<int64_t> since_ms{0};
    WatchEntry *parent = nullptr; // the watch that was open on this thread when this one began
};

std::mutex g_watch_mutex;
std::vector<WatchEntry *> g_watch_live;
std::once_flag g_watch_once;
thread_local WatchEntry *t_watch_top = nullptr; // innermost watch of the calling thread

void watch_loop()
{
    for (;;) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        const int64_t now = now_ms();
        std::lock_guard<std::mutex> lock(g_watch_mutex);
        for (const WatchEntry *w : g_watch_live) {
            if (!(w->limit_s > 0.0)) continue;
            const double waited = 1e-3 * (double)(now - w->since_ms.load(std::memory_order_acquire));
            if (waited < w->limit_s) continue;
            fprintf(stderr,
                    "[femshell watchdog] rank %d of %d: no progress for %.0f s in \"%s\" -- a peer rank never joined or stalled in a "
                    "collective.  Exiting with status 86 so that the launcher ends the rank group.  To look further: NCCL_DEBUG=WARN for "
                    "RCCL's own diagnostics, FEMSHELL_HALO_OVERLAP=0 to take the halo exchange off its second stream, "
                    "FEMSHELL_COMM_TIMEOUT=<seconds> (0 = wait forever).\n",
                    w->rank, w->world, waited, w->phase);
            fflush(stderr);
            _exit(86);
        }
    }
}

} // namespace

CommWatch::CommWatch(int rank, int world, const char *phase) : entry_(nullptr)
{
    if (world <= 1) return;
    const char *e = getenv("FEMSHELL_COMM_TIMEOUT"); // once per watch, on the caller's thread (tests shorten it between calls)
    WatchEntry *w = new WatchEntry();
    w->phase = phase;
    w->rank = rank;
    w->world = world;
    w->limit_s = e ? atof(e) : 120.0;
    w->since_ms.store(now_ms(), std::memory_order_release);
    w->parent = t_watch_top;
    t_watch_top = w;
    entry_ = w;
    std::call_once(g_watch_once, [] { std::thread(watch_loop).detach(); });
    std::lock_guard<std::mutex> lock(g_watch_mutex);
    g_watch_live.push_back(w);
}

CommWatch::~CommWatch()
{
    WatchEntry *w = static_cast<WatchEntry *>(entry_);
    if (w == nullptr) return;
    {
        std::lock_guard<std::mutex> lock(g_watch_mutex);
        for (size_t i = 0; i < g_watch_live.size(); i++)
            if (g_watch_live[i] == w) {
                g_watch_live.erase(g_watch_live.begin() + (long)i);
                break;
            }
    }
    t_watch_top = w->parent;
    delete w;
    heartbeat(); // (the outer phases of this thread start over)
}

// progress inside a phase: every watch open on the calling thread counts from here again
void CommWatch::heartbeat()
{
    const int64_t now = now_ms();
    for (WatchEntry *w = t_watch_top; w != nullptr; w = w->parent) w->since_ms.store(now, std::memory_order_release);
}

bool comm_unique_id(uint8_t id_out[128], std::string *err)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!load_api(err)) return false;
    ncclUniqueId id;
    if (!check(g_api.GetUniqueId(&id), "ncclGetUniqueId", err)) return false;
    std::memcpy(id_out, &id, 128);
    return true;
}

bool comm_init(Comm &c, const uint8_t id_bytes[128], int rank, int world, std::string *err)
{
    if (!load_api(err)) return false;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, 128);
    ncclComm_t comm = nullptr;
    {
        // every rank must arrive: a missing one leaves the others here for good (comm.hpp: CommWatch)
        CommWatch watch(rank, world, "ncclCommInitRank (waiting for all ranks of the row partition to join)");
        if (!check(g_api.CommInitRank(&comm, world, id, rank), "ncclCommInitRank", err)) return false;
    }
    c.lib = g_api.lib;
    c.comm = comm;
    c.rank = rank;
    c.world = world;
    return true;
}

// First-contact self-test (comm.hpp): the communication patterns of a solve, once each with a known answer, each under its own
// watch so that a transport that initialises but cannot carry one of them is named here and not as a hang in the first solve.
//  1. a grouped ncclSend / ncclRecv ring on `second` (every rank to its successor: the halo pattern on the halo stream) while an
//     ncclAllReduce of three words is enqueued on `main` (the CG's reduction on the main stream) -- two streams, one communicator
//  2. grouped ncclBroadcast, one per rank (the pattern of comm_gather_rows: femshell_get_solution, the all-gather in front of the
//     replicated levels of the multigrid)
// scratch: device memory for 16 + 2 x world doubles.  us_out[0..2]: wall microseconds of the two patterns and of a lone
// all-reduce of three words (enqueue to completion, host clock).
bool comm_selftest(Comm &c, hipStream_t main, hipStream_t second, double *scratch, double us_out[3], std::string *err)
{
    ncclComm_t comm = static_cast<ncclComm_t>(c.comm);
    const int W = c.world, me = c.rank;
    auto fail = [&](const std::string &what) {
        if (err) *err = what;
        return false;
    };
    auto now_us = [] { return 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<double> h((size_t)16 + 2 * (size_t)W, 0.0);
    // layout: [0..5] ring send, [6..11] ring recv, [12..14] all-reduce words, [16 + r] broadcast slots, [16 + W + r] their source
    for (int i = 0; i < 6; i++) h[(size_t)i] = 100.0 * me + i;
    for (int i = 0; i < 3; i++) h[(size_t)12 + i] = (double)(me + 1) * (i + 1);
    h[(size_t)16 + W + me] = 7.0 + me;
    if (hipMemcpy(scratch, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail("self-test: upload failed");
    const int next = (me + 1) % W, prev = (me + W - 1) % W;
    double t0 = now_us();
    {
        CommWatch watch(me, W, "self-test: grouped send/recv on the halo stream beside an all-reduce on the main stream");
        if (W > 1) {
            if (!check(g_api.GroupStart(), "ncclGroupStart", err)) return false;
            bool ok = check(g_api.Send(scratch, 6, ncclDouble, next, comm, second), "ncclSend", err);
            ok = ok && check(g_api.Recv(scratch + 6, 6, ncclDouble, prev, comm, second), "ncclRecv", err);
            const bool ended = check(g_api.GroupEnd(), "ncclGroupEnd", ok ? err : nullptr);
            if (!ok || !ended) return false;
        }
        if (!check(g_api.AllReduce(scratch + 12, scratch + 12, 3, ncclDouble, ncclSum, comm, main), "ncclAllReduce", err)) return false;
        if (hipStreamSynchronize(second) != hipSuccess || hipStreamSynchronize(main) != hipSuccess) return fail("self-test: stream synchronisation failed");
    }
    us_out[0] = now_us() - t0;
    t0 = now_us();
    {
        CommWatch watch(me, W, "self-test: grouped ncclBroadcast, one per rank (row gather)");
        if (!check(g_api.GroupStart(), "ncclGroupStart", err)) return false;
        bool ok = true;
        for (int r = 0; r < W && ok; r++)
            ok = check(g_api.Broadcast(r == me ? scratch + 16 + W + r : scratch + 16 + r, scratch + 16 + r, 1, ncclDouble, r, comm, main), "ncclBroadcast", err);
        const bool ended = check(g_api.GroupEnd(), "ncclGroupEnd", ok ? err : nullptr);
        if (!ok || !ended) return false;
        if (hipStreamSynchronize(main) != hipSuccess) return fail("self-test: stream synchronisation failed");
    }
    us_out[1] = now_us() - t0;
    t0 = now_us();
    {
        CommWatch watch(me, W, "self-test: all-reduce of three words");
        if (!check(g_api.AllReduce(scratch + 12, scratch + 12, 3, ncclDouble, ncclSum, comm, main), "ncclAllReduce", err)) return false;
        if (hipStreamSynchronize(main) != hipSuccess) return fail("self-test: stream synchronisation failed");
    }
    us_out[2] = now_us() - t0;
    if (hipMemcpy(h.data(), scratch, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail("self-test: download failed");
    const double tri = 0.5 * W * (W + 1.0);
    if (W > 1)
        for (int i = 0; i < 6; i++)
            if (h[(size_t)6 + i] != 100.0 * prev + i)
                return fail("self-test: the grouped send/recv on the second stream delivered " + std::to_string(h[(size_t)6 + i]) + " instead of " +
                            std::to_string(100.0 * prev + i) + " from rank " + std::to_string(prev));
    for (int i = 0; i < 3; i++)
        if (h[(size_t)12 + i] != tri * (i + 1) * W) // (reduced twice: the second all-reduce sums the W copies of the first sum)
            return fail("self-test: the all-reduces returned " + std::to_string(h[(size_t)12 + i]) + " instead of " + std::to_string(tri * (i + 1) * W));
    for (int r = 0; r < W; r++)
        if (h[(size_t)16 + r] != 7.0 + r)
            return fail("self-test: the grouped broadcast of rank " + std::to_string(r) + " delivered " + std::to_string(h[(size_t)16 + r]));
    return true;
}

int comm_count(const Comm &c)
{
    int n = 0;
    if (!c.comm || !g_api.CommCount || g_api.CommCount(static_cast<ncclComm_t>(c.comm), &n) != ncclSuccess) return 0;
    return n;
}

void comm_destroy(Comm &c)
{
    if (c.comm && g_api.CommDestroy) g_api.CommDestroy(static_cast<ncclComm_t>(c.comm));
    c.comm = nullptr;
}

bool comm_allreduce_sum(Comm &c, double *buf, int count, hipStream_t st, std::string *err)
{
    c.allreduces++;
    c.collective_bytes += 8ll * count;
    return check(g_api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c.comm), st),
                 "ncclAllReduce", err);
}

bool comm_halo(Comm &c, const std::vector<HaloPeer> &peers, const std::vector<int32_t> &send_offsets,
               const double *sendbuf, double *p_ghost, hipStream_t st, std::string *err, int width)
{
    const int64_t w = width;
    ncclComm_t comm = static_cast<ncclComm_t>(c.comm);
    (st == c.second && st != nullptr ? c.halo_groups_second : c.halo_groups_main)++;
    if (!check(g_api.GroupStart(), "ncclGroupStart", err)) return false;
    bool ok = true;
    for (size_t i = 0; i < peers.size() && ok; i++) {
        const HaloPeer &h = peers[i];
        c.halo_bytes_sent += 8ll * w * (int64_t)h.send_nodes.size();
        if (!h.send_nodes.empty())
            ok = check(g_api.Send(sendbuf + w * send_offsets[i], (size_t)w * h.send_nodes.size(), ncclDouble, h.rank, comm, st),
                       "ncclSend", err);
        if (ok && h.recv_count > 0)
            ok = check(g_api.Recv(p_ghost + w * h.recv_offset, (size_t)w * (size_t)h.recv_count, ncclDouble, h.rank, comm, st),
                       "ncclRecv", err);
    }
    const bool ended = check(g_api.GroupEnd(), "ncclGroupEnd", ok ? err : nullptr);
    return ok && ended;
}

bool comm_gather_rows(Comm &c, const double *x_owned, double *full, const std::vector<int32_t> &row_begin,
                      const std::vector<int32_t> &row_end, hipStream_t st, std::string *err)
{
    ncclComm_t comm = static_cast<ncclComm_t>(c.comm);
    c.gathers++;
    if (!check(g_api.GroupStart(), "ncclGroupStart", err)) return false;
    bool ok = true;
    for (int r = 0; r < c.world && ok; r++) {
        const size_t cnt = 6 * (size_t)(row_end[r] - row_begin[r]);
        if (cnt == 0) continue;
        if (r == c.rank) c.collective_bytes += 8ll * (int64_t)cnt;
        double *dst = full + 6ll * row_begin[r];
        ok = check(g_api.Broadcast(r == c.rank ? x_owned : dst, dst, cnt, ncclDouble, r, comm, st), "ncclBroadcast", err);
    }
    const bool ended = check(g_api.GroupEnd(), "ncclGroupEnd", ok ? err : nullptr);
    return ok && ended;
}

bool comm_gather_pieces(Comm &c, const double *mine, double *full, const std::vector<int64_t> &begin,
                        const std::vector<int64_t> &end, hipStream_t st, std::string *err)
{
    ncclComm_t comm = static_cast<ncclComm_t>(c.comm);
    c.gathers++;
    if (!check(g_api.GroupStart(), "ncclGroupStart", err)) return false;
    bool ok = true;
    for (int r = 0; r < c.world && ok; r++) {
        const size_t cnt = (size_t)(end[r] - begin[r]);
        if (cnt == 0) continue;
        if (r == c.rank) c.collective_bytes += 8ll * (int64_t)cnt;
        double *dst = full + begin[r];
        ok = check(g_api.Broadcast(r == c.rank ? mine : dst, dst, cnt, ncclDouble, r, comm, st), "ncclBroadcast", err);
    }
    const bool ended = check(g_api.GroupEnd(), "ncclGroupEnd", ok ? err : nullptr);
    return ok && ended;
}

} // namespace femshell
