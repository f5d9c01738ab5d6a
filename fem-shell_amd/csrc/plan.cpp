// plan.cpp -- host-side symbolic phase (see plan.hpp).
#include "plan.hpp"

#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <pthread.h>
#include <thread>

namespace femshell {

void (*plan_progress_hook)() = nullptr;


namespace {
constexpr std::size_t kHugeAlign = (std::size_t)2 << 20, kHugeFrom = (std::size_t)4 << 20;
}
void *raw_allocate(std::size_t bytes)
{
    if (bytes < kHugeFrom) return ::operator new(bytes);
    const std::size_t rounded = (bytes + kHugeAlign - 1) / kHugeAlign * kHugeAlign;
    void *p = std::aligned_alloc(kHugeAlign, rounded);
    if (p == nullptr) throw std::bad_alloc();
    // FEMSHELL_HUGEPAGES=0: no advice (a host whose memory is so fragmented that the kernel compacts it inside the page faults)
    // (read per allocation -- these are at least 4 MiB each --: the tests switch it inside one process)
    const char *env = getenv("FEMSHELL_HUGEPAGES");
    const bool advise = !(env && atoi(env) == 0);
    if (advise) (void)madvise(p, rounded, MADV_HUGEPAGE); // (advice: refused or unavailable, the array is an ordinary one)
    return p;
}
void raw_deallocate(void *p, std::size_t bytes) noexcept
{
    if (bytes < kHugeFrom) ::operator delete(p);
    else std::free(p);
}

int available_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int m = CPU_COUNT(&set);
        if (m > 0 && (n < 1 || m < n)) n = m;
    }
    // a container's CPU quota (cgroup v2 cpu.max "quota period", v1 cfs_quota_us / cfs_period_us): a box that shows 256
    // cores to sched_getaffinity under a quota of 16 runs 64 threads for a quarter of every period and stalls them for the
    // rest of it
    auto quota_cpus = []() -> int {
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (fq && fp && (fscanf(fq, "%lld", &quota) != 1 || fscanf(fp, "%lld", &period) != 1)) quota = -1;
            if (fq) fclose(fq);
            if (fp) fclose(fp);
        }
        if (quota <= 0 || period <= 0) return 0;
        return (int)((quota + period - 1) / period);
    };
    static const int quota = quota_cpus();
    if (quota > 0 && quota < n) n = quota;
    return n < 1 ? 1 : n;
}

// Ranks that share this host's cores (set by femshell_create of a multi-rank context: one process per GPU, all on one
// node): every rank takes its share of the cores for the symbolic phase and the multigrid setup instead of all of them.
static std::atomic<int> g_host_share{1};
void set_host_share(int ranks_on_this_host) { g_host_share.store(ranks_on_this_host < 1 ? 1 : ranks_on_this_host); }

// host threads of the symbolic phase and the multigrid setup: FEMSHELL_HOST_THREADS, else the available cores divided by
// the ranks on the host; at most 64
int host_thread_count()
{
    const char *e = getenv("FEMSHELL_HOST_THREADS"); // (read per call: tests switch it inside one process)
    const int from_env = e ? atoi(e) : 0;
    int t = from_env > 0 ? from_env : available_cpus() / g_host_share.load();
    return t < 1 ? 1 : (t > 64 ? 64 : t);
}
// ---- the pool behind run_on_host_threads
namespace {
struct HostPool {
    std::mutex use; // one job at a time
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    const std::function<void(int)> *job = nullptr;
    uint64_t generation = 0;
    int nt = 0;      // tasks of the current job (task 0 is the caller's; worker w takes task w + 1)
    int pending = 0; // tasks of the workers that have not finished
};
thread_local bool t_pool_worker = false; // this thread is inside a task of a pool job
std::atomic<HostPool *> g_host_pool{nullptr};

void pool_worker(HostPool *P, int w)
{
    t_pool_worker = true;
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(P->m);
    for (;;) {
        P->wake.wait(lk, [&] { return P->generation != seen; });
        seen = P->generation;
        if (w + 1 >= P->nt) continue; // (a job of fewer tasks)
        const std::function<void(int)> *job = P->job;
        lk.unlock();
        (*job)(w + 1);
        lk.lock();
        if (--P->pending == 0) P->done.notify_one();
    }
}
void spawn_and_join(int nt, const std::function<void(int)> &task)
{
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 1; t < nt; t++) th.emplace_back([&task, t] { task(t); });
    task(0);
    for (auto &t : th) t.join();
}
HostPool *host_pool()
{
    HostPool *P = g_host_pool.load(std::memory_order_acquire);
    if (P) return P;
    static std::mutex create;
    std::lock_guard<std::mutex> g(create);
    P = g_host_pool.load(std::memory_order_acquire);
    if (!P) {
        P = new HostPool(); // never freed: its workers wait in it until the process ends
        // a forked child has the parent's memory and none of its threads: it starts a pool of its own when it needs one
        static std::once_flag once;
        std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { g_host_pool.store(nullptr, std::memory_order_release); }); });
        g_host_pool.store(P, std::memory_order_release);
    }
    return P;
}
} // namespace

void run_on_host_threads(int nt, const std::function<void(int)> &task)
{
    if (nt <= 1) {
        task(0);
        return;
    }
    static const bool pool_off = getenv("FEMSHELL_HOST_POOL") && atoi(getenv("FEMSHELL_HOST_POOL")) == 0;
    if (t_pool_worker) { // a call from inside a task: the cores are taken, the tasks run one after the other
        for (int t = 0; t < nt; t++) task(t);
        return;
    }
    HostPool *P = pool_off ? nullptr : host_pool();
    if (P == nullptr || !P->use.try_lock()) { // the pool is at work for another host thread (another context's)
        spawn_and_join(nt, task);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(P->m);
        while ((int)P->workers.size() < nt - 1) {
            const int w = (int)P->workers.size();
            P->workers.emplace_back(pool_worker, P, w);
            P->workers.back().detach();
        }
        P->job = &task;
        P->nt = nt;
        P->pending = nt - 1;
        P->generation++;
    }
    P->wake.notify_all();
    t_pool_worker = true; // (the caller's task is a task like the others: calls from inside it run serially)
    task(0);
    t_pool_worker = false;
    {
        std::unique_lock<std::mutex> lk(P->m);
        P->done.wait(lk, [&] { return P->pending == 0; });
        P->job = nullptr;
        P->nt = 0;
    }
    P->use.unlock();
}

static int plan_threads() { return host_thread_count(); }
template <class F> static void plan_parallel(int64_t n, int64_t min_chunk, F f) // f(thread, begin, end)
{
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(plan_threads(), n / std::max<int64_t>(min_chunk, 1)));
    if (nt <= 1) {
        f(0, (int64_t)0, n);
        return;
    }
    run_on_host_threads(nt, [&f, n, nt](int t) { f(t, n * t / nt, n * (t + 1) / nt); });
}
template <class V, class T> static void plan_fill(V &v, T value) // v[i] = value on the host threads (first touch included)
{
    plan_parallel((int64_t)v.size(), 1 << 20, [&](int, int64_t b, int64_t e) { std::fill(v.begin() + b, v.begin() + e, value); });
}
static int plan_chunks(int64_t n, int64_t min_chunk)
{
    return (int)std::max<int64_t>(1, std::min<int64_t>(plan_threads(), n / std::max<int64_t>(min_chunk, 1)));
}

void partition_rows(int32_t n_nodes, int world, int rank, int32_t *begin, int32_t *end)
{
    const int64_t slices = ((int64_t)n_nodes + kSliceNodes - 1) / kSliceNodes;
    const int64_t s0 = slices * rank / world, s1 = slices * (rank + 1) / world;
    *begin = (int32_t)std::min<int64_t>(s0 * kSliceNodes, n_nodes);
    *end = (int32_t)std::min<int64_t>(s1 * kSliceNodes, n_nodes);
}

void partition_bounds(int32_t n_nodes, int32_t n_tri, const int32_t *tri, int32_t n_quad, const int32_t *quad, int world,
                      std::vector<int32_t> *bounds)
{
    bounds->assign((size_t)world + 1, n_nodes);
    (*bounds)[0] = 0;
    if (world <= 1) return;
    const int64_t slices = ((int64_t)n_nodes + kSliceNodes - 1) / kSliceNodes;
    std::vector<int64_t> wsum((size_t)slices + 1, 0); // weight of slice s at wsum[s + 1], prefix sums below
    auto count = [&](const int32_t *conn, int64_t n) {
        for (int64_t q = 0; q < n; q++)
            if (conn[q] >= 0 && conn[q] < n_nodes) wsum[(size_t)(conn[q] / kSliceNodes) + 1]++;
    };
    count(tri, 3ll * n_tri);
    count(quad, 4ll * n_quad);
    for (int64_t s = 0; s < slices; s++) {
        const int64_t nodes_in = std::min<int64_t>(kSliceNodes, (int64_t)n_nodes - s * kSliceNodes);
        wsum[(size_t)s + 1] += nodes_in + wsum[(size_t)s];
    }
    const int64_t total = wsum[(size_t)slices];
    int64_t s = 0;
    for (int r = 1; r < world; r++) {
        const int64_t want = total * r / world;
        while (s < slices && wsum[(size_t)s] < want) s++;
        // never an empty rank while slices remain for the ranks behind
        s = std::max<int64_t>(s, std::min<int64_t>(r, slices));
        s = std::min<int64_t>(s, std::max<int64_t>(slices - (world - r), (*bounds)[(size_t)r - 1] / kSliceNodes));
        (*bounds)[(size_t)r] = (int32_t)std::min<int64_t>(s * kSliceNodes, n_nodes);
    }
}

static int owner_of(int32_t node, const std::vector<int32_t> &bounds)
{
    // the rank r with bounds[r] <= node < bounds[r + 1] (empty ranges skipped by upper_bound)
    return (int)(std::upper_bound(bounds.begin(), bounds.end(), node) - bounds.begin()) - 1;
}

bool default_symmetric_storage()
{
    const char *e = getenv("FEMSHELL_SYMMETRIC");
    return !(e && atoi(e) == 0);
}

// Items of one slice (slot by slot, chunk 0 first: Plan::Item) into the lanes of the pipelined kernel's rounds: 192 lanes
// = three waves per round; all chunks of a slot in consecutive lanes of one wave (their partial sums meet through lane
// shifts, no LDS, no barrier).  The first wave holds diagonal slots only -- its lanes run the upper-triangle routine, three
// contributions each; diagonal slots it has no room for (a Delaunay slice has 74 chunks of three) join the other slots in
// the following waves, cut into chunks of two and marked (bit 31 of z): there every lane runs the general block routine
// on at most two or three contributions, and no wave runs both routines.  Unused lanes carry inert items.  Word w of every
// item of a wave: most chunks of a slot in this wave | all-upper-triangle flag << 8.
static void pack_items_pipe(const std::vector<Plan::Item> &in, std::vector<Plan::Item> *out)
{
    constexpr int kRound = 192, kWave = 64;
    const Plan::Item pad{0xffffu, 0, 0, 0};
    out->clear();
    size_t round_begin = 0;
    int pos = 0; // lane within the current round
    auto place = [&](const Plan::Item *first, int n) {
        if (pos % kWave + n > kWave) pos = (pos / kWave + 1) * kWave; // the slot's chunks stay in one wave
        if (pos + n > kRound) {
            out->resize(round_begin + kRound, pad);
            round_begin += kRound;
            pos = 0;
        }
        out->resize(round_begin + pos, pad);
        for (int c = 0; c < n; c++) out->push_back(first[c]);
        pos += n;
    };
    auto is_diag = [&](const Plan::Item &it) { return (it.x & 0xffffu) < (uint32_t)kSliceNodes; };
    // diagonal slots into the first wave while they fit; the others are cut into chunks of two for the general routine
    std::vector<Plan::Item> general;
    int off_lanes = 0;
    for (size_t i = 0; i < in.size(); i += in[i].x >> 24) {
        const int n = (int)(in[i].x >> 24);
        if (!is_diag(in[i])) {
            off_lanes += n;
            continue;
        }
        if (round_begin == 0 && pos + n <= kWave) {
            place(&in[i], n);
            continue;
        }
        uint32_t pr[3 * 64];
        int np = 0;
        for (int c = 0; c < n; c++) {
            const Plan::Item &it = in[i + c];
            const int k = (int)((it.z >> 16) & 0xffu);
            const uint32_t q[3] = {it.y & 0xffffu, it.y >> 16, it.z & 0xffffu};
            for (int j = 0; j < k; j++) pr[np++] = q[j];
        }
        const int nch = (np + 1) / 2;
        for (int c = 0; c < nch; c++) {
            const int k = std::min(2, np - 2 * c);
            Plan::Item g;
            g.x = (in[i].x & 0xffffu) | ((uint32_t)c << 16) | ((uint32_t)nch << 24);
            g.y = pr[2 * c] | ((k > 1 ? pr[2 * c + 1] : 0u) << 16);
            g.z = ((uint32_t)k << 16) | 0x80000000u;
            g.w = 0;
            general.push_back(g);
        }
    }
    // the other slots start a wave of their own when the round has the room
    const int general_lanes = (int)general.size();
    if (pos % kWave != 0 && (pos / kWave + 1) * kWave + general_lanes + off_lanes <= kRound) pos = (pos / kWave + 1) * kWave;
    for (size_t i = 0; i < general.size(); i += general[i].x >> 24) place(&general[i], (int)(general[i].x >> 24));
    for (int np = kItemPairs; np >= 0; np--)
        for (size_t i = 0; i < in.size(); i += in[i].x >> 24)
            if (!is_diag(in[i]) && (int)((in[i].z >> 16) & 0xffu) == np) place(&in[i], (int)(in[i].x >> 24));
    // A last wave that is at most half full of whole off-diagonal slots (32 of a structured slice's 96) is cut finer: every
    // slot of two or three contributions becomes two chunks in neighbouring lanes, so that the wave runs one contribution
    // (two for a slot of three) plus a lane sum instead of two or three -- it shares its SIMD with the other workgroup's
    // producer wave, the longest one (assemble_kernel.hpp)
    {
        const size_t w0 = round_begin + (size_t)(pos > 0 ? (pos - 1) / kWave : 0) * kWave;
        const size_t w1 = out->size();
        bool fits = pos > 0 && w1 > w0;
        int lanes_after = 0;
        for (size_t i = w0; i < w1 && fits; i++) {
            const Plan::Item &it = (*out)[i];
            const uint32_t nch = it.x >> 24, np = (it.z >> 16) & 0xffu;
            if (nch == 0) continue;
            if (nch != 1 || (it.x & 0xffffu) < (uint32_t)kSliceNodes) fits = false;
            lanes_after += np >= 2 ? 2 : 1;
        }
        if (fits && lanes_after <= kWave && lanes_after > (int)(w1 - w0) - 0) {
            std::vector<Plan::Item> cut;
            for (size_t i = w0; i < w1; i++) {
                const Plan::Item &it = (*out)[i];
                const uint32_t nch = it.x >> 24, np = (it.z >> 16) & 0xffu;
                if (nch == 0) continue; // (padding inside the wave is dropped)
                if (np < 2) {
                    cut.push_back(it);
                    continue;
                }
                const uint32_t pr[3] = {it.y & 0xffffu, it.y >> 16, it.z & 0xffffu};
                const uint32_t n0 = np - 1; // first chunk: all but the last contribution
                Plan::Item a = it, b = it;
                a.x = (it.x & 0xffffu) | (0u << 16) | (2u << 24);
                a.y = pr[0] | ((n0 > 1 ? pr[1] : 0u) << 16);
                a.z = 0u | (n0 << 16);
                b.x = (it.x & 0xffffu) | (1u << 16) | (2u << 24);
                b.y = pr[np - 1];
                b.z = 0u | (1u << 16);
                cut.push_back(a);
                cut.push_back(b);
            }
            out->resize(w0);
            out->insert(out->end(), cut.begin(), cut.end());
        }
    }
    // the waves' words
    for (size_t w0 = 0; w0 < out->size(); w0 += kWave) {
        const size_t w1 = std::min(out->size(), w0 + kWave);
        uint32_t most = 0, all_diag = 1;
        for (size_t i = w0; i < w1; i++) {
            const Plan::Item &it = (*out)[i];
            if ((it.x >> 24) == 0) continue; // inert
            most = std::max(most, it.x >> 24);
            if ((it.x & 0xffffu) >= (uint32_t)kSliceNodes || (it.z >> 31)) all_diag = 0;
        }
        for (size_t i = w0; i < w1; i++) (*out)[i].w = most | (all_diag << 8);
    }
}

bool build_plan(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                const int32_t *quad, int rank, int world, Plan *P, std::string *err, bool symmetric, bool geometric_orientation)
{
    auto fail = [&](const std::string &m) {
        if (err) *err = m;
        return false;
    };
    if (n_nodes <= 0) return fail("mesh has no nodes");
    if (n_tri < 0 || n_quad < 0 || (int64_t)n_tri + n_quad <= 0) return fail("mesh has no elements");
    if (world < 1 || rank < 0 || rank >= world) return fail("invalid rank/world_size");
    if ((int64_t)n_tri + n_quad >= (1ll << 28)) return fail("more than 2^28 elements");
    {
        // the first element (lowest index) of either kind that references a node out of range, and the first that repeats a
        // node: found on the host threads, reported in the order the serial checks would find them
        auto first_bad = [&](const int32_t *conn, int nn, int32_t ne, int64_t *out_of_range, int64_t *repeats) {
            std::atomic<int64_t> oor{INT64_MAX}, rep{INT64_MAX};
            auto lower = [](std::atomic<int64_t> &a, int64_t v) {
                int64_t cur = a.load();
                while (v < cur && !a.compare_exchange_weak(cur, v)) {}
            };
            plan_parallel(ne, 1 << 16, [&](int, int64_t e0, int64_t e1) {
                int64_t o = INT64_MAX, r = INT64_MAX;
                for (int64_t e = e0; e < e1 && (o == INT64_MAX || r == INT64_MAX); e++) {
                    const int32_t *c = conn + nn * e;
                    bool bad = false, twice = false;
                    for (int i = 0; i < nn; i++) {
                        bad |= c[i] < 0 || c[i] >= n_nodes;
                        for (int j = i + 1; j < nn; j++) twice |= c[i] == c[j];
                    }
                    if (bad && o == INT64_MAX) o = e;
                    if (twice && r == INT64_MAX) r = e;
                }
                lower(oor, o);
                lower(rep, r);
            });
            *out_of_range = oor.load();
            *repeats = rep.load();
        };
        int64_t t_oor, t_rep, q_oor, q_rep;
        first_bad(tri, 3, n_tri, &t_oor, &t_rep);
        first_bad(quad, 4, n_quad, &q_oor, &q_rep);
        if (t_oor != INT64_MAX) return fail("triangle " + std::to_string(t_oor) + " references a node out of range");
        if (q_oor != INT64_MAX) return fail("quad " + std::to_string(q_oor) + " references a node out of range");
        if (t_rep != INT64_MAX) return fail("triangle " + std::to_string(t_rep) + " repeats a node");
        if (q_rep != INT64_MAX) return fail("quad " + std::to_string(q_rep) + " repeats a node");
    }

    // FEMSHELL_PLAN_VERBOSE=1: wall time of the phases below on stderr.  lap(name) opens the phase `name` and prints the
    // time of the one it closes.
    static const bool verbose = getenv("FEMSHELL_PLAN_VERBOSE") && atoi(getenv("FEMSHELL_PLAN_VERBOSE")) != 0;
    auto t_lap = std::chrono::steady_clock::now();
    const char *open_phase = "input checks";
    auto lap = [&](const char *what) {
        if (plan_progress_hook != nullptr) plan_progress_hook(); // (the watchdog of a multi-rank context counts from here again)
        if (!verbose) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[femshell plan] %-44s %.3f s\n", open_phase, std::chrono::duration<double>(t - t_lap).count());
        t_lap = t;
        open_phase = what;
    };
    Plan &p = *P;
    p = Plan();
    p.n_nodes = n_nodes;
    p.n_tri = n_tri;
    p.n_quad = n_quad;
    p.rank = rank;
    p.world = world;
    p.symmetric = symmetric;
    partition_bounds(n_nodes, n_tri, tri, n_quad, quad, world, &p.part_bounds);
    p.row_begin = p.part_bounds[(size_t)rank];
    p.row_end = p.part_bounds[(size_t)rank + 1];
    const int32_t g0 = p.row_begin, g1 = p.row_end;
    p.n_own = g1 - g0;
    p.n_pad = (p.n_own + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    p.n_slices = p.n_pad / kSliceNodes;

    lap("node - element adjacency");
    // ---- node -> element adjacency for the owned nodes; entries (element << 2 | index in element),
    //      element = combined id (triangles first), ascending per node
    const int32_t n_own = p.n_own;
    std::vector<int64_t> adj_ptr((size_t)n_own + 1, 0);
    // Ranges of owned nodes on the host threads: every thread walks the whole connectivity and keeps what falls into its
    // range, once to count and once to fill -- the entries of a node in ascending element order, as a serial counting
    // sort leaves them.
    const int n_ranges = plan_chunks(n_own, 32768);
    auto range_of = [&](int64_t t, int32_t *a0, int32_t *a1) {
        *a0 = (int32_t)((int64_t)n_own * t / n_ranges);
        *a1 = (int32_t)((int64_t)n_own * (t + 1) / n_ranges);
    };
    plan_parallel(n_ranges, 1, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++) {
            int32_t a0, a1;
            range_of(t, &a0, &a1);
            const uint32_t span = (uint32_t)(a1 - a0);
            const int32_t first = g0 + a0;
            auto count = [&](const int32_t *conn, int64_t n) {
                for (int64_t q = 0; q < n; q++) {
                    const uint32_t d = (uint32_t)(conn[q] - first);
                    if (d < span) adj_ptr[(size_t)a0 + d + 1]++;
                }
            };
            count(tri, 3ll * n_tri);
            count(quad, 4ll * n_quad);
        }
    });
    for (int32_t a = 0; a < n_own; a++) {
        if (adj_ptr[a + 1] == 0) return fail("node " + std::to_string(g0 + a) + " is not attached to any element");
        adj_ptr[a + 1] += adj_ptr[a];
    }
    RawVec<uint32_t> adj((size_t)adj_ptr[n_own]);
    plan_parallel(n_ranges, 1, [&](int, int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++) {
            int32_t a0, a1;
            range_of(t, &a0, &a1);
            const uint32_t span = (uint32_t)(a1 - a0);
            const int32_t first = g0 + a0;
            std::vector<int64_t> fill(adj_ptr.begin() + a0, adj_ptr.begin() + a1);
            for (int32_t e = 0; e < n_tri; e++)
                for (int i = 0; i < 3; i++) {
                    const uint32_t d = (uint32_t)(tri[3ll * e + i] - first);
                    if (d < span) adj[(size_t)fill[d]++] = ((uint32_t)e << 2) | (uint32_t)i;
                }
            for (int32_t e = 0; e < n_quad; e++)
                for (int i = 0; i < 4; i++) {
                    const uint32_t d = (uint32_t)(quad[4ll * e + i] - first);
                    if (d < span) adj[(size_t)fill[d]++] = ((uint32_t)(n_tri + e) << 2) | (uint32_t)i;
                }
        }
    });

    lap("local elements");
    // ---- local elements: every element touching an owned node, numbered in input order (triangles, then quadrilaterals)
    RawVec<int32_t> elem_local((size_t)n_tri + n_quad);
    auto number_local = [&](const int32_t *conn, int nn, int32_t ne, int32_t id_offset, int32_t first_local, std::vector<int32_t> *global_ids) {
        const int nch = plan_chunks(ne, 1 << 16);
        std::vector<int32_t> chunk_count((size_t)nch + 1, 0);
        plan_parallel(nch, 1, [&](int, int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; c++) {
                int32_t k = 0;
                for (int64_t e = (int64_t)ne * c / nch; e < (int64_t)ne * (c + 1) / nch; e++) {
                    bool touches = false;
                    for (int i = 0; i < nn; i++) touches |= (uint32_t)(conn[nn * e + i] - g0) < (uint32_t)n_own;
                    elem_local[(size_t)(id_offset + e)] = touches ? 0 : -1;
                    k += touches ? 1 : 0;
                }
                chunk_count[(size_t)c + 1] = k;
            }
        });
        for (int c = 0; c < nch; c++) chunk_count[(size_t)c + 1] += chunk_count[(size_t)c];
        global_ids->resize((size_t)chunk_count[(size_t)nch]);
        plan_parallel(nch, 1, [&](int, int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; c++) {
                int32_t at = chunk_count[(size_t)c];
                for (int64_t e = (int64_t)ne * c / nch; e < (int64_t)ne * (c + 1) / nch; e++)
                    if (elem_local[(size_t)(id_offset + e)] == 0) {
                        elem_local[(size_t)(id_offset + e)] = first_local + at;
                        (*global_ids)[(size_t)at++] = (int32_t)e;
                    }
            }
        });
    };
    number_local(tri, 3, n_tri, 0, 0, &p.tri_global_id);
    const int32_t n_ltri = (int32_t)p.tri_global_id.size();
    number_local(quad, 4, n_quad, n_tri, n_ltri, &p.quad_global_id);

    lap("slot structures");
    // ---- per owned node: block slots (slot 0 = diagonal, then ascending global column) and
    //      the gather list of every slot
    struct Slot {
        int32_t col;   // global node id
        int32_t first; // head of a singly linked list into tmp_pairs (ascending element order)
        int32_t last;
        int32_t count;
    };
    std::vector<int32_t> node_slot_ptr((size_t)n_own + 1, 0);
    RawVec<int32_t> slot_col;                         // global ids, per node contiguous
    RawVec<int32_t> slot_pair_ptr(1, 0);              // per slot
    RawVec<uint32_t> slot_pairs;
    lap("symmetric storage: neighbour lists");
    // ---- symmetric storage: which row of an owned pair (a,c) holds the block.  Any choice works -- the SpMV applies
    // every stored off-diagonal block to both rows -- so it is made to balance the rows: the ELL width of a slice is the
    // largest slot count of its 32 rows.  Start: the lower-numbered row keeps the block (on a structured grid every
    // interior row then holds exactly its three higher neighbours and nothing below moves); then local repair: a row
    // above the mean hands a block to a neighbour that is at least two below it, directly or through one
    // intermediate row, until nothing moves.  Planar triangulations admit 3 per inner node (Schnyder); the repair gets unstructured Delaunay
    // meshes to width 4-5 where "lower row keeps" gives 6-7 and full storage 10-11.
    std::vector<int64_t> nb_ptr;
    RawVec<int32_t> nb;      // owned neighbours (local ids), ascending per row
    RawVec<uint8_t> nb_mine; // 1: the block (row, nb) is stored with this row
    int64_t lower_blocks = 0;
    if (symmetric) {
        nb_ptr.assign((size_t)n_own + 1, 0);
        std::vector<int32_t> cnt_ghost((size_t)n_own, 0);
        // one parallel pass: the sorted distinct neighbours of every row land in a scratch array at an upper-bound offset
        // (three or four entries per adjacent element), counts are taken, a second parallel pass compacts the owned ones
        std::vector<int64_t> ub((size_t)n_own + 1, 0);
        for (int32_t a = 0; a < n_own; a++) {
            int64_t m = 0;
            for (int64_t q = adj_ptr[a]; q < adj_ptr[a + 1]; q++) m += ((adj[q] >> 2) < (uint32_t)n_tri) ? 2 : 3;
            ub[(size_t)a + 1] = ub[(size_t)a] + m;
        }
        RawVec<int32_t> scratch((size_t)ub[(size_t)n_own]);
        std::vector<int32_t> n_distinct((size_t)n_own, 0);
        plan_parallel(n_own, 4096, [&](int, int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) {
                int32_t *t = scratch.data() + ub[(size_t)a];
                int m = 0;
                for (int64_t q = adj_ptr[a]; q < adj_ptr[a + 1]; q++) {
                    const uint32_t ge = adj[q] >> 2;
                    const bool is_tri = ge < (uint32_t)n_tri;
                    const int nn = is_tri ? 3 : 4;
                    const int32_t *c = is_tri ? tri + 3ll * ge : quad + 4ll * (ge - n_tri);
                    for (int ib = 0; ib < nn; ib++) {
                        const int32_t b = c[ib];
                        if (b == g0 + (int32_t)a) continue;
                        if (b >= g0 && b < g1) t[m++] = b - g0;
                        else t[m++] = n_own + (b < g0 ? b : b - n_own); // ghost marker (distinct per global id)
                    }
                }
                std::sort(t, t + m);
                m = (int)(std::unique(t, t + m) - t);
                int owned = 0;
                for (int k = 0; k < m; k++) owned += t[k] < n_own ? 1 : 0;
                n_distinct[(size_t)a] = owned; // (owned entries sort in front of the ghost markers)
                cnt_ghost[(size_t)a] = m - owned;
            }
        });
        for (int32_t a = 0; a < n_own; a++) nb_ptr[(size_t)a + 1] = nb_ptr[(size_t)a] + n_distinct[(size_t)a];
        nb.resize((size_t)nb_ptr[n_own]);
        plan_parallel(n_own, 4096, [&](int, int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++)
                std::copy_n(scratch.data() + ub[(size_t)a], n_distinct[(size_t)a], nb.data() + nb_ptr[(size_t)a]);
        });
        RawVec<int32_t>().swap(scratch);
        lap("symmetric storage: which row holds a block");
        std::vector<int32_t> cnt(cnt_ghost); // blocks a row stores besides its diagonal (ghost columns included)
        nb_mine.resize(nb.size()); // (every entry written by the loop below)
        auto index_of = [&](int32_t row, int32_t col) -> int64_t {
            return std::lower_bound(nb.begin() + nb_ptr[row], nb.begin() + nb_ptr[row + 1], col) - nb.begin();
        };
        // start: the lower-numbered row keeps the block -- or, for numberings that are not monotone along the mesh (Morton,
        // Cuthill-McKee, arbitrary input), the row from which the neighbour lies in the "positive half space" (first
        // non-zero coordinate difference positive): every interior node of a triangulation sees about half of its
        // neighbours there whatever the numbering is (exactly three on the structured grids)
        // (the caller asks for it when the library renumbered the nodes itself; FEMSHELL_SYM_ORIENT=geom|index overrides.
        //  4M-triangle panel, Morton numbering: SpMV 0.466 -> 0.456 ms, Hilbert 0.546 -> 0.453 ms; row-major: same plan)
        const char *oe = getenv("FEMSHELL_SYM_ORIENT");
        const bool geometric = oe ? std::strcmp(oe, "geom") == 0 : geometric_orientation;
        plan_parallel(n_own, 4096, [&](int, int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++)
                for (int64_t q = nb_ptr[a]; q < nb_ptr[a + 1]; q++) {
                    const int32_t c = nb[q];
                    bool mine = c > a;
                    if (geometric) {
                        const double *xa = xyz + 3ll * (g0 + a), *xc = xyz + 3ll * (g0 + c);
                        for (int d = 0; d < 3; d++) {
                            // (a non-finite coordinate compares false from both rows and the block would be lost: such a
                            //  pair keeps the index rule -- femshell_plan_create does not validate coordinates)
                            if (!std::isfinite(xa[d]) || !std::isfinite(xc[d])) break;
                            if (xc[d] != xa[d]) {
                                mine = xc[d] > xa[d];
                                break;
                            }
                        }
                    }
                    nb_mine[q] = mine ? 1 : 0;
                    if (mine) cnt[a]++;
                }
        });
        // local repair towards the mean
        const int target = (int)((nb.size() / 2 + (size_t)n_own - 1) / (size_t)std::max(n_own, 1));
        auto give = [&](int32_t from, int64_t q_from) { // the block (from, nb[q_from]) moves to the other row
            const int32_t to = nb[q_from];
            nb_mine[q_from] = 0;
            nb_mine[index_of(to, from)] = 1;
            cnt[from]--;
            cnt[to]++;
        };
        for (int sweep = 0; sweep < 32; sweep++) {
            int64_t moved = 0;
            for (int32_t a = 0; a < n_own; a++) {
                while (cnt[a] > target) {
                    bool done = false;
                    for (int64_t q = nb_ptr[a]; q < nb_ptr[a + 1] && !done; q++)
                        if (nb_mine[q] && cnt[nb[q]] + 2 <= cnt[a]) {
                            give(a, q);
                            done = true;
                        }
                    for (int64_t q = nb_ptr[a]; q < nb_ptr[a + 1] && !done; q++) {
                        if (!nb_mine[q]) continue;
                        const int32_t c = nb[q];
                        if (cnt[c] + 1 > cnt[a]) continue; // c would rise above a
                        for (int64_t r = nb_ptr[c]; r < nb_ptr[c + 1] && !done; r++)
                            if (nb_mine[r] && nb[r] != a && cnt[nb[r]] + 2 <= cnt[a]) {
                                give(c, r);
                                give(a, q);
                                done = true;
                            }
                    }
                    if (!done) break;
                    moved++;
                }
            }
            if (!moved) break;
        }
        for (size_t q = 0; q < nb_mine.size(); q++) lower_blocks += nb_mine[q] ? 0 : 1; // blocks of K without a slot in their row
        lap("slots and gather pairs per row");
    }
    auto stored_here = [&](int32_t a, int32_t b_global) { // symmetric storage: does row a hold the block (a, b)?
        if (b_global < g0 || b_global >= g1) return true; // ghost column
        const int32_t c = b_global - g0;
        const int64_t q = std::lower_bound(nb.begin() + nb_ptr[a], nb.begin() + nb_ptr[a + 1], c) - nb.begin();
        return nb_mine[q] != 0;
    };
    {
        // chunks of rows on the host threads, each into lists of its own (slot_pair_ptr relative to the chunk), joined in
        // row order afterwards
        const int nchunks_r = plan_chunks(n_own, 4096);
        std::vector<RawVec<int32_t>> part_col((size_t)nchunks_r), part_ptr((size_t)nchunks_r);
        std::vector<RawVec<uint32_t>> part_pairs((size_t)nchunks_r);
        plan_parallel(n_own, 4096, [&](int t, int64_t a0, int64_t a1) {
            std::vector<Slot> slots;
            std::vector<uint32_t> tmp_pairs; // packed pair
            std::vector<int32_t> tmp_next;
            std::vector<int> order;
            // (lists of the thread's own, handed over at the end: the headers of part_col[t] and part_col[t + 1] share a cache
            //  line, and every push_back through them made the threads take turns -- this loop did not scale at all)
            RawVec<int32_t> pc, pp;
            RawVec<uint32_t> pairs_t;
            pc.reserve((size_t)(a1 - a0) * 8);
            pp.reserve((size_t)(a1 - a0) * 8);
            pairs_t.reserve((size_t)(adj_ptr[a1] - adj_ptr[a0]) * 3);
            for (int64_t a = a0; a < a1; a++) {
                slots.clear();
                tmp_pairs.clear();
                tmp_next.clear();
                slots.push_back({g0 + (int32_t)a, -1, -1, 0});
                for (int64_t q = adj_ptr[a]; q < adj_ptr[a + 1]; q++) {
                    const uint32_t ge = adj[q] >> 2, ia = adj[q] & 3u;
                    const bool is_tri = ge < (uint32_t)n_tri;
                    const int nn = is_tri ? 3 : 4;
                    const int32_t *c = is_tri ? tri + 3ll * ge : quad + 4ll * (ge - n_tri);
                    const uint32_t le = (uint32_t)elem_local[ge];
                    for (int ib = 0; ib < nn; ib++) {
                        const int32_t b = c[ib];
                        if (symmetric && b != g0 + (int32_t)a && !stored_here((int32_t)a, b)) continue; // the block lives with row b
                                                                                                       // and acts here through its transpose
                        size_t s = 0;
                        for (; s < slots.size(); s++)
                            if (slots[s].col == b) break;
                        if (s == slots.size()) slots.push_back({b, -1, -1, 0});
                        const int32_t id = (int32_t)tmp_pairs.size();
                        tmp_pairs.push_back((le << 4) | (ia << 2) | (uint32_t)ib);
                        tmp_next.push_back(-1);
                        if (slots[s].first < 0) slots[s].first = id; else tmp_next[slots[s].last] = id;
                        slots[s].last = id;
                        slots[s].count++;
                    }
                }
                order.resize(slots.size());
                for (size_t s = 0; s < slots.size(); s++) order[s] = (int)s;
                std::sort(order.begin() + 1, order.end(), [&](int x, int y) { return slots[x].col < slots[y].col; });
                for (int s : order) {
                    pc.push_back(slots[s].col);
                    for (int32_t id = slots[s].first; id >= 0; id = tmp_next[id]) pairs_t.push_back(tmp_pairs[id]);
                    pp.push_back((int32_t)pairs_t.size()); // end of the slot's pairs, relative to the chunk
                }
                node_slot_ptr[(size_t)a + 1] = (int32_t)slots.size(); // count; prefix sums below
            }
            part_col[(size_t)t].swap(pc);
            part_ptr[(size_t)t].swap(pp);
            part_pairs[(size_t)t].swap(pairs_t);
        });
        size_t n_slots = 0, n_pairs = 0;
        for (int t = 0; t < nchunks_r; t++) {
            n_slots += part_col[(size_t)t].size();
            n_pairs += part_pairs[(size_t)t].size();
        }
        if (n_pairs > 0x7fffff00ull || n_slots > 0x7fffff00ull) return fail("gather list exceeds 2^31 entries");
        lap("join of the per-thread lists");
        for (int32_t a = 0; a < n_own; a++) node_slot_ptr[(size_t)a + 1] += node_slot_ptr[(size_t)a];
        slot_col.resize(n_slots);
        slot_pair_ptr.resize(n_slots + 1);
        slot_pair_ptr[0] = 0;
        slot_pairs.resize(n_pairs);
        std::vector<size_t> so_of((size_t)nchunks_r + 1, 0), po_of((size_t)nchunks_r + 1, 0);
        for (int t = 0; t < nchunks_r; t++) {
            so_of[(size_t)t + 1] = so_of[(size_t)t] + part_col[(size_t)t].size();
            po_of[(size_t)t + 1] = po_of[(size_t)t] + part_pairs[(size_t)t].size();
        }
        plan_parallel(nchunks_r, 1, [&](int, int64_t t0, int64_t t1) { // every chunk's lists to their place, chunks in parallel
            for (int64_t t = t0; t < t1; t++) {
                const size_t so = so_of[(size_t)t], po = po_of[(size_t)t];
                std::copy(part_col[(size_t)t].begin(), part_col[(size_t)t].end(), slot_col.begin() + so);
                for (size_t k = 0; k < part_ptr[(size_t)t].size(); k++) slot_pair_ptr[so + k + 1] = (int32_t)(po + (size_t)part_ptr[(size_t)t][k]);
                std::copy(part_pairs[(size_t)t].begin(), part_pairs[(size_t)t].end(), slot_pairs.begin() + po);
                RawVec<int32_t>().swap(part_col[(size_t)t]);
                RawVec<uint32_t>().swap(part_pairs[(size_t)t]);
            }
        });
    }
    p.stored_blocks = (int64_t)slot_col.size();
    p.nnz_blocks = p.stored_blocks + lower_blocks;

    lap("ghost columns");
    // ---- ghosts: referenced columns outside the owned range, ascending
    {
        std::vector<int32_t> g;
        for (int32_t c : slot_col)
            if (c < g0 || c >= g1) g.push_back(c);
        std::sort(g.begin(), g.end());
        g.erase(std::unique(g.begin(), g.end()), g.end());
        p.ghost_global.swap(g);
        p.n_ghost = (int32_t)p.ghost_global.size();
    }
    auto to_local = [&](int32_t gid) -> int32_t {
        if (gid >= g0 && gid < g1) return gid - g0;
        const auto it = std::lower_bound(p.ghost_global.begin(), p.ghost_global.end(), gid);
        return p.n_pad + (int32_t)(it - p.ghost_global.begin());
    };

    lap("local connectivity and coordinates");
    // ---- local copies of connectivity and coordinates
    p.tri_local.resize((size_t)n_ltri * 3);
    plan_parallel(n_ltri, 1 << 16, [&](int, int64_t e0, int64_t e1) {
        for (int64_t le = e0; le < e1; le++)
            for (int i = 0; i < 3; i++) p.tri_local[3 * le + i] = to_local(tri[3ll * p.tri_global_id[(size_t)le] + i]);
    });
    p.quad_local.resize(p.quad_global_id.size() * 4);
    plan_parallel((int64_t)p.quad_global_id.size(), 1 << 16, [&](int, int64_t e0, int64_t e1) {
        for (int64_t le = e0; le < e1; le++)
            for (int i = 0; i < 4; i++) p.quad_local[4 * le + i] = to_local(quad[4ll * p.quad_global_id[(size_t)le] + i]);
    });
    p.xyz_local.resize((size_t)p.n_local_nodes() * 3);
    plan_parallel(p.n_local_nodes(), 1 << 16, [&](int, int64_t a0, int64_t a1) {
        for (int64_t a = a0; a < a1; a++) {
            // owned rows; padding rows: harmless copies of a real point; ghosts
            const int64_t src = a < n_own ? g0 + a : (a < p.n_pad ? g0 : p.ghost_global[(size_t)(a - p.n_pad)]);
            for (int d = 0; d < 3; d++) p.xyz_local[3 * a + d] = xyz[3 * src + d];
        }
    });

    lap("pack into slices");
    // ---- pack into slices
    p.slice_width.assign(p.n_slices, 1);
    p.slice_base.assign((size_t)p.n_slices + 1, 0);
    for (int32_t s = 0; s < p.n_slices; s++) {
        int w = 1;
        for (int n = 0; n < kSliceNodes; n++) {
            const int32_t a = s * kSliceNodes + n;
            if (a < n_own) w = std::max(w, node_slot_ptr[a + 1] - node_slot_ptr[a]);
        }
        p.slice_width[s] = w;
        p.max_slice_width = std::max(p.max_slice_width, w);
        p.slice_base[s + 1] = p.slice_base[s] + (int64_t)w * kSliceNodes;
    }
    const int64_t total = p.slice_base[p.n_slices];
    if (total * 36 >= (1ll << 40)) return fail("matrix too large");
    if (symmetric && total >= (1ll << 31)) return fail("matrix too large for symmetric storage (slot indices are 32-bit)");
    p.cols.resize((size_t)total);
    p.pair_ptr.resize((size_t)total + 1);
    // slot order inside a slice is (k, n); the gather list follows the same order.  Slices on the host threads: the
    // gather entries of a slice first counted, then copied to the offset the counts of the slices before it give
    {
        std::vector<int64_t> slice_pairs((size_t)p.n_slices + 1, 0);
        auto real_slot = [&](int32_t a, int k) { return a < n_own && k < node_slot_ptr[a + 1] - node_slot_ptr[a]; };
        plan_parallel(p.n_slices, 256, [&](int, int64_t s0, int64_t s1) {
            for (int64_t s = s0; s < s1; s++) {
                int64_t m = 0;
                for (int n = 0; n < kSliceNodes; n++) {
                    const int32_t a = (int32_t)s * kSliceNodes + n;
                    if (a < n_own) m += slot_pair_ptr[(size_t)node_slot_ptr[a + 1]] - slot_pair_ptr[(size_t)node_slot_ptr[a]];
                }
                slice_pairs[(size_t)s + 1] = m;
            }
        });
        for (int32_t s = 0; s < p.n_slices; s++) slice_pairs[(size_t)s + 1] += slice_pairs[(size_t)s];
        p.pairs.resize((size_t)slice_pairs[(size_t)p.n_slices]);
        plan_parallel(p.n_slices, 256, [&](int, int64_t s0, int64_t s1) {
            for (int64_t s = s0; s < s1; s++) {
                const int w = p.slice_width[(size_t)s];
                int64_t at = slice_pairs[(size_t)s];
                for (int k = 0; k < w; k++)
                    for (int n = 0; n < kSliceNodes; n++) {
                        const int64_t idx = Plan::slot_index(p.slice_base[(size_t)s], k, n);
                        const int32_t a = (int32_t)s * kSliceNodes + n;
                        p.pair_ptr[(size_t)idx] = (int32_t)at;
                        if (real_slot(a, k)) {
                            const int32_t q = node_slot_ptr[a] + k;
                            p.cols[(size_t)idx] = to_local(slot_col[(size_t)q]);
                            const int32_t b = slot_pair_ptr[(size_t)q], e = slot_pair_ptr[(size_t)q + 1];
                            std::copy(slot_pairs.begin() + b, slot_pairs.begin() + e, p.pairs.begin() + at);
                            at += e - b;
                        } else {
                            p.cols[(size_t)idx] = std::min(a, p.n_pad - 1); // padding slot: zero block on the own row
                        }
                    }
            }
        });
    }
    p.pair_ptr[total] = (int32_t)p.pairs.size();

    lap("in-lists and in-slice products");
    // ---- symmetric storage: which stored blocks act on a row through their transpose
    p.in_width.assign(p.n_slices, 0);
    p.in_base.assign((size_t)p.n_slices + 1, 0);
    if (symmetric) {
        // per row c: the neighbours a whose row holds the block of the pair (nb_mine == 0 on c's side), in ascending slot
        // index of that block = ascending slice of the source row, then slot-major inside the slice -- a fixed order, so
        // the sums of the gather phase are reproducible.  Rows in parallel; the slot of (a, c) is found in a's sorted list.
        std::vector<int32_t> cnt((size_t)p.n_pad, 0);
        plan_parallel(n_own, 4096, [&](int, int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; c++) {
                int m = 0;
                for (int64_t q = nb_ptr[(size_t)c]; q < nb_ptr[(size_t)c + 1]; q++) m += nb_mine[(size_t)q] ? 0 : 1;
                cnt[(size_t)c] = m;
            }
        });
        for (int32_t s2 = 0; s2 < p.n_slices; s2++) {
            int w = 0;
            for (int n = 0; n < kSliceNodes; n++) w = std::max(w, cnt[(size_t)s2 * kSliceNodes + n]);
            p.in_width[s2] = w;
            p.max_in_width = std::max(p.max_in_width, w);
            p.in_base[s2 + 1] = p.in_base[s2] + (int64_t)w * kSliceNodes;
        }
        p.in_slots.resize((size_t)p.in_base[p.n_slices]);
        p.in_rows.resize((size_t)p.in_base[p.n_slices]);
        plan_fill(p.in_slots, (int32_t)-1);
        plan_fill(p.in_rows, (int32_t)0);
        plan_parallel(n_own, 4096, [&](int, int64_t c0, int64_t c1) {
            std::vector<std::pair<int64_t, int32_t>> src; // (slot index, source row)
            for (int64_t c = c0; c < c1; c++) {
                src.clear();
                for (int64_t q = nb_ptr[(size_t)c]; q < nb_ptr[(size_t)c + 1]; q++) {
                    if (nb_mine[(size_t)q]) continue;
                    const int32_t a = nb[(size_t)q];
                    // row a's slots: the diagonal, then ascending global columns
                    const int32_t *b = slot_col.data() + node_slot_ptr[(size_t)a] + 1, *e = slot_col.data() + node_slot_ptr[(size_t)a + 1];
                    const int k = 1 + (int)(std::lower_bound(b, e, g0 + (int32_t)c) - b);
                    src.emplace_back(Plan::slot_index(p.slice_base[a / kSliceNodes], k, a % kSliceNodes), a);
                }
                std::sort(src.begin(), src.end());
                const int32_t sc = (int32_t)(c / kSliceNodes), nc = (int32_t)(c % kSliceNodes);
                for (size_t j = 0; j < src.size(); j++) {
                    const size_t d = (size_t)(p.in_base[sc] + (int64_t)j * kSliceNodes + nc);
                    p.in_slots[d] = (int32_t)src[j].first;
                    p.in_rows[d] = src[j].second;
                }
            }
        });
        // Transposed products that stay inside a slice: a stored block (a, c) whose column c is a row of the same slice
        // hands u = K_ac^T x_a to row c through LDS inside the SpMV kernel instead of through HBM (48 bytes written by
        // the SpMV, read again by the kernel that collects the products).  loc_index: per slot, the position of its u
        // among the slice's in-slice blocks (255: the product leaves the slice); loc_list: per in-list entry the same
        // position (255: collect it from HBM); gat_slots: the in-list without the in-slice entries.
        p.loc_index.resize((size_t)total);
        p.loc_list.resize(p.in_slots.size());
        p.gat_slots.resize(p.in_slots.size());
        plan_fill(p.loc_index, (uint8_t)255);
        plan_fill(p.loc_list, (uint8_t)255);
        plan_parallel((int64_t)p.in_slots.size(), 1 << 20,
                      [&](int, int64_t b, int64_t e) { std::copy(p.in_slots.begin() + b, p.in_slots.begin() + e, p.gat_slots.begin() + b); });
        std::vector<int32_t> per_slice((size_t)p.n_slices, 0);
        plan_parallel(p.n_slices, 256, [&](int, int64_t s0, int64_t s1) {
            for (int64_t s2 = s0; s2 < s1; s2++) {
                int m = 0;
                for (int64_t e = p.in_base[s2]; e < p.in_base[s2 + 1]; e++) { // (k, n) order of the in-list
                    const int32_t slot = p.in_slots[(size_t)e];
                    if (slot < 0 || slot < p.slice_base[s2] || slot >= p.slice_base[s2 + 1] || m >= 255) continue;
                    p.loc_index[(size_t)slot] = (uint8_t)m;
                    p.loc_list[(size_t)e] = (uint8_t)m;
                    p.gat_slots[(size_t)e] = -1;
                    m++;
                }
                per_slice[(size_t)s2] = m;
            }
        });
        for (int32_t s2 = 0; s2 < p.n_slices; s2++) p.max_loc = std::max(p.max_loc, per_slice[(size_t)s2]);
    }

    lap("per-slice element lists");
    // ---- per-slice element lists and slice-relative 16-bit gather entries (chunks of slices on the host threads, each
    //      into lists of its own that are joined in slice order afterwards)
    p.slice_elem_ptr.assign((size_t)p.n_slices + 1, 0);
    p.pairs16.resize(p.pairs.size());
    {
        const int nchunks = plan_chunks(p.n_slices, 256);
        std::vector<RawVec<int32_t>> part_elems((size_t)nchunks), part_nodes((size_t)nchunks);
        std::vector<int32_t> part_max((size_t)nchunks, 0), part_bad((size_t)nchunks, 0);
        plan_parallel(p.n_slices, 256, [&](int t, int64_t s0, int64_t s1) {
            // local element ids of a slice within this span: bitmap + rank table (FEMSHELL_PLAN_DENSE_SPAN: the tests set 0 to
            // drive the sort-and-search path on small meshes; read per plan)
            const char *span_env = getenv("FEMSHELL_PLAN_DENSE_SPAN");
            const int64_t kDenseSpan = span_env ? atoll(span_env) : (int64_t)1 << 18;
            std::vector<int32_t> ids, placed;
            std::vector<uint64_t> bits;
            std::vector<uint16_t> pos_of;
            RawVec<int32_t> elems, nodes; // (the thread's own; handed over at the end, see the slot loop above)
            int32_t most = 0;
            elems.reserve((size_t)(s1 - s0) * 136);
            nodes.reserve((size_t)(s1 - s0) * 136 * 4);
            for (int64_t s = s0; s < s1; s++) {
                const int32_t q0 = p.pair_ptr[p.slice_base[s]], q1 = p.pair_ptr[p.slice_base[s + 1]];
                // the slice's elements, ascending: every element that touches a row of the slice contributes to that row's
                // diagonal slot, so the pairs of the first 32 slots (k = 0) name them all.  Local element ids of a slice lie
                // close together for numberings that follow the mesh: they are then marked in a bitmap and read off in order,
                // and a table indexed by (id - smallest id) answers the rank queries below; other slices sort and search.
                const int32_t qd = p.pair_ptr[p.slice_base[s] + kSliceNodes];
                int32_t lo = INT32_MAX, hi = -1;
                for (int32_t q = q0; q < qd; q++) {
                    const int32_t le = (int32_t)(p.pairs[q] >> 4);
                    lo = std::min(lo, le);
                    hi = std::max(hi, le);
                }
                const bool dense = hi >= lo && (int64_t)hi - lo < kDenseSpan;
                ids.clear();
                if (dense) {
                    const int32_t words = (hi - lo) / 64 + 1;
                    if ((int32_t)bits.size() < words) bits.resize((size_t)words, 0); // (all zero between slices)
                    for (int32_t q = q0; q < qd; q++) {
                        const int32_t d = (int32_t)(p.pairs[q] >> 4) - lo;
                        bits[(size_t)(d >> 6)] |= 1ull << (d & 63);
                    }
                    for (int32_t w = 0; w < words; w++) {
                        uint64_t b = bits[(size_t)w];
                        bits[(size_t)w] = 0;
                        while (b) {
                            ids.push_back(lo + w * 64 + __builtin_ctzll(b));
                            b &= b - 1;
                        }
                    }
                } else {
                    for (int32_t q = q0; q < qd; q++) ids.push_back((int32_t)(p.pairs[q] >> 4));
                    std::sort(ids.begin(), ids.end());
                    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                }
                if (ids.size() > 4095) {
                    part_bad[(size_t)t] = 1;
                    return;
                }
                most = std::max<int32_t>(most, (int32_t)ids.size());
                // LDS position of the records: even-ranked elements first, then the odd-ranked ones.  Neighbouring
                // lanes (neighbouring node rows) gather from neighbouring cells; mesh generators emit the two
                // triangles of a cell back to back, so in ascending order those records are two apart and a
                // wave's LDS reads fall on half of the banks; de-interleaved they are adjacent.
                const int32_t n_even = ((int32_t)ids.size() + 1) / 2;
                auto lds_pos = [&](int32_t sorted_pos) { return (sorted_pos & 1) ? n_even + (sorted_pos >> 1) : (sorted_pos >> 1); };
                if (dense) {
                    if (pos_of.size() < (size_t)(hi - lo + 1)) pos_of.resize((size_t)(hi - lo + 1));
                    for (size_t r = 0; r < ids.size(); r++) pos_of[(size_t)(ids[r] - lo)] = (uint16_t)lds_pos((int32_t)r);
                    for (int32_t q = q0; q < q1; q++) // (only ids of the slice are looked up: the table needs no clearing)
                        p.pairs16[q] = (uint16_t)(((uint32_t)pos_of[(size_t)((int32_t)(p.pairs[q] >> 4) - lo)] << 4) | (p.pairs[q] & 15u));
                } else {
                    for (int32_t q = q0; q < q1; q++) {
                        const int32_t le = (int32_t)(p.pairs[q] >> 4);
                        const int32_t idx = lds_pos((int32_t)(std::lower_bound(ids.begin(), ids.end(), le) - ids.begin()));
                        p.pairs16[q] = (uint16_t)((idx << 4) | (p.pairs[q] & 15u));
                    }
                }
                placed.resize(ids.size());
                for (size_t r = 0; r < ids.size(); r++) placed[lds_pos((int32_t)r)] = ids[r];
                elems.insert(elems.end(), placed.begin(), placed.end());
                for (int32_t le : placed) {
                    if (le < p.n_ltri()) {
                        for (int i = 0; i < 3; i++) nodes.push_back(p.tri_local[3ll * le + i]);
                        nodes.push_back(-1);
                    } else {
                        for (int i = 0; i < 4; i++) nodes.push_back(p.quad_local[4ll * (le - p.n_ltri()) + i]);
                    }
                }
                p.slice_elem_ptr[(size_t)s + 1] = (int32_t)placed.size(); // count; prefix sums below
            }
            part_max[(size_t)t] = most;
            part_elems[(size_t)t].swap(elems);
            part_nodes[(size_t)t].swap(nodes);
        });
        for (int t = 0; t < nchunks; t++) {
            if (part_bad[(size_t)t]) return fail("more than 4095 elements touch one 32-node slice");
            p.max_slice_elems = std::max(p.max_slice_elems, part_max[(size_t)t]);
        }
        for (int32_t s = 0; s < p.n_slices; s++) p.slice_elem_ptr[(size_t)s + 1] += p.slice_elem_ptr[(size_t)s];
        p.slice_elems.resize((size_t)p.slice_elem_ptr[(size_t)p.n_slices]);
        p.slice_elem_nodes.resize(4 * p.slice_elems.size());
        std::vector<size_t> off((size_t)nchunks + 1, 0); // chunks are contiguous slice ranges in thread order
        for (int t = 0; t < nchunks; t++) off[(size_t)t + 1] = off[(size_t)t] + part_elems[(size_t)t].size();
        plan_parallel(nchunks, 1, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; t++) {
                std::copy(part_elems[(size_t)t].begin(), part_elems[(size_t)t].end(), p.slice_elems.begin() + off[(size_t)t]);
                std::copy(part_nodes[(size_t)t].begin(), part_nodes[(size_t)t].end(), p.slice_elem_nodes.begin() + 4 * off[(size_t)t]);
                RawVec<int32_t>().swap(part_elems[(size_t)t]);
                RawVec<int32_t>().swap(part_nodes[(size_t)t]);
            }
        });
    }

    lap("assembly work items");
    // ---- assembly work items
    // contributions per item (FEMSHELL_ITEM_PAIRS overrides): 3 fills the four waves of a full-storage slice evenly (256
    // items); with symmetric storage a structured slice has 32 diagonal slots of 6 and 96 off-diagonal slots of 2
    // contributions, and items of 2 (192 equal items, three full waves while the fourth builds records) were measured
    // against items of 3 (160 items): 0.904 vs 0.875 ms -- fewer items and partial-sum rows win, 3 stays
    static const int item_pairs_env = getenv("FEMSHELL_ITEM_PAIRS") ? atoi(getenv("FEMSHELL_ITEM_PAIRS")) : 0;
    const int item_pairs = (item_pairs_env >= 1 && item_pairs_env <= kItemPairs) ? item_pairs_env : kItemPairs;
    // Item layout of the pipelined kernel (k_assemble_pipe, assemble_kernel.hpp): rounds of 192 lanes, the chunks of a
    // slot in neighbouring lanes of one wave, diagonal slots in waves of their own where the round has room.  For meshes of
    // triangles whose slices leave room for two record buffers per workgroup, two workgroups per CU; FEMSHELL_ASM_PIPE=0
    // keeps the two-phase kernel (read per plan).
    p.pipe = false;
    {
        const char *e = getenv("FEMSHELL_ASM_PIPE");
        const bool wanted = !(e && atoi(e) == 0);
        int32_t max_cnt = 0;
        for (size_t i = 0; i + 1 < p.pair_ptr.size(); i++) max_cnt = std::max(max_cnt, p.pair_ptr[i + 1] - p.pair_ptr[i]);
        p.pipe = wanted && !p.slice_elem_nodes.empty() && // (the kernel's idle lanes read a valid element: there must be one)
                 p.max_slice_elems <= (p.n_lquad() > 0 ? kPipeMaxSliceElemsQuad : kPipeMaxSliceElems) &&
                 item_pairs >= 2 && (max_cnt + 1) / 2 <= 64; // (a slot's chunks -- of two contributions at the least -- share
                                                             //  a wave; FEMSHELL_ITEM_PAIRS=1 cuts chunks of one: two-phase kernel)
        // ... and whose slices fit one round of the three consumer waves: the diagonal slots' chunks of three in the first,
        // the rest -- diagonal slots beyond 64 lanes as chunks of two -- in the other two.  Structured meshes (64 + 96
        // lanes): 0.57 against 0.72 ms at 4M triangles; Delaunay meshes (valences 3..12: 74 + 96 chunks per slice): 5.1
        // against 4.5 G elements/s at 1M triangles.  A second round per slice (full storage: 192 off-diagonal slots per
        // structured slice) is left to the two-phase kernel.  FEMSHELL_ASM_PIPE=2 takes the layout wherever the kernel can
        // run.
        if (p.pipe && !(e && atoi(e) == 2)) {
            std::vector<int64_t> part((size_t)plan_chunks(p.n_slices, 256), 0);
            plan_parallel(p.n_slices, 256, [&](int t, int64_t s0, int64_t s1) {
                int64_t r = 0;
                for (int64_t s = s0; s < s1; s++) {
                    int diag = 0, off = 0;
                    for (int k = 0; k < p.slice_width[s]; k++)
                        for (int n = 0; n < kSliceNodes; n++) {
                            const int64_t idx = Plan::slot_index(p.slice_base[s], k, n);
                            const int cnt = p.pair_ptr[idx + 1] - p.pair_ptr[idx];
                            (k == 0 ? diag : off) += (cnt + item_pairs - 1) / item_pairs;
                        }
                    const int beyond = diag > 64 ? ((diag - 64) * 3 + 1) / 2 : 0; // (chunks of two in the other waves)
                    r += std::min(diag, 64) + beyond + off > 192;
                }
                part[(size_t)t] += r;
            });
            int64_t ragged = 0;
            for (int64_t r : part) ragged += r;
            if (20 * ragged > p.n_slices) p.pipe = false;
        }
    }
    p.item_ptr.assign((size_t)p.n_slices + 1, 0);
    {
        // chunks of slices on the host threads, each into an item list of its own, joined in slice order afterwards
        const int nchunks_t = plan_chunks(p.n_slices, 256);
        std::vector<RawVec<Plan::Item>> part_items((size_t)nchunks_t);
        std::vector<int32_t> part_stage((size_t)nchunks_t, 0), part_bad((size_t)nchunks_t, 0);
        plan_parallel(p.n_slices, 256, [&](int t, int64_t s0, int64_t s1) {
            std::vector<Plan::Item> tmp, sorted, packed;
            RawVec<Plan::Item> out; // (the thread's own; handed over at the end)
            int32_t most_stage = 0;
            out.reserve((size_t)(s1 - s0) * 168);
            // stable order by decreasing number of contributions (0..kItemPairs): a bucket pass, no allocation
            auto order_by_work = [&](std::vector<Plan::Item> &v, size_t begin) {
                sorted.clear();
                for (int np = kItemPairs; np >= 0; np--)
                    for (size_t i = begin; i < v.size(); i++)
                        if ((int)(v[i].z >> 16) == np) sorted.push_back(v[i]);
                std::copy(sorted.begin(), sorted.end(), v.begin() + begin);
            };
            for (int64_t s = s0; s < s1; s++) {
                tmp.clear();
                int32_t stage = 0;
                const int w = p.slice_width[s];
                for (int k = 0; k < w; k++)
                    for (int n = 0; n < kSliceNodes; n++) {
                        const int64_t idx = Plan::slot_index(p.slice_base[s], k, n);
                        const int32_t q0 = p.pair_ptr[idx], cnt = p.pair_ptr[idx + 1] - q0;
                        if (cnt == 0) continue; // padding slot: its zero block is written once, when K is allocated
                        const int nchunks = (cnt + item_pairs - 1) / item_pairs;
                        if (nchunks > 255) {
                            part_bad[(size_t)t] = 1;
                            return;
                        }
                        const int32_t stage0 = stage;
                        for (int c = 0; c < nchunks; c++) {
                            const int32_t b = q0 + c * item_pairs, e = std::min(q0 + cnt, b + item_pairs);
                            const int np = std::max(0, e - b);
                            uint32_t pr[3] = {0, 0, 0};
                            for (int q = 0; q < np; q++) pr[q] = p.pairs16[b + q];
                            Plan::Item it;
                            it.x = (uint32_t)(k * kSliceNodes + n) | ((uint32_t)c << 16) | ((uint32_t)nchunks << 24);
                            it.y = pr[0] | (pr[1] << 16);
                            it.z = pr[2] | ((uint32_t)np << 16);
                            it.w = (uint32_t)(c == 0 ? stage0 : stage0 + c - 1);
                            tmp.push_back(it);
                        }
                        stage += nchunks - 1;
                    }
                if (p.pipe) {
                    pack_items_pipe(tmp, &packed);
                    tmp.swap(packed);
                    stage = 0;
                } else if (tmp.size() <= 256) {
                    // one round: order by decreasing work so that the waves are uniform
                    order_by_work(tmp, 0);
                } else {
                    // several rounds of 256 items: a slot's chunks must share a round (they meet in LDS),
                    // so fill rounds greedily with whole slots, then order each round by work
                    packed.clear();
                    size_t i = 0;
                    while (i < tmp.size()) {
                        const size_t round_begin = packed.size();
                        while (i < tmp.size()) {
                            const size_t nch = tmp[i].x >> 24;
                            if (packed.size() - round_begin + nch > 256) break;
                            for (size_t c = 0; c < nch; c++) packed.push_back(tmp[i + c]);
                            i += nch;
                        }
                        order_by_work(packed, round_begin);
                        if (i < tmp.size()) {
                            Plan::Item pad{0xffffu, 0, 0, 0}; // inert item: slot 0xffff, chunk 0, 0 chunks
                            while (packed.size() - round_begin < 256) packed.push_back(pad);
                        }
                    }
                    tmp.swap(packed);
                }
                most_stage = std::max(most_stage, stage);
                out.insert(out.end(), tmp.begin(), tmp.end());
                p.item_ptr[(size_t)s + 1] = (int32_t)tmp.size(); // count; prefix sums below
            }
            part_stage[(size_t)t] = most_stage;
            part_items[(size_t)t].swap(out);
        });
        size_t total_items = 0;
        for (int t = 0; t < nchunks_t; t++) {
            if (part_bad[(size_t)t]) return fail("a block slot has more than 765 contributions");
            p.max_stage_rows = std::max(p.max_stage_rows, part_stage[(size_t)t]);
            total_items += part_items[(size_t)t].size();
        }
        if (total_items > (size_t)0x7fffffff) return fail("more than 2^31 assembly work items on one rank");
        for (int32_t s = 0; s < p.n_slices; s++) p.item_ptr[(size_t)s + 1] += p.item_ptr[(size_t)s];
        p.items.resize(total_items);
        std::vector<size_t> off((size_t)nchunks_t + 1, 0);
        for (int t = 0; t < nchunks_t; t++) off[(size_t)t + 1] = off[(size_t)t] + part_items[(size_t)t].size();
        plan_parallel(nchunks_t, 1, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; t++) {
                std::copy(part_items[(size_t)t].begin(), part_items[(size_t)t].end(), p.items.begin() + off[(size_t)t]);
                RawVec<Plan::Item>().swap(part_items[(size_t)t]);
            }
        });
    }
    lap("slice descriptors");
    // ---- per-slice descriptors of the assembly kernel
    p.slice_desc.assign((size_t)p.n_slices * 8, 0);
    for (int32_t s = 0; s < p.n_slices; s++) {
        int32_t *d = &p.slice_desc[(size_t)s * 8];
        d[0] = p.slice_elem_ptr[s];
        d[1] = p.slice_elem_ptr[s + 1] - p.slice_elem_ptr[s];
        d[2] = p.item_ptr[s];
        d[3] = p.item_ptr[s + 1] - p.item_ptr[s];
        d[4] = (int32_t)(uint32_t)((uint64_t)p.slice_base[s] & 0xffffffffu);
        d[5] = (int32_t)(uint32_t)((uint64_t)p.slice_base[s] >> 32);
        d[6] = p.slice_width[s];
    }

    lap("halo exchange lists");
    // ---- halo exchange lists
    if (world > 1) {
        // receive side: ghosts grouped by owner (ghost_global is ascending, ranges are contiguous)
        std::vector<HaloPeer> peers;
        for (int32_t q = 0; q < p.n_ghost;) {
            const int r = owner_of(p.ghost_global[q], p.part_bounds);
            HaloPeer hp;
            hp.rank = r;
            hp.recv_offset = q;
            while (q < p.n_ghost && owner_of(p.ghost_global[q], p.part_bounds) == r) q++;
            hp.recv_count = q - hp.recv_offset;
            peers.push_back(hp);
        }
        // send side: owned node a goes to rank r iff a has a neighbour owned by r (adjacency is symmetric)
        std::vector<std::vector<int32_t>> send(world);
        for (int32_t a = 0; a < n_own; a++) {
            int last = -1;
            for (int32_t q = node_slot_ptr[a] + 1; q < node_slot_ptr[a + 1]; q++) {
                const int32_t c = slot_col[q];
                if (c >= g0 && c < g1) continue;
                const int r = owner_of(c, p.part_bounds);
                if (r != last && (send[r].empty() || send[r].back() != a)) send[r].push_back(a);
                last = r;
            }
        }
        for (int r = 0; r < world; r++) {
            if (send[r].empty()) continue;
            auto it = std::find_if(peers.begin(), peers.end(), [&](const HaloPeer &h) { return h.rank == r; });
            if (it == peers.end()) {
                HaloPeer hp;
                hp.rank = r;
                peers.push_back(hp);
                it = peers.end() - 1;
            }
            it->send_nodes.swap(send[r]);
        }
        std::sort(peers.begin(), peers.end(), [](const HaloPeer &x, const HaloPeer &y) { return x.rank < y.rank; });
        p.peers.swap(peers);
    }
    // SpMV order: slices whose blocks couple to owned columns only come first; they can be multiplied
    // while the halo exchange is in flight
    {
        std::vector<int32_t> boundary;
        p.spmv_order.clear();
        for (int32_t s = 0; s < p.n_slices; s++) {
            bool ghost = false;
            for (int64_t i = p.slice_base[s]; i < p.slice_base[s + 1] && !ghost; i++) ghost = p.cols[i] >= p.n_pad;
            (ghost ? boundary : p.spmv_order).push_back(s);
        }
        p.n_interior_slices = (int32_t)p.spmv_order.size();
        p.spmv_order.insert(p.spmv_order.end(), boundary.begin(), boundary.end());
    }
    lap("done");
    return true;
}

} // namespace femshell
