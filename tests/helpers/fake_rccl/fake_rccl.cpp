// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that moves data between
// PROCESSES THAT SHARE ONE GPU through POSIX shared memory (device -> host -> shm -> host -> device).
//
// A test box has one MI355X, and RCCL refuses two ranks on one device, so the multi-rank driver of
// libfemshell (halo packing, ghost placement, all-reduce points, row gather -- csrc/api.cpp,
// csrc/comm.cpp) could otherwise never run before the 8-GPU benchmark.  libfemshell opens the library
// named by FEMSHELL_RCCL_LIB instead of librccl.so.1 when that variable is set (tests only).
// Implements exactly the entry points comm.cpp resolves, with RCCL's signatures and semantics
// (stream-ordered, grouped send/recv/broadcast); reductions add the ranks' contributions in rank order.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace {

constexpr size_t kSlotBytes = 64u << 20; // per (rank) exchange slot (pages are touched on use only; the 4M-triangle rehearsal reduces 11 MB)
constexpr int kMaxRanks = 8;

struct Shared {
    std::atomic<int> arrive;
    std::atomic<int> generation;
    char pad[56];
    // slots follow: kMaxRanks x kMaxRanks message boxes of kSlotBytes would be too big; messages are
    // serialised instead: one slot per rank, one transfer step at a time
};

struct FakeComm {
    Shared *sh = nullptr;
    char *slots = nullptr; // kMaxRanks * kSlotBytes
    int rank = 0, nranks = 1;
    std::string name;
    size_t bytes = 0;
};

struct Op {
    int kind; // 0 send, 1 recv, 2 broadcast
    const void *send;
    void *recv;
    size_t bytes;
    int peer; // peer rank / root
    FakeComm *comm;
    hipStream_t stream;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;
FakeComm *g_comm = nullptr; // the process's communicator: a group without operations still takes part in the pair walk

void barrier(FakeComm *c)
{
    const int gen = c->sh->generation.load(std::memory_order_acquire);
    if (c->sh->arrive.fetch_add(1, std::memory_order_acq_rel) == c->nranks - 1) {
        c->sh->arrive.store(0, std::memory_order_relaxed);
        c->sh->generation.store(gen + 1, std::memory_order_release);
    } else {
        while (c->sh->generation.load(std::memory_order_acquire) == gen) sched_yield();
    }
}

char *slot(FakeComm *c, int r) { return c->slots + (size_t)r * kSlotBytes; }

ncclResult_t run_ops()
{
    // every rank issues the same NUMBER of steps: sends/recvs are matched pairwise by walking all
    // (from, to) pairs in a fixed global order; broadcasts in call order
    // (a rank whose lists for a coarse level's halo are empty issues an empty group while its peers issue theirs)
    if (g_ops.empty() && g_comm == nullptr) return ncclSuccess;
    FakeComm *c = g_ops.empty() ? g_comm : g_ops[0].comm;
    if (!g_ops.empty() && hipStreamSynchronize(g_ops[0].stream) != hipSuccess) return ncclUnhandledCudaError;
    // broadcasts first, in call order (comm.cpp never mixes them with send/recv in one group)
    for (const Op &op : g_ops)
        if (op.kind == 2) {
            if (op.bytes > kSlotBytes) return ncclInvalidArgument;
            if (c->rank == op.peer) {
                if (hipMemcpy(slot(c, op.peer), op.send, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            }
            barrier(c);
            if (hipMemcpy(op.recv, slot(c, op.peer), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            barrier(c);
        }
    // point-to-point: step (from, to) for all ordered pairs; a rank takes part if it has the matching op
    bool any_p2p = false;
    for (const Op &op : g_ops) any_p2p |= (op.kind != 2);
    // all ranks must agree on whether this group has p2p traffic: comm.cpp's halo groups always do on
    // every rank of a multi-rank run (a rank without neighbours still calls the group) -> walk pairs always
    if (any_p2p || true) {
        for (int from = 0; from < c->nranks; from++)
            for (int to = 0; to < c->nranks; to++) {
                if (from == to) continue;
                const Op *snd = nullptr, *rcv = nullptr;
                for (const Op &op : g_ops) {
                    if (op.kind == 0 && c->rank == from && op.peer == to) snd = &op;
                    if (op.kind == 1 && c->rank == to && op.peer == from) rcv = &op;
                }
                if (snd) {
                    if (snd->bytes > kSlotBytes) return ncclInvalidArgument;
                    if (hipMemcpy(slot(c, from), snd->send, snd->bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
                }
                barrier(c);
                if (rcv) {
                    if (hipMemcpy(rcv->recv, slot(c, from), rcv->bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
                }
                barrier(c);
            }
    }
    g_ops.clear();
    return ncclSuccess;
}

bool group_has_only_broadcasts()
{
    for (const Op &op : g_ops)
        if (op.kind != 2) return false;
    return true;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::random_device rd;
    std::memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fsfake_%08x%08x", rd(), (unsigned)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks > kMaxRanks) return ncclInvalidArgument;
    FakeComm *c = new FakeComm();
    c->rank = rank;
    c->nranks = nranks;
    c->name = id.internal;
    c->bytes = sizeof(Shared) + (size_t)kMaxRanks * kSlotBytes;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError; // zero-filled: counters start at 0
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->sh = static_cast<Shared *>(p);
    c->slots = static_cast<char *>(p) + sizeof(Shared);
    *comm = reinterpret_cast<ncclComm_t>(c);
    g_comm = c;
    barrier(c); // everybody has mapped the segment
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name.c_str());
    if (g_comm == c) g_comm = nullptr;
    munmap(c->sh, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    *count = reinterpret_cast<const FakeComm *>(comm)->nranks;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (datatype != ncclDouble || op != ncclSum || count * 8 > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(slot(c, c->rank), sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < c->nranks; r++) {
        const double *v = reinterpret_cast<const double *>(slot(c, r));
        for (size_t i = 0; i < count; i++) sum[i] += v[i];
    }
    barrier(c); // everybody has read the slots
    if (hipMemcpy(recvbuff, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    g_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (--g_group_depth > 0) return ncclSuccess;
    if (group_has_only_broadcasts()) {
        // run_ops handles broadcasts and would then walk the (empty) p2p pairs with barriers on every rank
    }
    return run_ops();
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclDouble) return ncclInvalidArgument;
    g_ops.push_back({0, sendbuff, nullptr, count * 8, peer, reinterpret_cast<FakeComm *>(comm), stream});
    return g_group_depth ? ncclSuccess : ncclInvalidUsage;
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclDouble) return ncclInvalidArgument;
    g_ops.push_back({1, nullptr, recvbuff, count * 8, peer, reinterpret_cast<FakeComm *>(comm), stream});
    return g_group_depth ? ncclSuccess : ncclInvalidUsage;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root,
                           ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclDouble) return ncclInvalidArgument;
    g_ops.push_back({2, sendbuff, recvbuff, count * 8, root, reinterpret_cast<FakeComm *>(comm), stream});
    if (g_group_depth) return ncclSuccess;
    return run_ops();
}

} // extern "C"
