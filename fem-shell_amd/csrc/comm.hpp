// comm.hpp -- RCCL (xGMI) communication of the row-partitioned solve.
//
// Replaces the implicit MPI traffic of the reference: PETSc's VecScatter halo exchange and
// MPI_Allreduce inside KSPSolve, and the gather/broadcast of build_solution_vector
// (fem-shell.cpp:141; fem-shell_precice.cpp:274-280).  Assembly needs no communication
// (interface elements are recomputed on both sides), so there is no counterpart of PETSc's
// MatSetValues stash exchange.
//
// librccl is opened with dlopen when a multi-rank context initialises its communicator, so
// single-GPU use has no RCCL dependency.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "plan.hpp"

namespace femshell {

struct Comm {
    void *lib = nullptr;
    void *comm = nullptr; // ncclComm_t
    int rank = 0, world = 1;
    bool active() const { return comm != nullptr; }
    // what has been enqueued on this communicator since the counters were last cleared (femshell_comm_counters): grouped send/recv
    // exchanges on the stream registered as `second` (the halo stream: they run beside kernels of the main stream) and on any
    // other stream (they sit in the main stream's dependency chain), all-reduces, grouped broadcasts (row gathers)
    hipStream_t second = nullptr;
    int64_t halo_groups_second = 0, halo_groups_main = 0, allreduces = 0, gathers = 0;
    int64_t halo_bytes_sent = 0, collective_bytes = 0; // (femshell_comm_bytes: what this rank handed to sends / to all-reduces and broadcasts)
};

// Hang protection of multi-rank contexts.  A rank that waits for a peer that never joined (ncclCommInitRank), or for a
// collective a stalled peer never enters, waits forever -- RCCL has no timeout of its own here and a blocked call cannot be
// cancelled.  The library's blocking phases are therefore watched: while a CommWatch is alive a monitor thread ends the
// PROCESS (exit status 86, one line on stderr naming rank, phase and what to try) once the phase has made no progress for
// FEMSHELL_COMM_TIMEOUT seconds (default 120; 0 switches the watch off).  A launcher that sees one rank exit tears the
// group down; nothing is ever re-executed.  heartbeat(): progress inside a long phase (the CG loops' polls, the laps of the
// symbolic phase and of the multigrid setup, every host-side agreement of the ranks).  Watches are per thread: contexts on
// different threads have their own phases and timers, and the time limit is read once when a watch opens, by its own thread.
struct CommWatch {
    CommWatch(int rank, int world, const char *phase);
    ~CommWatch();
    CommWatch(const CommWatch &) = delete;
    CommWatch &operator=(const CommWatch &) = delete;
    static void heartbeat();

  private:
    void *entry_; // this watch's slot in the registry the monitor thread walks (nullptr: single-rank context, nothing to watch)
};

bool comm_unique_id(uint8_t id_out[128], std::string *err);
bool comm_init(Comm &c, const uint8_t id[128], int rank, int world, std::string *err);
void comm_destroy(Comm &c);
int comm_count(const Comm &c); // ranks RCCL itself reports for the communicator (ncclCommCount); 0 without one
// first-contact self-test of the patterns a solve uses (comm.cpp): grouped send/recv on a second stream beside an all-reduce on
// the main one, grouped broadcasts, a lone all-reduce; each watched and timed (us_out[3], microseconds); scratch: 16 + 2 x world
// doubles of device memory
bool comm_selftest(Comm &c, hipStream_t main, hipStream_t second, double *scratch, double us_out[3], std::string *err);

// in-place sum of `count` doubles across ranks, stream-ordered
bool comm_allreduce_sum(Comm &c, double *buf, int count, hipStream_t st, std::string *err);

// one grouped send/recv per peer: sendbuf holds the packed owned entries peer after peer
// (send_offsets in nodes), ghosts of `p` start at p + 6*n_pad
// (width: doubles per node)
bool comm_halo(Comm &c, const std::vector<HaloPeer> &peers, const std::vector<int32_t> &send_offsets,
               const double *sendbuf, double *p_ghost, hipStream_t st, std::string *err, int width = 6);

// all ranks receive every rank's owned rows: full[6*row_begin(r) ...] <- rank r's x
bool comm_gather_rows(Comm &c, const double *x_owned, double *full, const std::vector<int32_t> &row_begin,
                      const std::vector<int32_t> &row_end, hipStream_t st, std::string *err);
// the same for pieces of any length: full[begin[r] .. end[r]) <- rank r's `mine` (doubles)
bool comm_gather_pieces(Comm &c, const double *mine, double *full, const std::vector<int64_t> &begin,
                        const std::vector<int64_t> &end, hipStream_t st, std::string *err);

} // namespace femshell
