// plan.hpp -- host-side symbolic phase of libfemshell.
//
// The reference leaves this to libMesh (EquationSystems::init builds the dof map and the
// sparsity pattern, fem-shell.cpp:125) and to PETSc (MatSetValues stash, matrix->close()).
// Here it produces the device data layout once per mesh:
//
//  * row ownership: node rows are split into contiguous ranges, one per rank (GPU), in
//    units of 32-node slices; a rank keeps its owned nodes, the ghost nodes they couple to
//    and every element that touches an owned node (interface elements are recomputed on
//    both sides, so assembly needs no communication);
//  * K in "sliced block ELL": a slice is 32 consecutive node rows = 192 scalar rows; every
//    node row has `width` block slots; slot 0 is the diagonal block, the others are the
//    neighbour blocks in ascending column order; values are stored so that the 192 scalar
//    rows of a slice are the fastest index (one lane per scalar row, 16-byte loads):
//        vals[(slot_base(s) + k*32)*36 + ((jp*6 + i)*32 + n)*2 + jj],  j = 2*jp + jj
//    for slice s, slot k, node-in-slice n, block row i, block column j;
//  * gather lists: for every block slot the (element, local row node, local column node)
//    triples that contribute to it, ordered by element id, so that the assembly kernel
//    writes every block exactly once, without atomics and in a fixed summation order.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <functional>
#include <vector>

namespace femshell {

// called between the phases of build_plan when set: libfemshell points it at CommWatch::heartbeat (comm.hpp), so that a long
// symbolic phase on a large mesh or a busy host is progress in the eyes of the watchdog of a multi-rank context; the host-only
// sanitizer builds leave it null
extern void (*plan_progress_hook)();


// The large arrays of a plan (hundreds of MB at 4M triangles): resize() leaves new elements uninitialised instead of
// zero-filling them on the calling thread, so that the host threads that write an array are also the ones that take its
// page faults.  Every element is written before it is read (fills where a value is needed are explicit and parallel).
// Allocations of 4 MiB and more are aligned to 2 MiB and marked MADV_HUGEPAGE: where the kernel hands out transparent huge
// pages on request (the usual "madvise" setting) the first touch of such an array takes one page fault per 2 MiB instead of
// 512 -- the first femshell_set_mesh of a process, whose heap is cold, spent a third of its time in page faults.
void *raw_allocate(std::size_t bytes);
void raw_deallocate(void *p, std::size_t bytes) noexcept;
template <class T> struct NoInitAlloc {
    using value_type = T;
    template <class U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(std::size_t n) { return static_cast<T *>(raw_allocate(n * sizeof(T))); }
    void deallocate(T *p, std::size_t n) noexcept { raw_deallocate(p, n * sizeof(T)); }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

constexpr int kSliceNodes = 32;             // node rows per slice
constexpr int kSliceRows = 6 * kSliceNodes; // scalar rows per slice
constexpr int kItemPairs = 3;                // element contributions per assembly work item
// most elements a slice may touch for the pipelined assembly kernel: two buffers of 34-double records per workgroup, two
// workgroups in the 160 KiB of a CU
constexpr int kPipeMaxSliceElems = 150;
constexpr int kPipeMaxSliceElemsQuad = 77; // meshes with quadrilaterals: 66-double records

struct HaloPeer {
    int rank = -1;
    int32_t recv_offset = 0;          // first ghost index (into the ghost list) owned by this peer
    int32_t recv_count = 0;           // ghost nodes received from this peer
    std::vector<int32_t> send_nodes;  // owned local node ids this peer needs, ascending
};

struct Plan {
    // global problem
    int32_t n_nodes = 0, n_tri = 0, n_quad = 0;
    int rank = 0, world = 1;
    int32_t row_begin = 0, row_end = 0; // owned global node range
    std::vector<int32_t> part_bounds;   // world + 1 row boundaries of the partition (partition_bounds)
    // local numbering: owned [0,n_own), padding [n_own,n_pad), ghosts [n_pad, n_pad+n_ghost)
    int32_t n_own = 0, n_pad = 0, n_ghost = 0;
    std::vector<int32_t> ghost_global;  // ascending global ids
    // local elements (those touching an owned node), local node ids
    std::vector<int32_t> tri_global_id, quad_global_id;
    RawVec<int32_t> tri_local, quad_local; // 3*n_ltri, 4*n_lquad
    RawVec<double> xyz_local;              // (n_pad+n_ghost)*3
    // sliced block ELL structure over the owned rows
    int32_t n_slices = 0;
    std::vector<int32_t> slice_width;  // n_slices
    std::vector<int64_t> slice_base;   // n_slices+1, in slots (one slot = one 6x6 block of one node)
    RawVec<int32_t> cols;         // per slot: local column node id (padding slots: own row, no pairs)
    RawVec<int32_t> pair_ptr;     // per slot + 1
    RawVec<uint32_t> pairs;       // (local element << 4) | (row node in element << 2) | column node in element
                                       // local element index: triangles [0,n_ltri), quads n_ltri + q
    // per slice: the distinct local elements its gather lists reference (ascending); the device
    // kernel stages one record per such element in LDS and addresses it with the 16-bit entries
    //   pairs16 = (index into the slice's element list << 4) | (row node << 2) | column node
    std::vector<int32_t> slice_elem_ptr; // n_slices+1
    RawVec<int32_t> slice_elems;
    RawVec<uint16_t> pairs16;       // same order as pairs
    int32_t max_slice_elems = 0;
    RawVec<int32_t> slice_elem_nodes; // 4 local node ids per entry of slice_elems (4th = -1 for TRI3)
    // work items of the assembly kernel: a block slot's gather list is cut into chunks of at most
    // kItemPairs contributions so that every lane has the same amount of work; chunk 0 owns the
    // slot (it adds the other chunks' partial sums in chunk order and writes the block).  Per
    // slice the items are ordered by decreasing contribution count.  An item carries its
    // contributions (pairs16 entries) itself, so the kernel has no dependent index loads:
    //   x = slot in slice (k*32+n) | chunk index << 16 | number of chunks << 24
    //   y = pair 0 | pair 1 << 16
    //   z = pair 2 | pair count << 16
    //   w = LDS staging row of this chunk's partial sum (chunks > 0), or first staging row of
    //       the slot's other chunks (chunk 0)
    struct Item { uint32_t x, y, z, w; };
    std::vector<int32_t> item_ptr; // n_slices+1
    std::vector<int32_t> slice_desc; // 8 per slice: elem begin, elem count, item begin, item count, slot base lo, hi, width, 0
    RawVec<Item> items;
    int32_t max_stage_rows = 0;    // staging rows (36 doubles each) a slice needs at most
    // pipe == true: the items are laid out for the pipelined kernel (plan.cpp pack_items_pipe: rounds of 192 lanes,
    // a slot's chunks in neighbouring lanes of one wave, w = wave word) and no staging rows are needed
    bool pipe = false;
    int32_t max_slice_width = 0;
    int64_t nnz_blocks = 0;            // blocks of the owned rows of K (what femshell_export_bsr returns)
    // Symmetric storage (default; FEMSHELL_SYMMETRIC=0 stores every block): K = K^T, so of an off-diagonal pair
    // (a,c), (c,a) with both nodes owned only one block is assembled, stored and streamed -- with the row that has
    // fewer blocks (plan.cpp: balanced orientation; on structured grids the lower-numbered row) -- and the other
    // one acts through its transpose.  Blocks whose column is a ghost node stay (the owner of the
    // column has its own copy and nobody applies a transpose across ranks).  The SpMV kernel writes the products
    // K_ac^T x_a next to the block's slot (6 doubles) and the rows they belong to collect them through `in_slots`:
    // per slice in_width[s] entries per node row, each the slot index of a stored block (a, this row) or -1.
    bool symmetric = false;
    int64_t stored_blocks = 0;         // blocks that have a slot (== nnz_blocks without symmetric storage)
    std::vector<int32_t> in_width;     // n_slices
    std::vector<int64_t> in_base;      // n_slices+1, in entries
    RawVec<int32_t> in_slots;     // entry (in_base[s] + k*32 + n): slot index, -1 = none
    RawVec<int32_t> in_rows;      // same shape: the local row a of that block (its x entries multiply the transpose)
    int32_t max_in_width = 0;
    // transposed products that stay inside a slice go through LDS (plan.cpp): per slot / per in-list entry the position
    // among the slice's in-slice blocks (255 = none), the in-list without those entries, and the largest count of a slice
    RawVec<uint8_t> loc_index, loc_list;
    RawVec<int32_t> gat_slots;
    int32_t max_loc = 0;
    std::vector<HaloPeer> peers;
    // slices in SpMV order: the first n_interior_slices read no ghost column (they overlap the halo exchange)
    std::vector<int32_t> spmv_order;
    int32_t n_interior_slices = 0;

    int64_t total_slots() const { return slice_base.empty() ? 0 : slice_base.back(); }
    int32_t n_local_nodes() const { return n_pad + n_ghost; }
    int32_t n_ltri() const { return (int32_t)tri_global_id.size(); }
    int32_t n_lquad() const { return (int32_t)quad_global_id.size(); }
    // slot index of (slice s, slot k, node n)
    static inline int64_t slot_index(int64_t base, int k, int n) { return base + (int64_t)k * kSliceNodes + n; }
};

// CPUs this process may run on (its affinity mask; a pinned or cpuset-confined rank does not start a thread per core of
// the machine), at least 1
int available_cpus();
// threads of the host-side symbolic work (plan.cpp): FEMSHELL_HOST_THREADS, else available_cpus() / ranks on the host, <= 64
int host_thread_count();
// task(t) for t = 0 .. nt - 1, each on a thread of its own (task 0 on the caller's), returns when all are through.  The threads
// are a pool that lives as long as the process: the symbolic phase and the multigrid setup make some sixty such calls per solve,
// and sixteen std::thread spawns and joins per call were a fifth of the multigrid setup.  A call from inside a task runs its tasks one after
// the other; a call while another host thread uses the pool runs on threads of its own as before.
void run_on_host_threads(int nt, const std::function<void(int)> &task);
void set_host_share(int ranks_on_this_host);

// owned node range of `rank` when n_nodes rows are split over `world` ranks
void partition_rows(int32_t n_nodes, int world, int rank, int32_t *begin, int32_t *end);
// Row boundaries of the `world` ranks, in whole slices of 32 nodes: contiguous ranges of the caller's numbering with equal
// shares of the element incidences (valence + 1 per node: blocks of K per row, contributions to assemble).  On a regular
// grid this is the equal split of partition_rows; on meshes with varying valence it balances work instead of rows.
void partition_bounds(int32_t n_nodes, int32_t n_tri, const int32_t *tri, int32_t n_quad, const int32_t *quad, int world,
                      std::vector<int32_t> *bounds);

// Builds the plan.  Returns false and sets err on invalid input (index out of range,
// repeated node in an element, too many elements).
bool build_plan(int32_t n_nodes, const double *xyz, int32_t n_tri, const int32_t *tri, int32_t n_quad,
                const int32_t *quad, int rank, int world, Plan *plan, std::string *err, bool symmetric = false,
                bool geometric_orientation = false);
// the library's default storage: symmetric unless FEMSHELL_SYMMETRIC=0
bool default_symmetric_storage();

} // namespace femshell
