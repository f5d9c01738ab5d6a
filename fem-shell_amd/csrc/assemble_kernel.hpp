// assemble_kernel.hpp -- the assembly kernel template (included by kernels.hip and by the
// ablation harness tools/lab/asm_lab.hip).
#pragma once

#include "device_common.hpp"
#include "kernels.hpp"
#include "plan.hpp"

namespace femshell {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding
// global stores (s_waitcnt vmcnt(0)), which would expose the latency of the K stores of every output pass;
// nothing in k_assemble reads global memory written by the same launch, so the stores may stay in flight.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// =====================================================================================
// Assembly: replaces the element loop of assemble_elasticity (fem-shell.cpp:1197-1232).
// One lane owns one 6x6 block slot of K (slot k of node n of the slice) and sums the
// contributions K_e(ia,ib) of the elements in the slot's gather list, in element order.
// Slot 0 is the diagonal block (about six elements), the other slots are edge blocks
// (two elements), so putting the diagonal slots of a slice into one wave keeps the other
// waves divergence-free.  Every block is written exactly once: no atomics, no zero-fill
// pass, bitwise reproducible.
// Dirichlet dofs follow libMesh's constrain_element_matrix_and_vector (fem-shell.cpp:1227):
// rows and columns of fixed dofs are zero, the diagonal entry is the number of elements
// touching the node.
// =====================================================================================
// kAblate (profiling builds only, 0 in the product): 1 = skip global stores, 2 = every lane reads
// record 0, 4 = skip the block math, 8 = skip the record math, 32 = s_memtime stamps at the phase boundaries
// kHasQuads = false compiles the QUAD4 code out (meshes of triangles only: every BASELINE config)
template <int kWavesPerSimd, int kAblate = 0, bool kHasQuads = false>
__global__ __launch_bounds__(256, kWavesPerSimd) void k_assemble(DeviceMatrix m, MatConst mc)
{
    // LDS: [element records | partial-sum staging] (assemble_lds_layout)
    extern __shared__ double lds[];
    double *lds_rec = lds + m.lds_rec_off;
    double *lds_stage = lds + m.lds_stage_off;
    constexpr int kRec = kHasQuads ? kRecDoublesQuad : kRecDoubles; // doubles per element record in LDS
    const int tid = threadIdx.x;
    // element records are built by the lanes counted from the END of the workgroup (element i of the slice by lane
    // 255 - i): the diagonal slots' work items, three contributions each, sit in the first wave, which would
    // otherwise also carry 64 of the 130 records of a structured slice; the last waves have two contributions per
    // lane and start the next slice's records while the first wave still finishes its blocks
    constexpr int kThreads = 256; // the launch's workgroup size (blockDim.x would be a scalar load per use)
    const int etid = kThreads - 1 - tid;

    // profiling build (kAblate & 32): s_memtime stamps at the phase boundaries, summed per wave and written to
    // m.stamps[wave][8]; never compiled into the product kernel
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot) {
        if (kAblate & 32) {
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            if (slot >= 0) tacc[slot] += tnow - tprev;
            tprev = tnow;
        }
    };
    SliceWalk w(m.n_slices);
    if (!w.valid()) return;
    unsigned long long rt0 = 0, ct0 = 0;
    if (kAblate & 32) {
        rt0 = __builtin_amdgcn_s_memrealtime();
        ct0 = __builtin_amdgcn_s_memtime();
    }
    stamp(-1);
    // software pipeline over slices, two deep: while slice s computes its blocks, the coordinates of this
    // lane's element of slice s+1 (node ids fetched one slice earlier) and the node ids of slice s+2 are in
    // flight, so phase A starts with its operands in registers
    auto fetch_coords = [&](const int4 &c, double X[9]) {
        const double *pa = m.xyz + 3 * (int64_t)c.x, *pb = m.xyz + 3 * (int64_t)c.y, *pc = m.xyz + 3 * (int64_t)c.z;
        X[0] = pa[0]; X[1] = pa[1]; X[2] = pa[2];
        X[3] = pb[0]; X[4] = pb[1]; X[5] = pb[2];
        X[6] = pc[0]; X[7] = pc[1]; X[8] = pc[2];
    };
    // The bookkeeping of a slice (element range, item range, slot base, width) is one 32-byte descriptor, fetched
    // three slices ahead.  The compiler reads it with a vector load (the kernel stores to memory it cannot tell
    // apart) and would move it to scalar registers at once -- a full memory round trip in every wave and slice, behind
    // the K stores of the previous slice because vmcnt retires in order.  So the words of the newest descriptor stay in
    // vector registers (desc_a, desc_b) and are moved to scalar registers at the top of the next slice, where the
    // wave waits for its prefetched coordinates anyway.
    struct Desc {
        int e0, ne, i0, ni;
        int64_t base;
        int W;
    };
    auto load_desc = [&](int s_) {
        Desc d = {0, 0, 0, 0, 0, 0};
        if (s_ < w.last) {
            const int4 a = m.slice_desc[2 * s_], b = m.slice_desc[2 * s_ + 1];
            d.e0 = a.x; d.ne = a.y; d.i0 = a.z; d.ni = a.w;
            d.base = (int64_t)(((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x);
            d.W = b.z;
        }
        return d;
    };
    int4 desc_a = make_int4(0, 0, 0, 0), desc_b = make_int4(0, 0, 0, 0);
    auto fetch_desc = [&](int s_) {
        desc_a = make_int4(0, 0, 0, 0);
        desc_b = make_int4(0, 0, 0, 0);
        if (s_ < w.last) {
            desc_a = m.slice_desc[2 * s_];
            desc_b = m.slice_desc[2 * s_ + 1];
        }
    };
    auto decode_desc = [&]() {
        Desc d;
        d.e0 = __builtin_amdgcn_readfirstlane(desc_a.x);
        d.ne = __builtin_amdgcn_readfirstlane(desc_a.y);
        d.i0 = __builtin_amdgcn_readfirstlane(desc_a.z);
        d.ni = __builtin_amdgcn_readfirstlane(desc_a.w);
        d.base = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(desc_b.y) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane(desc_b.x));
        d.W = __builtin_amdgcn_readfirstlane(desc_b.z);
        return d;
    };
    Desc d0 = load_desc(w.s), d1 = load_desc(w.s + w.step), d2 = d1;
    fetch_desc(w.s + 2 * w.step);
    int e0 = d0.e0, ne = d0.ne;
    int4 nd = make_int4(0, 0, 0, -1);
    if (etid < ne) nd = m.slice_elem_nodes[e0 + etid];
    double Xcur[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!kHasQuads && etid < ne) fetch_coords(nd, Xcur);
    int4 nd_n = make_int4(0, 0, 0, -1); // slice s+1
    if (etid < d1.ne) nd_n = m.slice_elem_nodes[d1.e0 + etid];
    // right-hand side of the slice (contribRHS, fem-shell.cpp:1118-1153: a masked copy of the nodal loads): loaded
    // one slice ahead in front of the coordinate prefetch, stored where phase A waits for the coordinates anyway
    double rhs_pre = 0.0;
    uint32_t rhs_mask = 0u;
    auto fetch_rhs = [&](int s_) {
        if (m.rhs_F != nullptr && s_ < w.last && tid < kSliceRows) {
            rhs_pre = m.rhs_loads[(int64_t)s_ * kSliceRows + tid];
            rhs_mask = m.dmask[s_ * kSliceNodes + tid / 6];
        }
    };
    fetch_rhs(w.s);
    uint4 item_pre = make_uint4(0, 0, 0, 0);
    uint32_t flags_pre = 0u;
    if (tid < d0.ni) {
        item_pre = m.items[d0.i0 + tid];
        flags_pre = m.item_flags[d0.i0 + tid];
    }

    for (; w.valid(); w.next()) {
        const int s = w.s;
        const int64_t base = d0.base;
        const int i0 = d0.i0, ni = d0.ni;
        uint4 item = item_pre; // fetched during the previous slice's block math
        uint32_t flags = flags_pre;
        // The wait for the item words belongs here, where the wave waits for its prefetched coordinates anyway.  The
        // vector-memory counter retires in order and the compiler can only count operations issued on every path: at the
        // first use of the item (phase B) its wait was vmcnt(1), i.e. for the prefetches the wave had issued a moment
        // before -- a memory round trip per slice in front of the block math (0.79 -> 0.71 ms).
        asm volatile("" : "+v"(item.x), "+v"(item.y), "+v"(item.z), "+v"(item.w), "+v"(flags));
        d2 = decode_desc(); // slice s+2, fetched during the previous slice

        if (m.rhs_F != nullptr && tid < kSliceRows) {
            const bool fixed = (rhs_mask >> (tid % 6)) & 1u;
            m.rhs_F[(int64_t)s * kSliceRows + tid] = (fixed || s * kSliceNodes + tid / 6 >= m.n_own) ? 0.0 : rhs_pre;
        }
        stamp(5); // top of the slice: wait for the prefetched operands, right-hand side rows, descriptor decode
        // ---- phase A: one record per element touching the slice
        for (int i = etid; i < ne; i += kThreads) {
            const int4 c = (i == etid) ? nd : m.slice_elem_nodes[e0 + i];
            double rec[kRec];
            bool ok = false;
            if (!kHasQuads || c.w < 0) {
                double X[9];
                if (!kHasQuads && i == etid) {
#pragma unroll
                    for (int q = 0; q < 9; q++) X[q] = Xcur[q];
                } else {
                    fetch_coords(c, X);
                }
                if (kAblate & 8) {
#pragma unroll
                    for (int q = 0; q < kRec; q++) rec[q] = 1.0 + 0.01 * q + X[q % 9] * 1e-9;
                    ok = true;
                } else {
                    ok = tri3_record(X, mc, rec);
                    if (kHasQuads) {
#pragma unroll
                        for (int q = kRecDoubles; q < kRec; q++) rec[q] = 0.0;
                    }
                }
            } else {
                double X[12];
                const int nid[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double *pt = m.xyz + 3 * (int64_t)nid[q];
                    X[3 * q + 0] = pt[0];
                    X[3 * q + 1] = pt[1];
                    X[3 * q + 2] = pt[2];
                }
                ok = quad4_record(X, mc, rec);
            }
            if (!ok) report_status(m.status, e0 + i + 1);
            double2 *dst = reinterpret_cast<double2 *>(lds_rec + (size_t)i * kRec);
#pragma unroll
            for (int q = 0; q < kRec / 2; q++) dst[q] = make_double2(rec[2 * q], rec[2 * q + 1]);
        }
        stamp(0); // phase A (coordinate gather + record math + LDS writes)
        lds_barrier();
        stamp(1); // barrier after phase A
        // prefetches that overlap the block math below: coordinates and first items of slice s+1, element node
        // ids of slice s+2, descriptor of slice s+3
        uint4 item_next = make_uint4(0, 0, 0, 0);
        uint32_t flags_next = 0u;
        fetch_desc(s + 3 * w.step);
        fetch_rhs(s + w.step);
        e0 = d1.e0;
        ne = d1.ne;
        nd = nd_n;
        if (!kHasQuads && etid < ne) fetch_coords(nd, Xcur);
        if (tid < d1.ni) {
            item_next = m.items[d1.i0 + tid];
            flags_next = m.item_flags[d1.i0 + tid];
        }
        if (etid < d2.ne) nd_n = m.slice_elem_nodes[d2.e0 + etid];

        stamp(6); // issue of the prefetches
        // ---- phase B: one lane per work item (at most kItemPairs element contributions), in rounds of 256
        //      items; the lane that owns a block slot stores its finished block straight to K: in the layout of
        //      plan.hpp the lanes of a slot (consecutive nodes) write consecutive 16-byte words with every store
        double2 *out = reinterpret_cast<double2 *>(m.vals + base * 36);
        const bool multi = ni > kThreads; // several rounds
        for (int r0 = 0; r0 < ni; r0 += kThreads) {
            const int it = r0 + tid;
            const bool live = it < ni;
            if (r0 > 0) {
                item = make_uint4(0, 0, 0, 0);
                flags = 0u;
                if (live) {
                    item = m.items[i0 + it];
                    flags = m.item_flags[i0 + it];
                }
                // wait for the two loads here, inside the branch: left to the compiler the wait sits behind the join, where
                // the first round would also wait for the prefetches it has just issued
                asm volatile("" : "+v"(item.x), "+v"(item.y), "+v"(item.z), "+v"(item.w), "+v"(flags));
            }
            const int slot_in_slice = (int)(item.x & 0xffffu), chunk = (int)((item.x >> 16) & 0xffu),
                      nchunks = (int)(item.x >> 24);
            const int cnt = (int)(item.z >> 16);
            // slot 0 of a node row is its diagonal block: all contributions are K_e(ia, ia), symmetric, and the lane
            // keeps the upper triangle only (tri3_diag_add_rec; blk[0..20] in sym6 order).  Items are ordered by work, so
            // the diagonal items of a slice fill whole waves.
            const bool sym_item = !kHasQuads && !(kAblate & 4) && slot_in_slice < kSliceNodes;
            double blk[36];
#pragma unroll
            for (int i = 0; i < 36; i++) blk[i] = 0.0;
            if (sym_item) {
                for (int q = 0; q < cnt; q++) {
                    const uint32_t pr = (q == 0) ? (item.y & 0xffffu) : (q == 1 ? (item.y >> 16) : (item.z & 0xffffu));
                    const double *rec = lds_rec + ((kAblate & 2) ? 0 : (size_t)(pr >> 4) * kRec);
                    tri3_diag_add_rec(rec, (int)(pr & 3u), mc, blk);
                }
            } else {
                for (int q = 0; q < cnt; q++) {
                    const uint32_t pr = (q == 0) ? (item.y & 0xffffu) : (q == 1 ? (item.y >> 16) : (item.z & 0xffffu));
                    const double *rec = lds_rec + ((kAblate & 2) ? 0 : (size_t)(pr >> 4) * kRec);
                    if (kAblate & 4) {
#pragma unroll
                        for (int i = 0; i < 26; i++) blk[i] += rec[i];
                    } else {
                        block_add_rec<kHasQuads>(rec, (int)((pr >> 2) & 3u), (int)(pr & 3u), mc, blk);
                    }
                }
            }
            const bool owner = live && chunk == 0 && nchunks > 0; // nchunks == 0: padding item
            // Dirichlet masks of the slot's row and column node, contributions in the slot, diagonal slot or
            // not: the item's constraint word (no global load in this phase: vmcnt retires in order, so waiting
            // for one would also wait for the previous slice's K stores)
            const uint32_t mrow = flags & 63u, mcol = (flags >> 6) & 63u;
            const int valence = (int)((flags >> 12) & 255u);
            const bool diag_slot = (flags >> 20) & 1u;
            // constraints (libMesh constrain_element_matrix_and_vector semantics) and the 18 stores of a finished block
            typedef double v2d __attribute__((ext_vector_type(2)));
            // the same for a diagonal block held as its upper triangle: row and column masks coincide
            auto finish_sym_block = [&]() {
                if (mrow) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = i; j < 6; j++)
                            if (((mrow >> i) & 1u) | ((mrow >> j) & 1u)) blk[sym6(i, j)] = (i == j) ? (double)valence : 0.0;
                }
                if (!(kAblate & 1)) {
                    v2d *dst = reinterpret_cast<v2d *>(out) + (slot_in_slice & 31);
#pragma unroll
                    for (int jp = 0; jp < 3; jp++)
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            const int j0 = 2 * jp, j1 = 2 * jp + 1;
                            v2d vv;
                            vv.x = blk[i <= j0 ? sym6(i, j0) : sym6(j0, i)];
                            vv.y = blk[i <= j1 ? sym6(i, j1) : sym6(j1, i)];
                            if (m.diag_upper && j1 < i) continue; // (a word below the diagonal: nobody reads it, kernels.hpp)
                            if (kAblate & 64) dst[(jp * 6 + i) * kSliceNodes] = vv;
                            else __builtin_nontemporal_store(vv, dst + (jp * 6 + i) * kSliceNodes);
                        }
                } else if (blk[0] == 1.2345e300) {
                    out[0] = make_double2(blk[1], blk[2]);
                }
            };
            auto finish_block = [&]() {
                if (sym_item) {
                    finish_sym_block();
                    return;
                }
                if (mrow | mcol) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++)
                            if (((mrow >> i) & 1u) | ((mcol >> j) & 1u)) blk[6 * i + j] = 0.0;
                    if (diag_slot) {
#pragma unroll
                        for (int i = 0; i < 6; i++)
                            if ((mrow >> i) & 1u) blk[7 * i] = (double)valence;
                    }
                }
                if (!(kAblate & 1)) {
                    v2d *dst = reinterpret_cast<v2d *>(out) + (size_t)(slot_in_slice >> 5) * 3 * kSliceRows + (slot_in_slice & 31);
#pragma unroll
                    for (int jp = 0; jp < 3; jp++)
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            v2d vv; vv.x = blk[6 * i + 2 * jp]; vv.y = blk[6 * i + 2 * jp + 1];
                            if (m.diag_upper && diag_slot && 2 * jp + 1 < i) continue; // (below the diagonal of a diagonal block)
                            if (kAblate & 64) dst[(jp * 6 + i) * kSliceNodes] = vv; // lab: plain instead of non-temporal stores
                            else __builtin_nontemporal_store(vv, dst + (jp * 6 + i) * kSliceNodes);
                        }
                } else if (blk[0] == 1.2345e300) {
                    out[0] = make_double2(blk[1], blk[2]);
                }
            };
            // slots with a single work item are complete now: their lanes store before the barrier, while the
            // lanes of the split (diagonal) slots are still in their last contribution
            if (owner && nchunks == 1) finish_block();
            stamp(2); // item decode + block math (+ stores of the single-item slots)
            if (live && chunk > 0) {
                double2 *st = reinterpret_cast<double2 *>(lds_stage + (size_t)item.w * 36);
#pragma unroll
                for (int i = 0; i < 11; i++) st[i] = make_double2(blk[2 * i], blk[2 * i + 1]);
                if (!sym_item) {
#pragma unroll
                    for (int i = 11; i < 18; i++) st[i] = make_double2(blk[2 * i], blk[2 * i + 1]);
                }
            }
            lds_barrier();
            stamp(3); // staging write + barrier
            if (owner && nchunks > 1) {
                // chunks > 0 sort after chunk 0, so with several rounds they may not have run yet:
                // plan.cpp keeps all chunks of a slot in one round when a slice has several rounds
                for (int c = 1; c < nchunks; c++) {
                    const double2 *st = reinterpret_cast<const double2 *>(lds_stage + (size_t)(item.w + c - 1) * 36);
#pragma unroll
                    for (int i = 0; i < 11; i++) {
                        const double2 v = st[i];
                        blk[2 * i] += v.x;
                        blk[2 * i + 1] += v.y;
                    }
                    if (!sym_item) {
#pragma unroll
                        for (int i = 11; i < 18; i++) {
                            const double2 v = st[i];
                            blk[2 * i] += v.x;
                            blk[2 * i + 1] += v.y;
                        }
                    }
                }
                finish_block();
            }
            stamp(4); // partial-sum reduction + constraints + K stores
            if (multi) lds_barrier(); // the next round reuses the partial-sum rows
        }
        item_pre = item_next;
        flags_pre = flags_next;
        d0 = d1;
        d1 = d2;
    }
    if (kAblate & 32) {
        if ((tid & 63) == 0) {
            unsigned long long *dst = m.stamps + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 8;
            for (int q = 0; q < 8; q++) dst[q] = tacc[q];
            if (blockIdx.x == 0 && tid == 0) { // clock estimate: shader cycles per 100 MHz tick
                m.stamps[(size_t)gridDim.x * 32] = __builtin_amdgcn_s_memtime() - ct0;
                m.stamps[(size_t)gridDim.x * 32 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
            }
        }
    }
}


// =====================================================================================
// The same assembly as a producer / consumer pipeline (plans with Plan::pipe: slices whose two record buffers fit twice
// into a CU's LDS and whose items fill the consumer waves evenly -- structured meshes; plan.cpp decides).
//
// k_assemble runs a slice in two phases -- records, barrier, blocks -- and its four waves do different amounts of work
// (one wave: diagonal items; two waves: records and off-diagonal items; one wave: records only): the longest wave and the
// serial phases set the time of a slice, and the two SIMDs that hold the long waves of both resident workgroups are busy
// while the others idle.  Here one wave of the workgroup -- the producer -- builds the records of the NEXT slice into a
// second LDS buffer while the other three -- the consumers -- compute and store the blocks of the current one: one
// barrier per slice, four waves of about equal work (records, as one stream of 64-lane passes across the workgroup's
// slices: 2.03 per structured slice | diagonal items | off-diagonal items x 2), no staging of partial sums in LDS (the
// chunks of a block slot sit in neighbouring lanes of one wave, plan.cpp pack_items_pipe, and meet through lane shifts).
// Which wave does what follows the SIMD it runs on, and the second workgroup of a CU shifts the roles by two SIMDs, so
// that every SIMD carries one long and one short wave; the producer runs at raised priority.
// Records of triangles are RecLean (34 doubles), items and flags as in k_assemble, in rounds of 192 lanes.
// kAblate (lab): 8 = no record math, 16 = roles by wave index and the same in every workgroup, 32 = s_memtime stamps,
// 64 = no priority for the producer, 128 = the second workgroup swaps neighbouring roles instead of shifting them by two SIMDs;
// 0 in the product.
// =====================================================================================
// value of the next lane of the wave (lane 63: unspecified): DPP wave_shl:1 on both halves of the double
__device__ __forceinline__ double lane_below(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi2, lo2);
}

constexpr int kPipeConsumers = 192;  // lanes that own work items (three waves)
constexpr int kPipeFlagShift = 9;    // item.w of the pipelined layout: wave word (9 bits) | constraint word << 9
constexpr int kPipeRecPasses = 3;    // record passes the producer wave runs at most per slice: 192 records, more than a slice
                                     // of a plan with Plan::pipe has (kPipeMaxSliceElems)

// kHasQuads: meshes with quadrilaterals -- records in the full 66-double layout (tri3_record / quad4_record as in k_assemble),
// every block through block_add_rec, diagonal blocks as full 6x6 blocks
template <int kAblate = 0, bool kHasQuads = false>
__global__ __launch_bounds__(256, 2) void k_assemble_pipe(DeviceMatrix m, MatConst mc)
{
    extern __shared__ double lds[];
    __shared__ int simd_of_wave[4];
    constexpr int kRec = kHasQuads ? kRecDoublesQuad : RecLean::doubles;
    constexpr int kCoords = kHasQuads ? 12 : 9; // coordinates of an element
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- roles: by the SIMD a wave runs on when the four waves sit on four SIMDs (they do; the order varies from
    //      workgroup to workgroup), else by wave index.  The workgroups b and b + G/2 of a 2-per-CU grid share a CU.
    int vwave;
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int simd = (int)((hw >> 4) & 3u);
        if (lane == 0) simd_of_wave[wave] = simd;
        __syncthreads();
        const int seen = (1 << simd_of_wave[0]) | (1 << simd_of_wave[1]) | (1 << simd_of_wave[2]) | (1 << simd_of_wave[3]);
        const int place = (seen == 15 && !(kAblate & 16)) ? simd : wave;
        const bool second = blockIdx.x >= (gridDim.x >> 1) && !(kAblate & 16);
        // roles of the first workgroup on SIMDs 0..3: diagonal | off-diagonal | off-diagonal (last wave) | producer; the
        // second one is shifted by two SIMDs: diagonal opposite the last off-diagonal wave, producer opposite the full
        // off-diagonal wave (measured against swapping neighbours, producer opposite the last wave: 0.56 against 0.62 ms)
        vwave = __builtin_amdgcn_readfirstlane((kAblate & 128) ? (second ? (place ^ 1) : place) : ((place + (second ? 2 : 0)) & 3));
    }
    SliceWalk w(m.n_slices);
    if (!w.valid()) return;
    const int n_iter = (w.last - w.s + w.step - 1) / w.step; // slices of this workgroup
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot) {
        if (kAblate & 32) {
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            if (slot >= 0) tacc[slot] += tnow - tprev;
            tprev = tnow;
        }
    };
    stamp(-1);
    // first int4 of a slice's descriptor {elem begin, elem count, item begin, item count}, second {slot base lo, hi, width}
    auto desc_a_of = [&](int s_) { return (s_ < w.last) ? m.slice_desc[2 * s_] : make_int4(0, 0, 0, 0); };
    auto desc_b_of = [&](int s_) { return (s_ < w.last) ? m.slice_desc[2 * s_ + 1] : make_int4(0, 0, 0, 0); };

    if (vwave == 3) {
        // ================= producer: records of slice j+1 while the consumers work on slice j =================
        // The element lists of the workgroup's slices are one stream of records, built in passes of 64 lanes: the lanes of a
        // slice's last pass that its own list does not fill (62 of 64 when a structured slice of 130 elements starts a
        // pass) build the first records of the NEXT slice.  Those cannot go to LDS yet -- their buffer is the one the
        // consumers are reading -- so the lane holds them in registers until the barrier and writes them first thing in
        // the next iteration.  2.03 passes per structured slice instead of 3.
        // Operands in flight: coordinates of the passes of the next iteration (fetched while this one computes), node ids
        // one iteration further, element ranges (first word of the descriptors) of four slices.
        const int4 *enodes = m.slice_elem_nodes;
        const int n_entries = m.slice_elem_ptr_last; // entries of slice_elem_nodes (clamp for idle lanes)
        struct Range {
            int e0, ne;
        };
        // passes of an iteration that starts at record r0 of a slice with range a
        auto passes_of = [&](int r0, const Range &a) { return (a.ne > r0) ? (a.ne - r0 + 63) >> 6 : 0; };
        // element of a lane in such an iteration, b being the range of the following slice
        // (slots 0 and 1 are the full passes, if any; slot 2 is the last pass, the one that may reach into the next slice)
        auto stream_elem = [&](int r0, const Range &a, const Range &b, int slot) {
            // (a full-pass slot the iteration does not use reads what slot 0 reads: a cache hit instead of HBM traffic)
            const int last = max(passes_of(r0, a) - 1, 0);
            const int pi = (slot < kPipeRecPasses - 1) ? (slot < last ? slot : 0) : last;
            const int li = r0 + 64 * pi + lane, li2 = li - a.ne;
            int e = (li < a.ne) ? a.e0 + li : ((li2 < b.ne) ? b.e0 + li2 : a.e0 + max(a.ne - 1, 0));
            return min(max(e, 0), n_entries - 1);
        };
        // records of the following slice the last pass builds
        auto carry_of = [&](int r0, const Range &a, const Range &b) {
            return min(max(r0 + 64 * passes_of(r0, a) - a.ne, 0), b.ne);
        };
        auto fetch_coords = [&](const int4 &c, double X[kCoords]) {
            const double *pa = m.xyz + 3 * (int64_t)c.x, *pb = m.xyz + 3 * (int64_t)c.y, *pc = m.xyz + 3 * (int64_t)c.z;
            X[0] = pa[0]; X[1] = pa[1]; X[2] = pa[2];
            X[3] = pb[0]; X[4] = pb[1]; X[5] = pb[2];
            X[6] = pc[0]; X[7] = pc[1]; X[8] = pc[2];
            if (kHasQuads) { // (a triangle's fourth node is -1: its first node again, a valid address)
                const double *pd = m.xyz + 3 * (int64_t)(c.w >= 0 ? c.w : c.x);
                X[kCoords - 3] = pd[0]; X[kCoords - 2] = pd[1]; X[kCoords - 1] = pd[2];
            }
        };
        auto range_of = [&](const int4 &a) {
            Range r;
            r.e0 = __builtin_amdgcn_readfirstlane(a.x);
            r.ne = __builtin_amdgcn_readfirstlane(a.y);
            return r;
        };
        // the producer is the longest wave of the four: it goes first whenever the SIMD has a choice
        if (!(kAblate & 64)) __builtin_amdgcn_s_setprio(3); // (1 does as well; without: 0.61 against 0.55 ms)
        int s_build = w.s; // slice whose records are built next
        Range ra = range_of(desc_a_of(s_build)), rb = range_of(desc_a_of(s_build + w.step)),
              rc = range_of(desc_a_of(s_build + 2 * w.step));
        int4 dvec = desc_a_of(s_build + 3 * w.step); // becomes rd in the first iteration
        int r0 = 0;                                   // records of slice s_build that are built already
        int r0n = carry_of(r0, ra, rb);
        double X[kPipeRecPasses][kCoords];
        int4 ndn[kPipeRecPasses];
        bool is_quad[kPipeRecPasses]; // the element whose coordinates X[p] holds
#pragma unroll
        for (int p = 0; p < kPipeRecPasses; p++) {
            const int4 c0 = enodes[stream_elem(r0, ra, rb, p)];
            fetch_coords(c0, X[p]);
            is_quad[p] = kHasQuads && c0.w >= 0;
            ndn[p] = enodes[stream_elem(r0n, rb, rc, p)];
        }
        double held[kRec]; // a record of the next slice, lean layout
        int held_at = -1;  // its index there, -1: none
#pragma unroll
        for (int q = 0; q < kRec; q++) held[q] = 0.0;
        auto write_lean = [&](double *buf, int i, const double rec[kRec]) {
            double2 *dst = reinterpret_cast<double2 *>(buf + pipe_rec_offset(i, kRec));
#pragma unroll
            for (int q = 0; q < kRec / 2; q++) dst[q] = make_double2(rec[2 * q], rec[2 * q + 1]);
        };
        auto record_of = [&](const double Xp[kCoords], bool quad, double lean[kRec]) __attribute__((always_inline)) {
            if (kHasQuads) { // the record as k_assemble keeps it
                bool okq;
                if (quad) {
                    okq = quad4_record(Xp, mc, lean);
                } else {
                    okq = tri3_record(Xp, mc, lean);
#pragma unroll
                    for (int q = kRecDoubles; q < kRec; q++) lean[q] = 0.0;
                }
                return okq;
            }
            double rec[kRecDoubles];
            bool ok;
            if (kAblate & 8) {
#pragma unroll
                for (int q = 0; q < kRecDoubles; q++) rec[q] = 1.0 + 0.01 * q + Xp[q % 9] * 1e-9;
                ok = true;
            } else {
                ok = tri3_record(Xp, mc, rec);
            }
#pragma unroll
            for (int q = 0; q < kRec; q++) lean[q] = rec[lean_from_full(q)];
            return ok;
        };
        auto produce = [&](double *buf) {
            if (held_at >= 0) write_lean(buf, held_at, held);
            held_at = -1;
            const int np = passes_of(r0, ra);
#pragma unroll
            for (int p = 0; p < kPipeRecPasses - 1; p++) { // full passes
                if (p < np - 1) {
                    double lean[kRec];
                    const bool ok = record_of(X[p], is_quad[p], lean);
                    const int li = r0 + 64 * p + lane;
                    if (!ok) report_status(m.status, ra.e0 + li + 1);
                    write_lean(buf, li, lean);
                }
                // (for every slot, needed by the next iteration or not: loads under a condition cost the wave's memory
                // counter its precision -- 0.62 against 0.55 ms with the second slot's nine loads skipped when unused)
                fetch_coords(ndn[p], X[p]); // the next iteration's element of this lane and slot
                is_quad[p] = kHasQuads && ndn[p].w >= 0;
            }
            if (np > 0) { // last pass: the slice's last records, then the first of the next slice
                const bool ok = record_of(X[kPipeRecPasses - 1], is_quad[kPipeRecPasses - 1], held);
                const int li = r0 + 64 * (np - 1) + lane, li2 = li - ra.ne;
                if (li < ra.ne) {
                    if (!ok) report_status(m.status, ra.e0 + li + 1);
                    write_lean(buf, li, held);
                } else if (li2 < rb.ne) {
                    if (!ok) report_status(m.status, rb.e0 + li2 + 1);
                    held_at = li2;
                }
            } else { // (every path defines the held record anew: its registers are free during the full passes)
#pragma unroll
                for (int q = 0; q < kRec; q++) held[q] = 0.0;
            }
            fetch_coords(ndn[kPipeRecPasses - 1], X[kPipeRecPasses - 1]);
            is_quad[kPipeRecPasses - 1] = kHasQuads && ndn[kPipeRecPasses - 1].w >= 0;
            // roll the window
            const Range rd = range_of(dvec);
            const int r0nn = carry_of(r0n, rb, rc);
            r0 = r0n;
            r0n = r0nn;
            ra = rb;
            rb = rc;
            rc = rd;
            s_build += w.step;
#pragma unroll
            for (int p = 0; p < kPipeRecPasses; p++) ndn[p] = enodes[stream_elem(r0n, rb, rc, p)];
            dvec = desc_a_of(s_build + 3 * w.step);
        };
        produce(lds);
        stamp(0);
        lds_barrier();
        stamp(1);
        for (int j = 0; j < n_iter; j++) {
            if (j + 1 < n_iter) produce(lds + (size_t)((j + 1) & 1) * m.lds_rec_off);
            stamp(0);
            lds_barrier();
            stamp(1);
        }
    } else {
        // ================= consumers: block slots of slice j from record buffer j & 1 =================
        const int vtid = vwave * 64 + lane; // 0..191
        struct Desc {
            int e0, ne, i0, ni;
            int64_t base;
        };
        auto decode = [&](const int4 &a, const int4 &b) {
            Desc d;
            d.e0 = __builtin_amdgcn_readfirstlane(a.x);
            d.ne = __builtin_amdgcn_readfirstlane(a.y);
            d.i0 = __builtin_amdgcn_readfirstlane(a.z);
            d.ni = __builtin_amdgcn_readfirstlane(a.w);
            d.base = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(b.y) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane(b.x));
            return d;
        };
        Desc d0 = decode(desc_a_of(w.s), desc_b_of(w.s)), d1 = decode(desc_a_of(w.s + w.step), desc_b_of(w.s + w.step));
        int4 dva = desc_a_of(w.s + 2 * w.step), dvb = desc_b_of(w.s + 2 * w.step);
        double rhs_pre = 0.0;
        uint32_t rhs_mask = 0u;
        auto fetch_rhs = [&](int s_) {
            if (m.rhs_F != nullptr && s_ < w.last) {
                rhs_pre = m.rhs_loads[(int64_t)s_ * kSliceRows + vtid];
                rhs_mask = m.dmask[s_ * kSliceNodes + vtid / 6];
            }
        };
        fetch_rhs(w.s);
        uint4 item_pre = make_uint4(0xffffu, 0, 0, 0);
        if (vtid < d0.ni) item_pre = m.items[d0.i0 + vtid];
        lds_barrier(); // records of the first slice
        stamp(1);
        for (int j = 0; j < n_iter; j++, w.next()) {
            const int s = w.s;
            const double *lds_rec = lds + (size_t)(j & 1) * m.lds_rec_off;
            const int64_t base = d0.base;
            const int i0 = d0.i0, ni = d0.ni;
            uint4 item = item_pre;
            // the wait for the prefetched words belongs here, in front of this slice's prefetches (see k_assemble)
            asm volatile("" : "+v"(item.x), "+v"(item.y), "+v"(item.z), "+v"(item.w));
            const Desc d2 = decode(dva, dvb);
            if (m.rhs_F != nullptr) {
                const bool fixed = (rhs_mask >> (vtid % 6)) & 1u;
                m.rhs_F[(int64_t)s * kSliceRows + vtid] = (fixed || s * kSliceNodes + vtid / 6 >= m.n_own) ? 0.0 : rhs_pre;
            }
            stamp(5);
            // prefetches that overlap the block math: descriptor three slices ahead, right-hand side and items of the next
            dva = desc_a_of(s + 3 * w.step);
            dvb = desc_b_of(s + 3 * w.step);
            fetch_rhs(s + w.step);
            uint4 item_next = make_uint4(0xffffu, 0, 0, 0);
            if (vtid < d1.ni) item_next = m.items[d1.i0 + vtid];
            stamp(6);
            double2 *out = reinterpret_cast<double2 *>(m.vals + base * 36);
            for (int r0 = 0; r0 < ni; r0 += kPipeConsumers) {
                const int it = r0 + vtid;
                const bool live = it < ni;
                if (r0 > 0) {
                    item = make_uint4(0xffffu, 0, 0, 0);
                    if (live) item = m.items[i0 + it];
                    asm volatile("" : "+v"(item.x), "+v"(item.y), "+v"(item.z), "+v"(item.w));
                }
                const int slot_in_slice = (int)(item.x & 0xffffu), chunk = (int)((item.x >> 16) & 0xffu),
                          nchunks = (int)(item.x >> 24);
                const int cnt = (int)((item.z >> 16) & 0xffu);
                const bool general = (item.z >> 31) != 0u; // a diagonal slot outside the first wave: the general routine
                // what the plan says about this wave of the round: most chunks of a slot, and whether every item is diagonal
                const int wave_chunks = __builtin_amdgcn_readfirstlane((int)(item.w & 0xffu));
                const bool wave_sym = !kHasQuads && __builtin_amdgcn_readfirstlane((int)((item.w >> 8) & 1u)) != 0;
                const bool sym_item = !kHasQuads && slot_in_slice < kSliceNodes && !general;
                double blk[36];
#pragma unroll
                for (int i = 0; i < 36; i++) blk[i] = 0.0;
                if (sym_item) {
                    for (int q = 0; q < cnt; q++) {
                        const uint32_t pr = (q == 0) ? (item.y & 0xffffu) : (q == 1 ? (item.y >> 16) : (item.z & 0xffffu));
                        tri3_diag_add_rec<RecLean>(lds_rec + pipe_rec_offset((int)(pr >> 4), kRec), (int)(pr & 3u), mc, blk);
                    }
                } else {
                    for (int q = 0; q < cnt; q++) {
                        const uint32_t pr = (q == 0) ? (item.y & 0xffffu) : (q == 1 ? (item.y >> 16) : (item.z & 0xffffu));
                        const double *rec = lds_rec + pipe_rec_offset((int)(pr >> 4), kRec);
                        if (kHasQuads) block_add_rec<true>(rec, (int)((pr >> 2) & 3u), (int)(pr & 3u), mc, blk);
                        else tri3_block_add_rec<RecLean>(rec, (int)((pr >> 2) & 3u), (int)(pr & 3u), mc, blk);
                    }
                }
                stamp(2);
                const bool owner = live && chunk == 0 && nchunks > 0;
                // partial sums of the slot's other chunks: they sit in the next lanes of this wave, in chunk order
                // (wave_shl:1 moves every lane's value one lane down, a VALU move; c steps bring chunk c to its owner)
                // (the moves run in every lane -- a lane shift reads the lanes it reads whether they want the result or not --
                // and the owners' additions under one branch: per word two moves and an addition instead of seven instructions)
                if (wave_chunks == 2 && wave_sym) { // the common case spelled out: six-element nodes, two chunks of three
                    double t[21];
#pragma unroll
                    for (int i = 0; i < 21; i++) t[i] = lane_below(blk[i]);
                    if (owner && nchunks == 2) {
#pragma unroll
                        for (int i = 0; i < 21; i++) blk[i] += t[i];
                    }
                } else if (wave_chunks == 2) { // (a wave of off-diagonal slots cut in two: plan.cpp pack_items_pipe)
                    double t[36];
#pragma unroll
                    for (int i = 0; i < 36; i++) t[i] = lane_below(blk[i]);
                    if (owner && nchunks == 2) {
#pragma unroll
                        for (int i = 0; i < 36; i++) blk[i] += t[i];
                    }
                } else if (wave_chunks > 1 && wave_sym) { // more chunks: shift by shift, every owner adding what is its own
                    double t[21];
#pragma unroll
                    for (int i = 0; i < 21; i++) t[i] = blk[i];
                    for (int c = 1; c < wave_chunks; c++) {
#pragma unroll
                        for (int i = 0; i < 21; i++) t[i] = lane_below(t[i]);
                        if (owner && c < nchunks) {
#pragma unroll
                            for (int i = 0; i < 21; i++) blk[i] += t[i];
                        }
                    }
                } else if (wave_chunks > 1) {
                    double t[36];
#pragma unroll
                    for (int i = 0; i < 36; i++) t[i] = blk[i];
                    for (int c = 1; c < wave_chunks; c++) {
#pragma unroll
                        for (int i = 0; i < 36; i++) t[i] = lane_below(t[i]);
                        if (owner && c < nchunks) {
#pragma unroll
                            for (int i = 0; i < 36; i++) blk[i] += t[i];
                        }
                    }
                }
                stamp(3);
                if (owner) {
                    // the slot's constraint word rides in the item (k_item_flags: bits 9.. of w, beside the wave's word)
                    const uint32_t flags = item.w >> kPipeFlagShift;
                    const uint32_t mrow = flags & 63u, mcol = (flags >> 6) & 63u;
                    const int valence = (int)((flags >> 12) & 255u);
                    const bool diag_slot = (flags >> 20) & 1u;
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    if (sym_item) {
                        if (mrow) {
#pragma unroll
                            for (int i = 0; i < 6; i++)
#pragma unroll
                                for (int jj = i; jj < 6; jj++)
                                    if (((mrow >> i) & 1u) | ((mrow >> jj) & 1u)) blk[sym6(i, jj)] = (i == jj) ? (double)valence : 0.0;
                        }
                        v2d *dst = reinterpret_cast<v2d *>(out) + (slot_in_slice & 31);
#pragma unroll
                        for (int jp = 0; jp < 3; jp++)
#pragma unroll
                            for (int i = 0; i < 6; i++) {
                                const int j0 = 2 * jp, j1 = 2 * jp + 1;
                                v2d vv;
                                vv.x = blk[i <= j0 ? sym6(i, j0) : sym6(j0, i)];
                                vv.y = blk[i <= j1 ? sym6(i, j1) : sym6(j1, i)];
                                if (m.diag_upper && j1 < i) continue; // (a word below the diagonal: nobody reads it, kernels.hpp)
                                __builtin_nontemporal_store(vv, dst + (jp * 6 + i) * kSliceNodes);
                            }
                    } else {
                        if (mrow | mcol) {
#pragma unroll
                            for (int i = 0; i < 6; i++)
#pragma unroll
                                for (int jj = 0; jj < 6; jj++)
                                    if (((mrow >> i) & 1u) | ((mcol >> jj) & 1u)) blk[6 * i + jj] = 0.0;
                            if (diag_slot) {
#pragma unroll
                                for (int i = 0; i < 6; i++)
                                    if ((mrow >> i) & 1u) blk[7 * i] = (double)valence;
                            }
                        }
                        if (!kHasQuads && general) { // the two halves of a diagonal block as mirror images, as the
                                                     // upper-triangle routine leaves them (k_spmv_sym reads one half)
#pragma unroll
                            for (int i = 0; i < 6; i++)
#pragma unroll
                                for (int jj = i + 1; jj < 6; jj++) blk[6 * jj + i] = blk[6 * i + jj];
                        }
                        v2d *dst = reinterpret_cast<v2d *>(out) + (size_t)(slot_in_slice >> 5) * 3 * kSliceRows + (slot_in_slice & 31);
#pragma unroll
                        for (int jp = 0; jp < 3; jp++)
#pragma unroll
                            for (int i = 0; i < 6; i++) {
                                v2d vv;
                                vv.x = blk[6 * i + 2 * jp];
                                vv.y = blk[6 * i + 2 * jp + 1];
                                if (m.diag_upper && diag_slot && 2 * jp + 1 < i) continue;
                                __builtin_nontemporal_store(vv, dst + (jp * 6 + i) * kSliceNodes);
                            }
                    }
                }
                stamp(4);
            }
            item_pre = item_next;
            d0 = d1;
            d1 = d2;
            lds_barrier(); // this slice's records are free, the next slice's are complete
            stamp(1);
        }
    }
    if (kAblate & 32) {
        if (lane == 0) {
            unsigned long long *dst = m.stamps + ((size_t)blockIdx.x * 4 + vwave) * 8;
            for (int q = 0; q < 8; q++) dst[q] = tacc[q];
        }
    }
}

} // namespace femshell
