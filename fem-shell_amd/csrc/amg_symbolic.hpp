// amg_symbolic.hpp -- the integer half of a coarsening step on the device (amg_symbolic.hip).
//
// Until round 6 the host did all the integer work of amg_device_coarsen (amg_device_setup.cpp): the block graph as sorted lists,
// the patterns of P, A P, R = P^T and A_c = P^T A P as lists, their sliced ELL images, the maps from the blocks of A to the
// slots of P, the in-lists of A_c -- some 80 of the 165 ms a multigrid setup of a 4M-triangle mesh takes, loops over index arrays of
// 10-60 MB that the host touches for the first time.  Here the same patterns are built in HBM from the operator's own pattern
// (which the kernels of the cycle read anyway) and the aggregates: one lane per row collects the row as a sorted set in LDS, once
// to count and once to fill.  The host keeps the two sequential greedy passes of the aggregation and nothing else.  The
// results are the host's patterns slot for slot (tests/test_gpu_amg.py holds the two paths against each other), so the values
// the numeric kernels compute on them are the same bits.
#pragma once

#include "amg_device.hpp"
#include "amg_pattern.hpp"

namespace femshell {

// the pattern of a level operator as the symbolic kernels walk it: own slots, then the in-list (symmetric storage)
struct GraphView {
    int32_t n = 0, n_slices = 0;
    const int32_t *slice_width = nullptr;
    const int64_t *slice_base = nullptr;
    const int32_t *cols = nullptr;
    const uint8_t *count = nullptr; // real slots per row; nullptr: the plan's convention (padding slots repeat the row's index)
    int32_t symmetric = 0;
    const int32_t *in_width = nullptr;
    const int64_t *in_base = nullptr;
    const int32_t *in_slots = nullptr;
    const int32_t *in_rows = nullptr;
};

// what one coarsening step's numeric kernels need, all in HBM
struct DevSymbolic {
    DevBuf<int32_t> agg;          // n
    DevBuf<int32_t> gptr, order;  // nodes grouped by aggregate (ascending inside an aggregate)
    int32_t largest = 0;          // nodes of the largest aggregate
    DevPattern P, AP, R, Ac;
    EllPattern iP, iAP, iR, iAc;  // scalars only (n_rows, n_pad, n_slices, max_width, nnzb; slice_base holds just the total)
    int64_t totP = 0, totAP = 0, totR = 0, totAc = 0;
    DevBuf<uint8_t> pmap_own, pmap_in, rk;
    DevBuf<int64_t> rptr;
    DevBuf<int32_t> rrow;
    // in-lists of a symmetric coarse operator
    DevBuf<int32_t> in_width, in_slots, in_rows;
    DevBuf<int64_t> in_base;
    int64_t in_total = 0;
    int32_t max_in_width = 0;
    double useful_flops = 0.0, mfma_flops = 0.0; // work of the Galerkin product (AmgSetupStats)
};

// Builds everything above from the operator's pattern G (total_slots / in_total: sizes of its slot and in-list arrays) and the
// aggregates (host array, n entries, na aggregates).  Returns FEMSHELL_ERR_UNSUPPORTED (without an error text of its own
// mattering) when a row outgrows the lane sets -- the caller then takes the host path.
int amg_symbolic_device(femshell_ctx *c, hipStream_t st, const GraphView &G, int64_t total_slots, int64_t in_total, const std::vector<int32_t> &agg, int32_t na,
                        bool sym_coarse, DevSymbolic *out);
// host copies (for the inspection exports and for steps that continue on the host)
int download_pattern(femshell_ctx *c, const DevPattern &D, int64_t total, EllPattern *E, hipStream_t st);

} // namespace femshell
