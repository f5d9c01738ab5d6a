// amg_pattern.hpp -- host-side pattern helpers shared by the coarsening steps on the device (amg_device_setup.cpp: one rank;
// amg_dist.cpp: row-partitioned levels): rows as sorted column lists -> sliced block ELL patterns, their upload, and the way
// back from ELL values to a host BSR matrix.
#pragma once

#include <vector>

#include "amg_device.hpp"

namespace femshell {

// rows as sorted lists -> sliced ELL pattern (32 rows per slice, `count` real entries per row, padding columns 0;
// diag_first: the entry equal to the row index (+ diag_key) is moved to slot 0)
struct EllPattern {
    int32_t n_rows = 0, n_pad = 0, n_slices = 0, max_width = 0;
    std::vector<int32_t> slice_width;
    std::vector<int64_t> slice_base;
    RawVec<int32_t> cols;
    std::vector<uint8_t> count;
    int64_t nnzb = 0;
    int64_t total() const { return slice_base.empty() ? 0 : slice_base.back(); }
};

bool pack_pattern(int32_t n_rows, const int64_t *ptr, const int32_t *col, bool diag_first, EllPattern *out, int32_t diag_key = 0);

struct DevPattern {
    DevBuf<int32_t> slice_width, cols;
    DevBuf<int64_t> slice_base;
    DevBuf<uint8_t> count;
};

int upload_pattern(const EllPattern &E, DevPattern &D, double *vals, EllView *view, hipStream_t st);
// the pattern's device arrays become the operator's own (the cycle multiplies with them)
void adopt(AmgOperator &op, const EllPattern &E, DevPattern &D, DevBuf<double> &vals, int32_t n_cols_pad);
// host BSR (ascending columns) from a pattern and the ELL values brought back from the device
void ell_to_bsr(const EllPattern &E, const double *vals, int32_t n_cols, Bsr *out);
int download_vals(const DevBuf<double> &d, ValueArray *h, hipStream_t st);
// the block graph of a level operator as a pattern-only BSR with ascending columns (both directions of a symmetric one)
void graph_of_pattern(const HostEllPattern &H, Bsr *G);

} // namespace femshell
