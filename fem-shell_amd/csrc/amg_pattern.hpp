// amg_pattern.hpp -- host-side pattern helpers shared by the coarsening steps on the device (amg_device_setup.cpp: one rank;
// amg_dist.cpp: row-partitioned levels): rows as sorted column lists -> sliced block ELL patterns, their upload, and the way
// back from ELL values to a host BSR matrix.
#pragma once

#include <algorithm>
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

// Rows [0, n) as lists built by row(i, out) -- which appends the entries of row i to `out` (it arrives empty) -- on the host
// threads: ranges of rows into lists of their own (a thread's list is its own until it is handed over: neighbouring
// std::vector headers share cache lines), joined in row order.  Result: CSR arrays.
template <class F> void build_rows(int32_t n, F row, std::vector<int64_t> *ptr_out, RawVec<int32_t> *col_out)
{
    const int nchunks = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), ((int64_t)n + 1023) / 1024));
    std::vector<RawVec<int32_t>> parts((size_t)nchunks);
    std::vector<int32_t> cnt((size_t)std::max(n, 0), 0);
    parallel_chunks(nchunks, [&](int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++) {
            const int64_t a0 = (int64_t)n * t / nchunks, a1 = (int64_t)n * (t + 1) / nchunks;
            std::vector<int32_t> tmp;
            RawVec<int32_t> mine;
            mine.reserve((size_t)(a1 - a0) * 8);
            for (int64_t a = a0; a < a1; a++) {
                tmp.clear();
                row((int32_t)a, tmp);
                cnt[(size_t)a] = (int32_t)tmp.size();
                mine.insert(mine.end(), tmp.begin(), tmp.end());
            }
            parts[(size_t)t].swap(mine);
        }
    }, 1);
    std::vector<int64_t> &ptr = *ptr_out;
    ptr.assign((size_t)n + 1, 0);
    for (int32_t a = 0; a < n; a++) ptr[(size_t)a + 1] = ptr[(size_t)a] + cnt[(size_t)a];
    col_out->resize((size_t)ptr[(size_t)n]);
    parallel_chunks(nchunks, [&](int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; t++)
            std::copy(parts[(size_t)t].begin(), parts[(size_t)t].end(), col_out->begin() + ptr[(size_t)((int64_t)n * t / nchunks)]);
    }, 1);
}

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
// index arrays between host and HBM through the context's pinned staging buffer (context.hpp stage_host); both return when the
// data has arrived
int staged_download(femshell_ctx *c, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st);
int staged_upload(femshell_ctx *c, void *dst_dev, const void *src_host, size_t bytes, hipStream_t st, int region = 0);
// the block graph of a level operator as a pattern-only BSR with ascending columns (both directions of a symmetric one)
void graph_of_pattern(const HostEllPattern &H, Bsr *G);

} // namespace femshell
