// amg_symbolic.hip -- the patterns of a coarsening step built in HBM (amg_symbolic.hpp says why).
//
// Integer work, HBM- and latency-bound: no matrix cores, no staging of tiles.  The unit is "one lane builds one row as a sorted set":
// the set lives in LDS, entry k of lane l at word k * 64 + l (a wave's accesses to one k fall into 64 different banks' words:
// conflict-free whatever the lanes' set sizes are), insertion by bisection, duplicates -- nine candidates in ten -- rejected by the
// search.  Every pattern is built twice, once to count (row lengths -> slice widths -> slice bases by one workgroup's scan) and
// once to fill: the sets are cheap (a row of A P on the 4M-triangle panel: 7 neighbours x 5 aggregates), storing them between the
// two passes would need the lists the passes exist to avoid.  Orders that atomics leave open (nodes inside an aggregate, entries
// of an in-list) are sorted afterwards by the lane that owns the list, so every array is a function of its inputs alone and equal
// to what the host path (amg_device_setup.cpp) builds.
#include "amg_symbolic.hpp"

#include "device_common.hpp"

namespace femshell {

namespace {

constexpr int kLanes = 64;

template <int CAP> struct LaneSet {
    int32_t *s; // this lane's column of the workgroup's array
    int m = 0;
    bool over = false;
    __device__ __forceinline__ int32_t at(int k) const { return s[k * kLanes]; }
    __device__ __forceinline__ int lower_bound(int32_t v) const
    {
        int lo = 0, hi = m;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s[mid * kLanes] < v) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    __device__ __forceinline__ void insert(int32_t v)
    {
        const int lo = lower_bound(v);
        if (lo < m && s[lo * kLanes] == v) return;
        if (m == CAP) {
            over = true;
            return;
        }
        for (int k = m; k > lo; k--) s[k * kLanes] = s[(k - 1) * kLanes];
        s[lo * kLanes] = v;
        m++;
    }
};

// a finished ELL pattern as the builders of the next one read it
struct PatView {
    const int64_t *slice_base = nullptr;
    const int32_t *cols = nullptr;
    const uint8_t *count = nullptr;
};

__device__ __forceinline__ int real_slots(const GraphView &G, int a, int sl, int n)
{
    if (G.count != nullptr) return G.count[a];
    // the plan's convention: real slots first (slot 0 the diagonal), padding slots repeat the row's index
    const int W = G.slice_width[sl];
    const int64_t base = G.slice_base[sl] + n;
    int cnt = 1;
    for (int k = 1; k < W; k++)
        if (G.cols[base + (int64_t)k * kSliceNodes] != a) cnt = k + 1;
    return cnt;
}

// the neighbours of row a (the row itself first), own slots then in-list
template <class F> __device__ __forceinline__ void for_each_nb(const GraphView &G, int a, F f)
{
    const int sl = a / kSliceNodes, n = a % kSliceNodes;
    const int64_t base = G.slice_base[sl] + n;
    const int W = G.count != nullptr ? (int)G.count[a] : G.slice_width[sl];
    f(a);
    for (int k = 1; k < W; k++) {
        const int c = G.cols[base + (int64_t)k * kSliceNodes];
        if (G.count == nullptr && c == a) continue; // padding slot
        f(c);
    }
    if (G.symmetric) {
        const int Wi = G.in_width[sl];
        const int64_t ib = G.in_base[sl] + n;
        for (int k = 0; k < Wi; k++) {
            const int64_t e = ib + (int64_t)k * kSliceNodes;
            if (G.in_slots[e] >= 0) f(G.in_rows[e]);
        }
    }
}

template <class F> __device__ __forceinline__ void for_each_entry(const PatView &M, int r, F f)
{
    const int cnt = M.count[r];
    const int64_t base = M.slice_base[r / kSliceNodes] + (r % kSliceNodes);
    for (int k = 0; k < cnt; k++) f(M.cols[base + (int64_t)k * kSliceNodes]);
}

// index of column J in row r of a pattern whose rows ascend (0 when absent: the callers only ask for columns that are there)
__device__ __forceinline__ int index_in_row(const PatView &M, int r, int J)
{
    const int cnt = M.count[r];
    const int64_t base = M.slice_base[r / kSliceNodes] + (r % kSliceNodes);
    for (int k = 0; k < cnt; k++)
        if (M.cols[base + (int64_t)k * kSliceNodes] == J) return k;
    return 0;
}

// ---- the four row builders
// P: the aggregates of a fine row's neighbours
struct RowP {
    GraphView G;
    const int32_t *agg;
    template <class S> __device__ __forceinline__ void build(int a, S &set) const
    {
        for_each_nb(G, a, [&](int c) { set.insert(agg[c]); });
    }
};
// A P: the union of the P rows of a fine row's neighbours
struct RowAP {
    GraphView G;
    PatView P;
    template <class S> __device__ __forceinline__ void build(int a, S &set) const
    {
        for_each_nb(G, a, [&](int c) { for_each_entry(P, c, [&](int J) { set.insert(J); }); });
    }
};
// R = P^T: the fine rows that see an aggregate = the neighbourhoods of its members (the pattern of A is symmetric)
struct RowR {
    GraphView G;
    const int32_t *gptr, *order;
    template <class S> __device__ __forceinline__ void build(int I, S &set) const
    {
        for (int t = gptr[I]; t < gptr[I + 1]; t++) for_each_nb(G, order[t], [&](int c) { set.insert(c); });
    }
};
// A_c = R (A P): the union of the A P rows of the fine rows an aggregate is seen by
struct RowAc {
    const int64_t *rptr;
    const int32_t *rrow;
    PatView AP;
    template <class S> __device__ __forceinline__ void build(int I, S &set) const
    {
        for (int64_t q = rptr[I]; q < rptr[I + 1]; q++) for_each_entry(AP, rrow[q], [&](int J) { set.insert(J); });
    }
};

// info words of a pattern under construction
enum { kInfoNnzb = 0, kInfoOver = 1, kInfoTotal = 2, kInfoMax = 3, kInfoWords = 8 };

// Pass 1: row lengths.  upper: only the entries >= the row index count (symmetric storage of A_c); the ones below it are the row's
// in-list length (lower / in_width, optional).
template <class Row, int CAP>
__global__ __launch_bounds__(kLanes) void k_rows_count(Row row, int n_rows, int n_pad, int upper, uint8_t *__restrict__ count,
                                                       int32_t *__restrict__ slice_width, int32_t *__restrict__ count32,
                                                       int32_t *__restrict__ lower, int32_t *__restrict__ in_width, unsigned long long *info)
{
    __shared__ int32_t lds[CAP * kLanes];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * kLanes + lane;
    LaneSet<CAP> set;
    set.s = lds + lane;
    int cnt = 0, low = 0;
    if (r < n_rows) {
        row.build(r, set);
        if (upper) {
            low = set.lower_bound(r);
            cnt = set.m - low;
        } else {
            cnt = set.m;
        }
        if (set.over || cnt > 255 || low > 255) atomicOr(&info[kInfoOver], 1ull);
    }
    if (r < n_pad) {
        count[r] = (uint8_t)min(cnt, 255);
        if (count32 != nullptr) count32[r] = cnt;
        if (lower != nullptr) lower[r] = low;
    }
    int w = cnt, wl = low;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { // the 32 rows of a slice are half a wave
        w = max(w, __shfl_xor(w, d, 64));
        wl = max(wl, __shfl_xor(wl, d, 64));
    }
    if ((lane & 31) == 0 && r < n_pad) {
        slice_width[r / kSliceNodes] = max(w, 1);
        if (in_width != nullptr) in_width[r / kSliceNodes] = wl;
    }
    unsigned long long sum = (unsigned long long)cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0 && sum != 0) atomicAdd(&info[kInfoNnzb], sum);
}

// Pass 2: the columns into their slots (ascending; diag_first: the row's own index in slot 0; padding slots: column 0)
template <class Row, int CAP>
__global__ __launch_bounds__(kLanes) void k_rows_fill(Row row, int n_rows, int n_pad, int upper, int diag_first,
                                                      const int32_t *__restrict__ slice_width, const int64_t *__restrict__ slice_base,
                                                      int32_t *__restrict__ cols)
{
    __shared__ int32_t lds[CAP * kLanes];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * kLanes + lane;
    if (r >= n_pad) return;
    LaneSet<CAP> set;
    set.s = lds + lane;
    const int sl = r / kSliceNodes, n = r % kSliceNodes;
    const int W = slice_width[sl];
    const int64_t base = slice_base[sl] + n;
    int k = 0;
    if (r < n_rows) {
        row.build(r, set);
        const int first = upper ? set.lower_bound(r) : 0;
        if (diag_first) {
            cols[base] = r;
            k = 1;
        }
        for (int q = first; q < set.m; q++) {
            const int32_t v = set.at(q);
            if (diag_first && v == r) continue;
            if (k < W) cols[base + (int64_t)k * kSliceNodes] = v;
            k++;
        }
    }
    for (; k < W; k++) cols[base + (int64_t)k * kSliceNodes] = 0;
}

// out[i] = mult * (in[0] + ... + in[i-1]), i = 0 .. n; info[kInfoTotal] = out[n], info[kInfoMax] = the largest input.  One workgroup
// of 1024 walks the array in tiles of 8192 consecutive entries, eight per thread (a wave reads 2 KB in a row); a tile is scanned by
// the threads' own eight, wave shuffles and one exchange of the sixteen wave sums through LDS, the running total carried from tile
// to tile: 222,784 counters in 27 tiles.  (The first version gave every thread one contiguous piece of the whole array -- 64 cache
// lines per load of a wave, 1.2 ms for that array; tiles of 1024 with one entry per thread took 0.8 ms: 218 dependent round trips.)
constexpr int kScanItems = 8;
template <class TIn, class TOut>
__global__ __launch_bounds__(1024) void k_exclusive_scan(const TIn *__restrict__ in, int64_t n, int64_t mult, TOut *__restrict__ out,
                                                         unsigned long long *info)
{
    __shared__ long long wsum[2][16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    long long carry = 0, mx = 0;
    int flip = 0;
    for (int64_t base = 0; base < n; base += 1024 * kScanItems, flip ^= 1) {
        const int64_t i0 = base + (int64_t)t * kScanItems;
        long long v[kScanItems];
        long long mine = 0;
#pragma unroll
        for (int q = 0; q < kScanItems; q++) {
            v[q] = i0 + q < n ? (long long)in[i0 + q] : 0;
            mx = v[q] > mx ? v[q] : mx;
            mine += v[q] * mult;
        }
        long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[flip][w] = incl;
        __syncthreads(); // (the other half of wsum is what the previous tile's readers may still hold: one barrier per tile)
        long long before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const long long sk = wsum[flip][k];
            if (k < w) before += sk;
            total += sk;
        }
        long long run = carry + before + incl - mine;
#pragma unroll
        for (int q = 0; q < kScanItems; q++) {
            if (i0 + q < n) out[i0 + q] = (TOut)run;
            run += v[q] * mult;
        }
        carry += total;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const long long o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    __shared__ long long wmax[16];
    if (lane == 0) wmax[w] = mx;
    __syncthreads();
    if (t == 0) {
        long long gmax = 0;
        for (int k = 0; k < 16; k++) gmax = wmax[k] > gmax ? wmax[k] : gmax;
        out[n] = (TOut)carry;
        if (info != nullptr) {
            info[kInfoTotal] = (unsigned long long)carry;
            info[kInfoMax] = (unsigned long long)gmax;
        }
    }
}

// ---- nodes grouped by aggregate
__global__ __launch_bounds__(256) void k_histogram(const int32_t *__restrict__ key, int32_t n, int32_t *__restrict__ cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[key[i]], 1);
}
__global__ __launch_bounds__(256) void k_group_scatter(const int32_t *__restrict__ key, int32_t n, const int32_t *__restrict__ gptr,
                                                       int32_t *__restrict__ fill, int32_t *__restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int I = key[i];
        order[gptr[I] + atomicAdd(&fill[I], 1)] = i;
    }
}
// ascending inside every group (the atomics above leave the order open; groups hold some ten nodes)
__global__ __launch_bounds__(256) void k_group_sort(const int32_t *__restrict__ gptr, int32_t na, int32_t *__restrict__ order)
{
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= na) return;
    const int b = gptr[I], e = gptr[I + 1];
    for (int i = b + 1; i < e; i++) {
        const int32_t v = order[i];
        int j = i - 1;
        while (j >= b && order[j] > v) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = v;
    }
}

// ---- which slot of P's row every block of A feeds (own slots and in-list entries; padding: 0)
__global__ __launch_bounds__(256) void k_pmap(GraphView G, const int32_t *__restrict__ agg, PatView P, uint8_t *__restrict__ pmap_own,
                                              uint8_t *__restrict__ pmap_in)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= G.n_slices * kSliceNodes) return;
    const int sl = a / kSliceNodes, n = a % kSliceNodes;
    const int W = G.slice_width[sl];
    const int64_t base = G.slice_base[sl] + n;
    const int real = a < G.n ? real_slots(G, a, sl, n) : 0;
    for (int k = 0; k < W; k++) {
        const int64_t slot = base + (int64_t)k * kSliceNodes;
        pmap_own[slot] = k < real ? (uint8_t)index_in_row(P, a, agg[G.cols[slot]]) : (uint8_t)0;
    }
    if (G.symmetric) {
        const int Wi = G.in_width[sl];
        const int64_t ib = G.in_base[sl] + n;
        for (int k = 0; k < Wi; k++) {
            const int64_t e = ib + (int64_t)k * kSliceNodes;
            pmap_in[e] = (a < G.n && G.in_slots[e] >= 0) ? (uint8_t)index_in_row(P, a, agg[G.in_rows[e]]) : (uint8_t)0;
        }
    }
}

// ---- R as lists beside its ELL pattern: entry q of aggregate I = fine row rrow[q], whose P row holds I in slot rk[q]
__global__ __launch_bounds__(256) void k_r_lists(PatView R, int32_t na, const int64_t *__restrict__ rptr, PatView P, int32_t *__restrict__ rrow,
                                                 uint8_t *__restrict__ rk)
{
    const int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= na) return;
    const int cnt = R.count[I];
    const int64_t base = R.slice_base[I / kSliceNodes] + (I % kSliceNodes);
    const int64_t q0 = rptr[I];
    for (int k = 0; k < cnt; k++) {
        const int i = R.cols[base + (int64_t)k * kSliceNodes];
        rrow[q0 + k] = i;
        rk[q0 + k] = (uint8_t)index_in_row(P, i, I);
    }
}

// ---- in-lists of the symmetric coarse operator: row c lists the stored blocks (a, c), a < c, by ascending slot index
__global__ __launch_bounds__(256) void k_in_scatter(PatView Ac, int32_t na, const int64_t *__restrict__ in_base, int32_t *__restrict__ fill,
                                                    int32_t *__restrict__ in_slots)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= na) return;
    const int cnt = Ac.count[a];
    const int64_t base = Ac.slice_base[a / kSliceNodes] + (a % kSliceNodes);
    for (int k = 1; k < cnt; k++) {
        const int64_t slot = base + (int64_t)k * kSliceNodes;
        const int c = Ac.cols[slot];
        const int pos = atomicAdd(&fill[c], 1);
        in_slots[in_base[c / kSliceNodes] + (int64_t)pos * kSliceNodes + (c % kSliceNodes)] = (int32_t)slot;
    }
}
__global__ __launch_bounds__(256) void k_in_sort(int32_t na, int32_t n_slices, const int64_t *__restrict__ ac_slice_base, const int32_t *__restrict__ lower,
                                                 const int64_t *__restrict__ in_base, int32_t *__restrict__ in_slots, int32_t *__restrict__ in_rows)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= na) return;
    const int w = lower[c];
    const int64_t ib = in_base[c / kSliceNodes] + (c % kSliceNodes);
    for (int i = 1; i < w; i++) {
        const int32_t v = in_slots[ib + (int64_t)i * kSliceNodes];
        int j = i - 1;
        while (j >= 0 && in_slots[ib + (int64_t)j * kSliceNodes] > v) {
            in_slots[ib + (int64_t)(j + 1) * kSliceNodes] = in_slots[ib + (int64_t)j * kSliceNodes];
            j--;
        }
        in_slots[ib + (int64_t)(j + 1) * kSliceNodes] = v;
    }
    for (int i = 0; i < w; i++) {
        const int64_t slot = in_slots[ib + (int64_t)i * kSliceNodes];
        int lo = 0, hi = n_slices - 1; // the slice that holds the slot
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ac_slice_base[mid] <= slot) lo = mid;
            else hi = mid - 1;
        }
        in_rows[ib + (int64_t)i * kSliceNodes] = lo * kSliceNodes + (int)((slot - ac_slice_base[lo]) % kSliceNodes);
    }
}

// ---- work of the Galerkin product (AmgSetupStats): useful = 432 flops per (fine row, aggregate of its P row, block of its A P row);
// issued on the matrix cores = 16x16x4 tiles, two k-steps per fine row and panel tile
__global__ __launch_bounds__(256) void k_galerkin_work(int32_t n, const uint8_t *__restrict__ pcount, const uint8_t *__restrict__ apcount, int32_t na,
                                                       const uint8_t *__restrict__ account, const int32_t *__restrict__ rcount, unsigned long long *info)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long useful = 0, issued = 0;
    if (t < n) useful = 432ull * pcount[t] * apcount[t];
    if (t < na) {
        const int cnt = account[t];
        unsigned long long tiles = 0;
        for (int g0 = 0; g0 < cnt; g0 += 16) tiles += (unsigned long long)((6 * min(16, cnt - g0) + 15) / 16);
        issued = 4096ull * tiles * (unsigned long long)rcount[t];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        useful += __shfl_xor(useful, d, 64);
        issued += __shfl_xor(issued, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (useful) atomicAdd(&info[4], useful);
        if (issued) atomicAdd(&info[5], issued);
    }
}

struct InfoWords {
    unsigned long long w[kInfoWords];
};

// One pattern: count, slice bases, fill.  Returns the scalars in *E (host) and the arrays in *D.
template <class Row, int CAP>
int build_pattern(hipStream_t st, const Row &row, int32_t n_rows, bool upper, bool diag_first, DevBuf<unsigned long long> &info, DevPattern *D,
                  EllPattern *E, int64_t *total, int32_t *count32, int32_t *lower, int32_t *in_width)
{
    *E = EllPattern();
    E->n_rows = n_rows;
    E->n_pad = (n_rows + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    E->n_slices = E->n_pad / kSliceNodes;
    FS_HIP(D->slice_width.alloc((size_t)E->n_slices));
    FS_HIP(D->slice_base.alloc((size_t)E->n_slices + 1));
    FS_HIP(D->count.alloc((size_t)E->n_pad));
    FS_HIP(info.zero(st));
    const unsigned grid = (unsigned)((E->n_pad + kLanes - 1) / kLanes);
    hipLaunchKernelGGL((k_rows_count<Row, CAP>), dim3(grid), dim3(kLanes), 0, st, row, n_rows, E->n_pad, upper ? 1 : 0, D->count.p, D->slice_width.p,
                       count32, lower, in_width, info.p);
    hipLaunchKernelGGL((k_exclusive_scan<int32_t, int64_t>), dim3(1), dim3(1024), 0, st, D->slice_width.p, (int64_t)E->n_slices, (int64_t)kSliceNodes,
                       D->slice_base.p, info.p);
    FS_HIP(hipGetLastError());
    InfoWords h;
    FS_HIP(hipMemcpyAsync(h.w, info.p, sizeof(h.w), hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    if (h.w[kInfoOver] != 0) return FEMSHELL_ERR_UNSUPPORTED;
    E->nnzb = (int64_t)h.w[kInfoNnzb];
    E->max_width = (int32_t)h.w[kInfoMax];
    *total = (int64_t)h.w[kInfoTotal];
    E->slice_base.assign(1, *total); // (scalars only: total() reads the last entry)
    FS_HIP(D->cols.alloc((size_t)*total));
    hipLaunchKernelGGL((k_rows_fill<Row, CAP>), dim3(grid), dim3(kLanes), 0, st, row, n_rows, E->n_pad, upper ? 1 : 0, diag_first ? 1 : 0,
                       D->slice_width.p, D->slice_base.p, D->cols.p);
    FS_HIP(hipGetLastError());
    return FEMSHELL_OK;
}

PatView view_of(const DevPattern &D)
{
    PatView v;
    v.slice_base = D.slice_base.p;
    v.cols = D.cols.p;
    v.count = D.count.p;
    return v;
}

} // namespace

int amg_symbolic_device(femshell_ctx *c, hipStream_t st, const GraphView &G, int64_t total_slots, int64_t in_total, const std::vector<int32_t> &agg, int32_t na,
                        bool sym_coarse, DevSymbolic *out)
{
    DevSymbolic &S = *out;
    const int32_t n = G.n;
    const int32_t na_pad = (na + kSliceNodes - 1) / kSliceNodes * kSliceNodes;
    DevBuf<unsigned long long> info;
    FS_HIP(info.alloc(kInfoWords));
    FS_HIP(S.agg.alloc(agg.size()));
    {
        const int rcu = staged_upload(c, S.agg.p, agg.data(), agg.size() * sizeof(int32_t), st);
        if (rcu) return rcu;
    }
    // ---- nodes by aggregate (the tentative prolongator's QR runs per aggregate; R's rows are collected from the members)
    DevBuf<int32_t> gcnt;
    FS_HIP(gcnt.alloc((size_t)na));
    FS_HIP(gcnt.zero(st));
    FS_HIP(S.gptr.alloc((size_t)na + 1));
    FS_HIP(S.order.alloc((size_t)n));
    FS_HIP(info.zero(st));
    hipLaunchKernelGGL(k_histogram, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S.agg.p, n, gcnt.p);
    hipLaunchKernelGGL((k_exclusive_scan<int32_t, int32_t>), dim3(1), dim3(1024), 0, st, gcnt.p, (int64_t)na, (int64_t)1, S.gptr.p, info.p);
    FS_HIP(gcnt.zero(st));
    hipLaunchKernelGGL(k_group_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S.agg.p, n, S.gptr.p, gcnt.p, S.order.p);
    hipLaunchKernelGGL(k_group_sort, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, S.gptr.p, na, S.order.p);
    FS_HIP(hipGetLastError());
    {
        InfoWords h;
        FS_HIP(hipMemcpyAsync(h.w, info.p, sizeof(h.w), hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        S.largest = (int32_t)h.w[kInfoMax];
    }
    // ---- P and the slots of P the blocks of A feed
    RowP rp{G, S.agg.p};
    int rc = build_pattern<RowP, 64>(st, rp, n, false, false, info, &S.P, &S.iP, &S.totP, nullptr, nullptr, nullptr);
    if (rc) return rc;
    FS_HIP(S.pmap_own.alloc((size_t)total_slots));
    FS_HIP(S.pmap_in.alloc((size_t)std::max<int64_t>(in_total, 1)));
    hipLaunchKernelGGL(k_pmap, dim3((unsigned)((G.n_slices * kSliceNodes + 255) / 256)), dim3(256), 0, st, G, S.agg.p, view_of(S.P), S.pmap_own.p,
                       S.pmap_in.p);
    // ---- A P
    RowAP rap{G, view_of(S.P)};
    rc = build_pattern<RowAP, 128>(st, rap, n, false, false, info, &S.AP, &S.iAP, &S.totAP, nullptr, nullptr, nullptr);
    if (rc) return rc;
    // ---- R: ELL pattern (columns = fine rows) and the lists the restriction and the Galerkin product walk
    DevBuf<int32_t> rcount;
    FS_HIP(rcount.alloc((size_t)na_pad));
    RowR rr{G, S.gptr.p, S.order.p};
    rc = build_pattern<RowR, 256>(st, rr, na, false, false, info, &S.R, &S.iR, &S.totR, rcount.p, nullptr, nullptr);
    if (rc) return rc;
    FS_HIP(S.rptr.alloc((size_t)na + 1));
    FS_HIP(S.rrow.alloc((size_t)std::max<int64_t>(S.iR.nnzb, 1)));
    FS_HIP(S.rk.alloc((size_t)std::max<int64_t>(S.iR.nnzb, 1)));
    hipLaunchKernelGGL((k_exclusive_scan<int32_t, int64_t>), dim3(1), dim3(1024), 0, st, rcount.p, (int64_t)na, (int64_t)1, S.rptr.p,
                       (unsigned long long *)nullptr);
    hipLaunchKernelGGL(k_r_lists, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, view_of(S.R), na, S.rptr.p, view_of(S.P), S.rrow.p, S.rk.p);
    FS_HIP(hipGetLastError());
    // ---- A_c (symmetric storage: diagonal and upper blocks; the blocks below the diagonal are the row's in-list)
    DevBuf<int32_t> lower;
    RowAc rac{S.rptr.p, S.rrow.p, view_of(S.AP)};
    if (sym_coarse) {
        FS_HIP(lower.alloc((size_t)na_pad));
        FS_HIP(S.in_width.alloc((size_t)(na_pad / kSliceNodes)));
        FS_HIP(S.in_base.alloc((size_t)(na_pad / kSliceNodes) + 1));
    }
    rc = build_pattern<RowAc, 256>(st, rac, na, sym_coarse, true, info, &S.Ac, &S.iAc, &S.totAc, nullptr, sym_coarse ? lower.p : nullptr,
                                   sym_coarse ? S.in_width.p : nullptr);
    if (rc) return rc;
    if (sym_coarse) {
        const int32_t ns = na_pad / kSliceNodes;
        FS_HIP(info.zero(st));
        hipLaunchKernelGGL((k_exclusive_scan<int32_t, int64_t>), dim3(1), dim3(1024), 0, st, S.in_width.p, (int64_t)ns, (int64_t)kSliceNodes, S.in_base.p,
                           info.p);
        InfoWords h;
        FS_HIP(hipMemcpyAsync(h.w, info.p, sizeof(h.w), hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st));
        S.in_total = (int64_t)h.w[kInfoTotal];
        S.max_in_width = (int32_t)h.w[kInfoMax];
        FS_HIP(S.in_slots.alloc((size_t)std::max<int64_t>(S.in_total, 1)));
        FS_HIP(S.in_rows.alloc((size_t)std::max<int64_t>(S.in_total, 1)));
        FS_HIP(hipMemsetAsync(S.in_slots.p, 0xFF, S.in_slots.n * sizeof(int32_t), st)); // -1: no entry
        FS_HIP(S.in_rows.zero(st));
        FS_HIP(gcnt.zero(st)); // (na counters again)
        hipLaunchKernelGGL(k_in_scatter, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, view_of(S.Ac), na, S.in_base.p, gcnt.p, S.in_slots.p);
        hipLaunchKernelGGL(k_in_sort, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, na, ns, S.Ac.slice_base.p, lower.p, S.in_base.p, S.in_slots.p,
                           S.in_rows.p);
        FS_HIP(hipGetLastError());
    }
    // ---- statistics of the Galerkin product
    {
        FS_HIP(info.zero(st));
        const int32_t m = std::max(n, na);
        hipLaunchKernelGGL(k_galerkin_work, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, n, S.P.count.p, S.AP.count.p, na, S.Ac.count.p, rcount.p,
                           info.p);
        InfoWords h;
        FS_HIP(hipMemcpyAsync(h.w, info.p, sizeof(h.w), hipMemcpyDeviceToHost, st));
        FS_HIP(hipStreamSynchronize(st)); // (also: rcount, lower, gcnt go out of scope)
        S.useful_flops = (double)h.w[4];
        S.mfma_flops = (double)h.w[5];
    }
    return FEMSHELL_OK;
}

int download_pattern(femshell_ctx *c, const DevPattern &D, int64_t total, EllPattern *E, hipStream_t st)
{
    E->slice_width.resize((size_t)E->n_slices);
    E->slice_base.resize((size_t)E->n_slices + 1);
    E->cols.resize((size_t)total);
    E->count.resize((size_t)E->n_pad);
    int rc = staged_download(c, E->slice_width.data(), D.slice_width.p, E->slice_width.size() * sizeof(int32_t), st);
    if (!rc) rc = staged_download(c, E->slice_base.data(), D.slice_base.p, E->slice_base.size() * sizeof(int64_t), st);
    if (!rc) rc = staged_download(c, E->cols.data(), D.cols.p, (size_t)total * sizeof(int32_t), st);
    if (!rc) rc = staged_download(c, E->count.data(), D.count.p, E->count.size(), st);
    return rc;
}

} // namespace femshell
