// reorder.cpp -- optional node renumbering inside femshell_set_mesh (FEMSHELL_REORDER_MORTON / FEMSHELL_REORDER_RCM).
// The library's row slices are 32 consecutive nodes, and a slice's SpMV gathers the x entries of all its neighbours:
// with a numbering that follows the geometry those come from a few cache lines, with an arbitrary one (a mesh
// generator's insertion order, a shuffled file) from everywhere.  libMesh renumbers nodes for locality by default --
// the reference switches that off only because its force file is indexed by the original ids (fem-shell.cpp:36,
// doc/implementation.tex:150-160); here the permutation stays inside the library and every node-indexed argument of
// the C ABI keeps the caller's numbering.
#include "reorder.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace femshell {

// nodes along a Morton (Z-order) curve through the bounding box, 21 bits per axis
void morton_order(int32_t n, const double *xyz, std::vector<int32_t> *perm)
{
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int32_t a = 0; a < n; a++)
        for (int d = 0; d < 3; d++) {
            lo[d] = std::min(lo[d], xyz[3ll * a + d]);
            hi[d] = std::max(hi[d], xyz[3ll * a + d]);
        }
    double scale[3];
    for (int d = 0; d < 3; d++) scale[d] = hi[d] > lo[d] ? 2097151.0 / (hi[d] - lo[d]) : 0.0;
    auto spread = [](uint64_t v) { // 21 bits -> every third bit
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    std::vector<uint64_t> key((size_t)n);
    for (int32_t a = 0; a < n; a++) {
        uint64_t k = 0;
        for (int d = 0; d < 3; d++) k |= spread((uint64_t)((xyz[3ll * a + d] - lo[d]) * scale[d])) << d;
        key[a] = k;
    }
    perm->resize((size_t)n);
    std::iota(perm->begin(), perm->end(), 0);
    std::stable_sort(perm->begin(), perm->end(), [&](int32_t x, int32_t y) { return key[x] < key[y]; });
}

// reverse Cuthill-McKee on the node graph of the elements; components in order of their lowest-degree node
void rcm_order(int32_t n, int32_t n_tri, const int32_t *tri, int32_t n_quad, const int32_t *quad, std::vector<int32_t> *perm)
{
    std::vector<int64_t> ptr((size_t)n + 1, 0);
    auto each_edge = [&](auto f) {
        for (int32_t e = 0; e < n_tri; e++)
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    if (i != j) f(tri[3ll * e + i], tri[3ll * e + j]);
        for (int32_t e = 0; e < n_quad; e++)
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    if (i != j) f(quad[4ll * e + i], quad[4ll * e + j]);
    };
    each_edge([&](int32_t a, int32_t) { ptr[a + 1]++; });
    for (int32_t a = 0; a < n; a++) ptr[a + 1] += ptr[a];
    std::vector<int32_t> adj((size_t)ptr[n]);
    {
        std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
        each_edge([&](int32_t a, int32_t b) { adj[(size_t)fill[a]++] = b; });
    }
    std::vector<int32_t> deg((size_t)n);
    for (int32_t a = 0; a < n; a++) { // unique neighbours
        auto b = adj.begin() + ptr[a], e = adj.begin() + ptr[a + 1];
        std::sort(b, e);
        deg[a] = (int32_t)(std::unique(b, e) - b);
    }
    std::vector<int32_t> by_degree((size_t)n);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int32_t x, int32_t y) { return deg[x] < deg[y]; });
    std::vector<char> seen((size_t)n, 0);
    std::vector<int32_t> order, nb;
    order.reserve((size_t)n);
    for (int32_t start : by_degree) {
        if (seen[start]) continue;
        seen[start] = 1;
        size_t head = order.size();
        order.push_back(start);
        while (head < order.size()) {
            const int32_t a = order[head++];
            nb.clear();
            for (int64_t q = ptr[a]; q < ptr[a] + deg[a]; q++)
                if (!seen[adj[(size_t)q]]) {
                    seen[adj[(size_t)q]] = 1;
                    nb.push_back(adj[(size_t)q]);
                }
            std::stable_sort(nb.begin(), nb.end(), [&](int32_t x, int32_t y) { return deg[x] < deg[y]; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(order.begin(), order.end());
    perm->swap(order);
}

} // namespace femshell
