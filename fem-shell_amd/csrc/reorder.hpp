// reorder.hpp -- node orderings for the optional renumbering inside femshell_set_mesh (reorder.cpp)
#pragma once

#include <cstdint>
#include <vector>

namespace femshell {

// perm[new index] = caller's node id
void morton_order(int32_t n, const double *xyz, std::vector<int32_t> *perm);
void rcm_order(int32_t n, int32_t n_tri, const int32_t *tri, int32_t n_quad, const int32_t *quad, std::vector<int32_t> *perm);

} // namespace femshell
