// matrix-reorder.hpp -- symmetric reordering of a loaded matrix.
//
// The reference reorders a matrix when its path ends in "__RCM" (reverse Cuthill-McKee) or
// "__GP<n>" (METIS k-way partition) -- src/matrix/matrix-market.cpp:782-802,
// src/matrix/matrix-market-reorder.cpp.  RCM is reproduced step for step (same start-node rule,
// same neighbour ordering by out-degree through std::sort, same reversal), so that the
// permutation is the one the reference computes with this toolchain.  METIS -- a third-party
// library the reference links optionally -- is not available here; where a reference build
// without USE_METIS leaves the order unchanged, "__GP<n>" here clusters the rows with the repo's
// own k-way partitioner (greedy graph growing, matrix-reorder.cpp) and orders them exactly as
// the reference orders METIS's parts: a documented stand-in, not METIS's partition.
//
// Why it matters on a GPU: reordering narrows the band, which shortens the x window of a tile
// (more tiles qualify for 16-bit column offsets) and turns scattered gathers into cache hits.
#pragma once

#include "matrix-market.hpp"

#include <iosfwd>
#include <vector>

namespace matrix_market {

// new_order[old index] = new index (0-based), for a square, real, coordinate matrix.
std::vector<int> find_new_order_RCM(Matrix const & m, std::ostream & log, bool verbose);
std::vector<int> find_new_order_GP(Matrix const & m, int nparts, std::ostream & log, bool verbose);

// (i, j) -> (new_order[i], new_order[j]) for every entry (reference Matrix::permute,
// src/matrix/matrix-market.cpp:309-334); file order of the entries is kept.
Matrix permute(Matrix const & m, std::vector<int> const & new_order);

// largest |i - j| over the entries
int bandwidth(Matrix const & m);

} // namespace matrix_market
