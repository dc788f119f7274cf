// matrix-reorder.hpp -- symmetric reordering of a loaded matrix.
//
// The reference reorders a matrix when its path ends in "__RCM" (reverse Cuthill-McKee) or
// "__GP<n>" (METIS k-way partition) -- src/matrix/matrix-market.cpp:782-802,
// src/matrix/matrix-market-reorder.cpp.  RCM is reproduced step for step (same start-node rule,
// same neighbour ordering by out-degree through std::sort, same reversal), so that the
// permutation is the one the reference computes with this toolchain.  METIS -- a third-party
// library the reference links optionally -- is not available here, and "__GP<n>" does what a
// reference build without USE_METIS does (:172-181): one warning, the order unchanged.
// EXTENSION: "__GPX<n>" clusters the rows with the repo's own k-way partitioner
// (kPartitionerName, greedy graph growing, matrix-reorder.cpp) and orders them exactly as the
// reference orders METIS's parts -- under a name of its own, so that nobody takes it for METIS.
//
// Why it matters on a GPU: reordering narrows the band, which shortens the x window of a tile
// (more tiles qualify for 16-bit column offsets) and turns scattered gathers into cache hits.
#pragma once

#include "matrix-market.hpp"

#include <iosfwd>
#include <string>
#include <vector>

namespace matrix_market {

// new_order[old index] = new index (0-based), for a square, real, coordinate matrix.
std::vector<int> find_new_order_RCM(Matrix const & m, std::ostream & log, bool verbose);
std::vector<int> find_new_order_GP(Matrix const & m, int nparts, std::ostream & log, bool verbose);   // identity + the reference's warning
std::vector<int> find_new_order_GPX(Matrix const & m, int nparts, std::ostream & log, bool verbose);  // EXTENSION: own partitioner
constexpr char const * kPartitionerName = "greedy-bfs-kway";

// The reordering suffixes of a matrix path, parsed as the reference parses them (matrix-market.cpp: parse_reorder_suffixes)
struct ReorderSuffixes
{
    std::string file;   // the path without them
    bool rcm = false, gp = false, gpx = false;
    int nparts = 0;
};
ReorderSuffixes parse_reorder_suffixes(std::string const & path);

// What the suffixes of `path` ask for, for the JSON document: "" (none), "rcm", "gp (no METIS in this build: order unchanged)",
// "gpx:greedy-bfs-kway:<n>", joined by '+' when both are present.
std::string reordering_of(std::string const & path);

// (i, j) -> (new_order[i], new_order[j]) for every entry (reference Matrix::permute,
// src/matrix/matrix-market.cpp:309-334); file order of the entries is kept.
Matrix permute(Matrix const & m, std::vector<int> const & new_order);

// largest |i - j| over the entries
int bandwidth(Matrix const & m);

} // namespace matrix_market
