// matrix-cache.hpp -- optional binary cache of parsed Matrix Market files.
//
// Parsing text dominates the time to first multiply for large matrices (nlpkkt200 has 232 M lines);
// the parsed coordinate entries are therefore kept in a binary file next to nothing else, keyed by
// the source file's identity (canonical path, size, modification time), and read back with three
// bulk reads.  Off unless a cache directory is given (environment SPMV_MATRIX_CACHE or the CLI's
// --matrix-cache DIR).  The reference has no counterpart; what comes out of the cache is the same
// Matrix that parsing the file gives (tests/test_host.py::test_matrix_cache_*).
#pragma once

#include <ostream>
#include <string>

#include "matrix-market.hpp"

namespace matrix_market
{

// Directory of the cache, empty = off (reads the environment once per call).
std::string cache_directory();

// File the cache keeps for `source` (which must exist), or "" when the cache is off or the source
// cannot be identified.
std::string cache_file_for(std::string const & source, std::string const & directory);

// true and `m` filled when a valid cache entry for `source` exists.
bool load_cached(std::string const & source, std::string const & directory, Matrix & m);

// Store `m` (the result of parsing `source`); failures are silent -- a cache is best effort.
void store_cached(std::string const & source, std::string const & directory, Matrix const & m);

} // namespace matrix_market
