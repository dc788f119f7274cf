// synthetic.hpp -- generated matrices that stand in for files which cannot be fetched.
//
// EXTENSION (the reference only reads files, src/matrix/matrix-market.cpp:777-861): a matrix
// "path" of the form  synthetic:<family>[:<parameters>]  is generated in memory instead of
// being read, so that the full-size configurations of BASELINE.json run on a box without the
// SuiteSparse files and without multi-gigabyte text round trips.  Everything is a pure function
// of the specification (counter-based hashing, no thread-count dependence): the same spec gives
// the same arrays everywhere, bit for bit.
//
//   synthetic:poisson2d:<n>[,1]      5-point stencil on an n x n grid (BASELINE configs[1]: n = 4096); ",1" = the
//                                    pessimistic twin: every coefficient perturbed by a hash (84 M distinct values)
//   synthetic:queen[:gx,gy,gz[,J]]   Queen_4147-like (configs[2]): 3 unknowns per node of a gx*gy*gz
//                                    mesh, half of whose 26 neighbour links are moved by 3 nodes,
//                                    symmetric structure, dense 3x3 blocks, ~81 entries per row;
//                                    default 110,71,177 -> N = 4 147 110, ~330 M entries; J (default 3) = how far a
//                                    moved link ends from its grid neighbour (twin: 6)
//   synthetic:kkt[:<n>[,<jitter %>]] nlpkkt200-like (configs[3]): KKT matrix [H 0 A'; 0 R C'; A C 0]
//                                    of a boundary-control problem on an n^3 grid, A = 27-point
//                                    operator; N = 2n^3 + 6n^2 (n = 200: 16 240 000), ~436 M entries,
//                                    two row populations (28 and <= 30 entries) plus 2-entry control rows;
//                                    jitter (default 0): that share of the stencil links is moved by 3 cells, hashed
//                                    per row, so that no interior row is a shifted copy of its neighbour (twin: 50)
//   synthetic:webbase[:N,Z,maxrow,locality%]
//                                    webbase-1M-like (configs[4]): power-law row lengths (every row
//                                    >= 1 entry, longest = maxrow, exactly Z entries), `locality` per
//                                    cent of a row's links inside its host block, the rest to
//                                    power-law-popular pages; default 1000005,3105536,4700,75
//   synthetic:powerlaw[:N,Z,maxrow]  the same with locality 0: uniformly scattered columns (worst case)
//   synthetic:banded:N,b[,seed]      SURVEY 8d S-banded: N rows, the 2b + 1 diagonals -b .. +b, values U(-1,1)
//   synthetic:scrambled:N,b[,seed]   the same band matrix with rows and columns renumbered by a pseudo-random permutation (P B P^T): what
//                                    the reordering suffixes __RCM / __GP<n> are for
//   synthetic:random:N,k[,seed]      SURVEY 8d S-random: k distinct uniform columns per row, ascending, values U(-1,1)
//   synthetic:<any of the above>:tril
//                                    the STORED TRIANGLE of that matrix (entries with column <= row): what a Matrix Market file
//                                    with a `symmetric` header holds and what the reference multiplies when it is handed such a
//                                    file (it mirrors nothing: src/matrix/matrix-market.cpp:530-555, README.md:106).  As
//                                    coordinate entries the matrix carries the `symmetric` header word, so that
//                                    --expand-symmetric gives the whole matrix back.  queen:tril = 168.8 M entries
//                                    (Queen_4147's file: 166.8 M), kkt:200:tril = 222 M (nlpkkt200's file: 232.2 M)
//
// What these are NOT: the SuiteSparse matrices themselves.  They reproduce size, row-length
// populations, symmetry of structure and the kind of column locality of their namesakes
// (DESIGN.md section 5 states the assumptions); values are U(-1,1) hashes.
#pragma once

#include "csr-matrix.hpp"
#include "matrix-market.hpp"

#include <string>

namespace synthetic {

// true for paths that start with "synthetic:"
bool is_spec(std::string const & path);

// true for a spec that ends in ":tril" (the stored triangle of a symmetric matrix)
bool is_stored_triangle(std::string const & spec);

// The matrix of `spec`, rows [row_begin, row_end) (row_end < 0: all rows) as its own CSR matrix:
// row_ptr rebased to 0, column indices global, columns ascending inside every row.
// `rows_total`, if given, receives the row count of the whole matrix.  Throws matrix_error on a
// bad spec or when the entries do not fit int32.
csr_matrix::Matrix generate_csr(std::string const & spec, long long row_begin = 0, long long row_end = -1,
                                long long * rows_total = nullptr);

// The same matrix as coordinate entries in row-major order ("general", real): the input of the
// COO / ELLPACK / hybrid converters.
matrix_market::Matrix generate(std::string const & spec);

} // namespace synthetic
