// hybrid-matrix.hpp -- ELLPACK for the bulk of every row + COO for what is longer.
//
// Mirrors src/matrix/hybrid-matrix.hpp:25-110: the ELL row length is the "2/3 median" of the
// row lengths, so two thirds of the rows fit without spilling; the only reference format that can
// hold a skewed matrix such as webbase-1M, where plain ELLPACK overflows int32.
#pragma once

#include "aligned-vector.hpp"

#include <cstdint>

namespace matrix_market { class Matrix; }

namespace hybrid_matrix {

typedef int32_t size_type;
typedef int32_t index_type;
typedef double value_type;
typedef aligned_vector<index_type> index_array_type;
typedef aligned_vector<value_type> value_array_type;

struct Matrix
{
    Matrix() = default;
    Matrix(Matrix const &) = delete;
    Matrix & operator=(Matrix const &) = delete;
    Matrix(Matrix &&) = default;
    Matrix & operator=(Matrix &&) = default;

    // value bytes + ELL column bytes + COO column bytes: the reference's count, which leaves
    // the COO row indices out (src/matrix/hybrid-matrix.cpp:72-87)
    std::size_t size() const;

    index_type rows = 0;
    index_type columns = 0;
    size_type num_entries = 0;

    index_type ell_row_length = 0;
    size_type num_ell_entries = 0; // rows * ell_row_length, padding included
    index_array_type ell_column_index; // row-major, k = i*ell_row_length + l
    value_array_type ell_value;
    bool ell_skip_padding = false;

    size_type num_coo_entries = 0;
    index_array_type coo_row_index; // (row, column) order
    index_array_type coo_column_index;
    value_array_type coo_value;
};

Matrix from_matrix_market(matrix_market::Matrix const & m, bool ell_skip_padding = false);

// y += A*x: the ELL part over static row blocks, then the COO remainder (one thread: in order;
// more: private workspaces folded into y, chunk = ceil(rows/num_threads) for every loop, the
// workspace never cleared here -- src/matrix/hybrid-matrix.cpp:535-567).  Orphaned OpenMP loops.
void spmv(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
          value_array_type & workspace, index_type chunk_size = 0);

} // namespace hybrid_matrix
