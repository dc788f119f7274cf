// ell-matrix.hpp -- ELLPACK storage (row-major, every row padded to the longest), CPU y += A*x.
// Mirrors src/matrix/ell-matrix.hpp:22-100.
#pragma once

#include "aligned-vector.hpp"

#include <cstdint>

namespace matrix_market { class Matrix; }

namespace ell_matrix {

typedef int32_t size_type;
typedef int32_t index_type;
typedef double value_type;
typedef aligned_vector<index_type> index_array_type;
typedef aligned_vector<value_type> value_array_type;

struct Matrix
{
    Matrix() = default;
    Matrix(index_type rows, index_type columns, size_type num_entries, index_type row_length,
           index_array_type column_index, value_array_type value, bool skip_padding = false);
    Matrix(Matrix const &) = delete;
    Matrix & operator=(Matrix const &) = delete;
    Matrix(Matrix &&) = default;
    Matrix & operator=(Matrix &&) = default;

    std::size_t size() const;
    std::size_t value_size() const;
    std::size_t index_size() const;
    size_type num_padding_entries() const;

    index_type rows = 0;
    index_type columns = 0;
    size_type num_entries = 0; // true entries; rows*row_length are stored
    index_type row_length = 0;
    index_array_type column_index; // [rows * row_length], k = i*row_length + l
    value_array_type value;
    bool skip_padding = false;
};

bool operator==(Matrix const & a, Matrix const & b);

// row_length = longest row; throws matrix_error("Failed to convert to ELLPACK: Integer overflow
// when computing number of non-zeros") when rows*row_length does not fit int32
// (src/matrix/ell-matrix.cpp:199-205).  Padding: value 0.0, column = the column of the last real
// entry placed so far (so padded lanes re-read a nearby x), or INT32_MAX with skip_padding.
// Where the reference would read before the start of its array (row 0 empty) the pad column is 0.
Matrix from_matrix_market(matrix_market::Matrix const & m, bool skip_padding = false);

// y += A*x over all rows*row_length stored entries (padding multiplies 0.0), or stopping at the
// INT32_MAX sentinel when A.skip_padding.  Orphaned OpenMP loop, static row blocks.
void spmv(Matrix const & A, value_array_type const & x, value_array_type & y, index_type chunk_size = 0);

value_array_type operator*(Matrix const & A, value_array_type const & x);

} // namespace ell_matrix
