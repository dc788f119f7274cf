// coo-matrix.hpp -- coordinate storage in FILE ORDER, and the CPU y += A*x.
// Mirrors src/matrix/coo-matrix.hpp:22-96.
#pragma once

#include "aligned-vector.hpp"

#include <cstdint>

namespace matrix_market { class Matrix; }

namespace coo_matrix {

typedef int32_t size_type;
typedef int32_t index_type;
typedef double value_type;
typedef aligned_vector<index_type> index_array_type;
typedef aligned_vector<value_type> value_array_type;

struct Matrix
{
    Matrix() = default;
    Matrix(index_type rows, index_type columns, size_type num_entries, index_array_type row_index,
           index_array_type column_index, value_array_type value);
    Matrix(Matrix const &) = delete;
    Matrix & operator=(Matrix const &) = delete;
    Matrix(Matrix &&) = default;
    Matrix & operator=(Matrix &&) = default;

    std::size_t size() const;
    std::size_t value_size() const;
    std::size_t index_size() const;

    index_type rows = 0;
    index_type columns = 0;
    size_type num_entries = 0;
    index_array_type row_index;
    index_array_type column_index;
    value_array_type value;
};

bool operator==(Matrix const & a, Matrix const & b);

Matrix from_matrix_market(matrix_market::Matrix const & m);

// y += A*x.  One thread: entries in file order.  More threads: every thread scatters its
// static block of entries into its private slice of `workspace` (num_threads*rows doubles),
// then the slices are added into y row by row.  As in the reference the workspace is NOT
// cleared here (it is zeroed once by whoever allocates it), so with several threads repeated
// calls re-add earlier products.  Call from every thread of a parallel region.
void spmv(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
          value_array_type & workspace, index_type chunk_size = 0);

// y += A*x with atomic updates of y (src/matrix/coo-matrix.cpp:287-309).
void spmv_atomic(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
                 index_type chunk_size = 0);

value_array_type operator*(Matrix const & A, value_array_type const & x);

} // namespace coo_matrix
