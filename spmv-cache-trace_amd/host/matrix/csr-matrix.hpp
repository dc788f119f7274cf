// csr-matrix.hpp -- compressed sparse row storage, converter, and the CPU y += A*x.
//
// Mirrors the reference's csr_matrix interface (src/matrix/csr-matrix.hpp:22-65, :67-82):
// int32 row_ptr / column_index, fp64 value, optional zero padding of every row to a multiple
// of `row_alignment`.  The CPU spmv() is the reference's OpenMP kernel restated; it exists so
// the CLI keeps its CPU kernel types (BASELINE configs[0]) and is never a fallback for the
// GPU kernels.
#pragma once

#include "aligned-vector.hpp"

#include <cstdint>
#include <iosfwd>

namespace matrix_market { class Matrix; }

namespace csr_matrix {

typedef int32_t size_type;
typedef int32_t index_type;
typedef double value_type;
typedef aligned_vector<size_type> size_array_type;
typedef aligned_vector<index_type> index_array_type;
typedef aligned_vector<value_type> value_array_type;

struct Matrix
{
    Matrix() = default;
    Matrix(index_type rows, index_type columns, size_type num_entries, index_type row_alignment,
           size_array_type row_ptr, index_array_type column_index, value_array_type value);
    Matrix(Matrix const &) = delete;
    Matrix & operator=(Matrix const &) = delete;
    Matrix(Matrix &&) = default;
    Matrix & operator=(Matrix &&) = default;

    std::size_t size() const;       // bytes of row_ptr + column_index + value (JSON "matrix_size")
    std::size_t value_size() const;
    std::size_t index_size() const;
    // rows / stored entries of thread `thread` under the static ceil(rows/T) partition
    index_type spmv_rows_per_thread(int thread, int num_threads) const;
    size_type spmv_nonzeros_per_thread(int thread, int num_threads) const;

    index_type rows = 0;
    index_type columns = 0;
    size_type num_entries = 0; // entries of the Matrix Market file (padding not counted)
    index_type row_alignment = 1;
    size_array_type row_ptr;
    index_array_type column_index;
    value_array_type value;
};

bool operator==(Matrix const & a, Matrix const & b);

Matrix from_matrix_market(matrix_market::Matrix const & m);
Matrix from_matrix_market_row_aligned(matrix_market::Matrix const & m, index_type row_alignment);

// y += A*x.  An orphaned OpenMP worksharing loop, like the reference's: call it from every
// thread of a parallel region (or from serial code, where it runs on the calling thread).
void spmv(Matrix const & A, value_array_type const & x, value_array_type & y, index_type chunk_size = 0);

// y = A*x (fresh zero y); throws matrix_error on a size mismatch.
value_array_type operator*(Matrix const & A, value_array_type const & x);

} // namespace csr_matrix
