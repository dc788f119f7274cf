#include "csr-matrix.hpp"

#include "matrix-error.hpp"
#include "matrix-market.hpp"

#include <algorithm>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace csr_matrix {

Matrix::Matrix(index_type rows_, index_type columns_, size_type num_entries_, index_type row_alignment_,
               size_array_type row_ptr_, index_array_type column_index_, value_array_type value_)
    : rows(rows_)
    , columns(columns_)
    , num_entries(num_entries_)
    , row_alignment(row_alignment_)
    , row_ptr(std::move(row_ptr_))
    , column_index(std::move(column_index_))
    , value(std::move(value_))
{
}

std::size_t Matrix::value_size() const { return sizeof(value_type) * value.size(); }
std::size_t Matrix::index_size() const
{
    return sizeof(size_type) * row_ptr.size() + sizeof(index_type) * column_index.size();
}
std::size_t Matrix::size() const { return value_size() + index_size(); }

index_type Matrix::spmv_rows_per_thread(int thread, int num_threads) const
{
    index_type const chunk = (rows + num_threads - 1) / num_threads;
    return std::min(rows, (thread + 1) * chunk) - std::min(rows, thread * chunk);
}

size_type Matrix::spmv_nonzeros_per_thread(int thread, int num_threads) const
{
    index_type const chunk = (rows + num_threads - 1) / num_threads;
    return row_ptr[std::min(rows, (thread + 1) * chunk)] - row_ptr[std::min(rows, thread * chunk)];
}

bool operator==(Matrix const & a, Matrix const & b)
{
    return a.rows == b.rows && a.columns == b.columns && a.num_entries == b.num_entries &&
        a.row_ptr == b.row_ptr && a.column_index == b.column_index && a.value == b.value;
}

Matrix from_matrix_market(matrix_market::Matrix const & m)
{
    return from_matrix_market_row_aligned(m, 1);
}

Matrix from_matrix_market_row_aligned(matrix_market::Matrix const & m, index_type row_alignment)
{
    if (m.format() != matrix_market::Format::coordinate)
        throw matrix::matrix_error("Expected matrix in coordinate format");
    if (row_alignment < 1)
        throw matrix::matrix_error("Expected a positive row alignment");

    // entries in (row, column) order; duplicates are kept as separate entries
    matrix_market::RowMajorEntries const e = matrix_market::row_major_entries(m);
    index_type const rows = m.rows();

    // padded row lengths -> row_ptr
    size_array_type row_ptr((std::size_t) rows + 1, 0);
    long long k = 0;
    for (index_type r = 0; r < rows; ++r) {
        k += (long long) (e.start[(std::size_t) r + 1] - e.start[(std::size_t) r]);
        k = ((k + (row_alignment - 1)) / row_alignment) * row_alignment;
        if (k > INT32_MAX)
            throw matrix::matrix_error("Failed to convert to CSR: Integer overflow when computing number of non-zeros");
        row_ptr[(std::size_t) r + 1] = (size_type) k;
    }

    index_array_type column_index((std::size_t) k);
    value_array_type value((std::size_t) k);
#pragma omp parallel for schedule(static) if ((std::size_t) k > (1u << 16))
    for (long long r = 0; r < (long long) rows; ++r) {
        std::size_t dst = (std::size_t) row_ptr[(std::size_t) r];
        for (std::size_t q = e.start[(std::size_t) r]; q < e.start[(std::size_t) r + 1]; ++q, ++dst) {
            column_index[dst] = e.col[q];
            value[dst] = e.val[q];
        }
        for (; dst < (std::size_t) row_ptr[(std::size_t) r + 1]; ++dst) { // alignment padding
            column_index[dst] = 0;
            value[dst] = 0.0;
        }
    }
    return Matrix(rows, m.columns(), m.num_entries(), row_alignment, std::move(row_ptr),
                  std::move(column_index), std::move(value));
}

void spmv(Matrix const & A, value_array_type const & x, value_array_type & y, index_type chunk_size)
{
    if (chunk_size <= 0) {
#ifdef _OPENMP
        int const team = omp_get_num_threads();
#else
        int const team = 1;
#endif
        chunk_size = std::max<index_type>(1, (A.rows + team - 1) / team);
    }
    size_type const * const p = A.row_ptr.data();
    index_type const * const j = A.column_index.data();
    value_type const * const a = A.value.data();
    value_type const * const xv = x.data();
    value_type * const yv = y.data();
    // one contiguous block of rows per thread, no barrier at the end
#pragma omp for nowait schedule(static, chunk_size)
    for (index_type i = 0; i < A.rows; ++i) {
        value_type z = 0.0;
        for (size_type k = p[i]; k < p[i + 1]; ++k)
            z += a[k] * xv[j[k]];
        yv[i] += z;
    }
}

value_array_type operator*(Matrix const & A, value_array_type const & x)
{
    if (A.columns != (index_type) x.size())
        throw matrix::matrix_error("Size mismatch: A.size()=" + std::to_string(A.rows) + "x" +
                                   std::to_string(A.columns) + ", x.size()=" + std::to_string(x.size()));
    value_array_type y((std::size_t) A.rows, 0.0);
    spmv(A, x, y);
    return y;
}

} // namespace csr_matrix
