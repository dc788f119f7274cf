#include "ell-matrix.hpp"

#include "matrix-error.hpp"
#include "matrix-market.hpp"

#include <algorithm>
#include <limits>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace ell_matrix {

Matrix::Matrix(index_type rows_, index_type columns_, size_type num_entries_, index_type row_length_,
               index_array_type column_index_, value_array_type value_, bool skip_padding_)
    : rows(rows_)
    , columns(columns_)
    , num_entries(num_entries_)
    , row_length(row_length_)
    , column_index(std::move(column_index_))
    , value(std::move(value_))
    , skip_padding(skip_padding_)
{
}

std::size_t Matrix::value_size() const { return sizeof(value_type) * value.size(); }
std::size_t Matrix::index_size() const { return sizeof(index_type) * column_index.size(); }
std::size_t Matrix::size() const { return value_size() + index_size(); }
size_type Matrix::num_padding_entries() const { return (size_type) value.size() - num_entries; }

bool operator==(Matrix const & a, Matrix const & b)
{
    return a.rows == b.rows && a.columns == b.columns && a.num_entries == b.num_entries &&
        a.row_length == b.row_length && a.column_index == b.column_index && a.value == b.value;
}

Matrix from_matrix_market(matrix_market::Matrix const & m, bool skip_padding)
{
    if (m.format() != matrix_market::Format::coordinate)
        throw matrix::matrix_error("Expected matrix in coordinate format");
    index_type const rows = m.rows();
    auto const len = m.row_lengths();
    index_type const row_length = len.empty() ? 0 : *std::max_element(len.begin(), len.end());
    size_type padded;
    if (__builtin_mul_overflow(rows, row_length, &padded))
        throw matrix::matrix_error(
            "Failed to convert to ELLPACK: Integer overflow when computing number of non-zeros");

    matrix_market::RowMajorEntries const e = matrix_market::row_major_entries(m);

    index_array_type column_index((std::size_t) padded, 0);
    value_array_type value((std::size_t) padded, 0.0);
    // the pad column of a row is the column of the last entry BEFORE the padding in (row, column)
    // order: the row's own last entry, or that of the nearest non-empty row above it
    std::vector<index_type> last_before((std::size_t) rows + 1, 0);
    for (index_type r = 0; r < rows; ++r)
        last_before[(std::size_t) r + 1] = e.start[(std::size_t) r + 1] > e.start[(std::size_t) r]
            ? e.col[e.start[(std::size_t) r + 1] - 1] : last_before[(std::size_t) r];
#pragma omp parallel for schedule(static) if ((std::size_t) rows * (std::size_t) row_length > (1u << 16))
    for (long long r = 0; r < (long long) rows; ++r) {
        std::size_t dst = (std::size_t) r * (std::size_t) row_length;
        std::size_t const b = e.start[(std::size_t) r], n = e.start[(std::size_t) r + 1] - b;
        for (std::size_t q = 0; q < n; ++q, ++dst) {
            column_index[dst] = e.col[b + q];
            value[dst] = e.val[b + q];
        }
        index_type const pad = skip_padding ? std::numeric_limits<index_type>::max() : last_before[(std::size_t) r + 1];
        for (std::size_t q = n; q < (std::size_t) row_length; ++q, ++dst)
            column_index[dst] = pad;
    }
    return Matrix(rows, m.columns(), m.num_entries(), row_length, std::move(column_index),
                  std::move(value), skip_padding);
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
    index_type const L = A.row_length;
    index_type const * const j = A.column_index.data();
    value_type const * const a = A.value.data();
    value_type const * const xv = x.data();
    value_type * const yv = y.data();
    bool const stop_at_sentinel = A.skip_padding;
#pragma omp for nowait schedule(static, chunk_size)
    for (index_type i = 0; i < A.rows; ++i) {
        std::size_t const base = (std::size_t) i * (std::size_t) L;
        value_type z = 0.0;
        for (index_type l = 0; l < L; ++l) {
            if (stop_at_sentinel && j[base + l] == std::numeric_limits<index_type>::max())
                break;
            z += a[base + l] * xv[j[base + l]];
        }
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

} // namespace ell_matrix
