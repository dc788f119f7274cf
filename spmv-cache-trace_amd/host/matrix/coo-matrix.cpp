#include "coo-matrix.hpp"

#include "matrix-error.hpp"
#include "matrix-market.hpp"

#include <algorithm>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace coo_matrix {

Matrix::Matrix(index_type rows_, index_type columns_, size_type num_entries_, index_array_type row_index_,
               index_array_type column_index_, value_array_type value_)
    : rows(rows_)
    , columns(columns_)
    , num_entries(num_entries_)
    , row_index(std::move(row_index_))
    , column_index(std::move(column_index_))
    , value(std::move(value_))
{
}

std::size_t Matrix::value_size() const { return sizeof(value_type) * value.size(); }
std::size_t Matrix::index_size() const
{
    return sizeof(index_type) * (row_index.size() + column_index.size());
}
std::size_t Matrix::size() const { return value_size() + index_size(); }

bool operator==(Matrix const & a, Matrix const & b)
{
    return a.rows == b.rows && a.columns == b.columns && a.num_entries == b.num_entries &&
        a.row_index == b.row_index && a.column_index == b.column_index && a.value == b.value;
}

Matrix from_matrix_market(matrix_market::Matrix const & m)
{
    if (m.format() != matrix_market::Format::coordinate)
        throw matrix::matrix_error("Expected matrix in coordinate format");
    auto const & ri = m.row_indices();
    auto const & ci = m.column_indices();
    auto const va = m.values_real();
    std::size_t const n = (std::size_t) m.num_entries();
    index_array_type r(n), c(n);
    value_array_type v(n);
    for (std::size_t k = 0; k < n; ++k) {
        if (ri[k] < 1 || ri[k] > m.rows())
            throw matrix::matrix_error("Row index out of bounds: " + std::to_string(ri[k]));
        if (ci[k] < 1 || ci[k] > m.columns())
            throw matrix::matrix_error("Column index out of bounds: " + std::to_string(ci[k]));
        r[k] = ri[k] - 1; // file order kept, 1-based -> 0-based
        c[k] = ci[k] - 1;
        v[k] = va[k];
    }
    return Matrix(m.rows(), m.columns(), m.num_entries(), std::move(r), std::move(c), std::move(v));
}

void spmv(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
          value_array_type & workspace, index_type chunk_size)
{
    if (chunk_size <= 0)
        chunk_size = std::max<index_type>(1, (A.num_entries + num_threads - 1) / num_threads);
    index_type const * const r = A.row_index.data();
    index_type const * const c = A.column_index.data();
    value_type const * const v = A.value.data();
    value_type const * const xv = x.data();
    value_type * const yv = y.data();
    if (num_threads == 1) {
        for (size_type k = 0; k < A.num_entries; ++k)
            yv[r[k]] += v[k] * xv[c[k]];
        return;
    }
#ifdef _OPENMP
    std::size_t const me = (std::size_t) omp_get_thread_num();
#else
    std::size_t const me = 0;
#endif
    value_type * const mine = workspace.data() + me * (std::size_t) A.rows;
#pragma omp for schedule(static, chunk_size)
    for (size_type k = 0; k < A.num_entries; ++k)
        mine[r[k]] += v[k] * xv[c[k]];
    // (implicit barrier) fold the per-thread vectors into y; the reference reuses the entry
    // chunk size for this row loop
#pragma omp for schedule(static, chunk_size)
    for (index_type i = 0; i < A.rows; ++i)
        for (int t = 0; t < num_threads; ++t)
            yv[i] += workspace[(std::size_t) t * (std::size_t) A.rows + (std::size_t) i];
}

void spmv_atomic(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
                 index_type chunk_size)
{
    if (chunk_size <= 0)
        chunk_size = std::max<index_type>(1, (A.num_entries + num_threads - 1) / num_threads);
    index_type const * const r = A.row_index.data();
    index_type const * const c = A.column_index.data();
    value_type const * const v = A.value.data();
    value_type const * const xv = x.data();
    value_type * const yv = y.data();
    if (num_threads == 1) {
        for (size_type k = 0; k < A.num_entries; ++k)
            yv[r[k]] += v[k] * xv[c[k]];
        return;
    }
#pragma omp for schedule(static, chunk_size)
    for (size_type k = 0; k < A.num_entries; ++k) {
        value_type const t = v[k] * xv[c[k]];
#pragma omp atomic
        yv[r[k]] += t;
    }
}

value_array_type operator*(Matrix const & A, value_array_type const & x)
{
    if (A.columns != (index_type) x.size())
        throw matrix::matrix_error("Size mismatch: A.size()=" + std::to_string(A.rows) + "x" +
                                   std::to_string(A.columns) + ", x.size()=" + std::to_string(x.size()));
#ifdef _OPENMP
    int const team = omp_get_num_threads();
#else
    int const team = 1;
#endif
    value_array_type y((std::size_t) A.rows, 0.0);
    value_array_type workspace((std::size_t) team * (std::size_t) A.rows, 0.0);
    spmv(team, A, x, y, workspace);
    return y;
}

} // namespace coo_matrix
