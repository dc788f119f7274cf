#include "hybrid-matrix.hpp"

#include "matrix-error.hpp"
#include "matrix-market.hpp"

#include <algorithm>
#include <limits>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace hybrid_matrix {

std::size_t Matrix::size() const
{
    return sizeof(value_type) * (ell_value.size() + coo_value.size()) +
        sizeof(index_type) * (ell_column_index.size() + coo_column_index.size());
}

Matrix from_matrix_market(matrix_market::Matrix const & m, bool ell_skip_padding)
{
    if (m.format() != matrix_market::Format::coordinate)
        throw matrix::matrix_error("Expected matrix in coordinate format");
    index_type const rows = m.rows();
    auto const len = m.row_lengths();
    index_type const longest = len.empty() ? 0 : *std::max_element(len.begin(), len.end());

    // ELL row length: walk the histogram of row lengths until two thirds of the rows are covered
    std::vector<index_type> hist((std::size_t) longest + 1, 0);
    for (index_type l : len)
        ++hist[(std::size_t) l];
    index_type width = 0, covered = 0;
    while (covered < (2 * rows) / 3) {
        covered += hist[(std::size_t) width];
        ++width;
    }
    width = width == 0 ? 0 : width - 1;

    size_type n_ell;
    if (__builtin_mul_overflow(rows, width, &n_ell))
        throw matrix::matrix_error("Failed to convert to HYBRID: Integer overflow when computing number of non-zeros");
    long long n_coo = 0;
    for (index_type l = width + 1; l <= longest; ++l)
        n_coo += (long long) hist[(std::size_t) l] * (l - width);

    matrix_market::RowMajorEntries const e = matrix_market::row_major_entries(m);

    Matrix A;
    A.rows = rows;
    A.columns = m.columns();
    A.num_entries = m.num_entries();
    A.ell_row_length = width;
    A.num_ell_entries = n_ell;
    A.ell_skip_padding = ell_skip_padding;
    A.num_coo_entries = (size_type) n_coo;
    A.ell_column_index.assign((std::size_t) n_ell, 0);
    A.ell_value.assign((std::size_t) n_ell, 0.0);
    A.coo_row_index.assign((std::size_t) n_coo, 0);
    A.coo_column_index.assign((std::size_t) n_coo, 0);
    A.coo_value.assign((std::size_t) n_coo, 0.0);

    std::size_t spill = 0;      // COO entries written
    index_type last_column = 0; // column of the entry consumed last (0 before the first)
    for (index_type r = 0; r < rows; ++r) {
        std::size_t dst = (std::size_t) r * (std::size_t) width;
        std::size_t const b = e.start[(std::size_t) r];
        index_type const n = len[(std::size_t) r];
        index_type const in_ell = std::min(n, width);
        for (index_type q = 0; q < in_ell; ++q, ++dst) {
            last_column = e.col[b + (std::size_t) q];
            A.ell_column_index[dst] = last_column;
            A.ell_value[dst] = e.val[b + (std::size_t) q];
        }
        for (index_type q = in_ell; q < width; ++q, ++dst)
            A.ell_column_index[dst] = ell_skip_padding ? std::numeric_limits<index_type>::max() : last_column;
        for (index_type q = width; q < n; ++q, ++spill) {
            last_column = e.col[b + (std::size_t) q];
            A.coo_row_index[spill] = r;
            A.coo_column_index[spill] = last_column;
            A.coo_value[spill] = e.val[b + (std::size_t) q];
        }
    }
    return A;
}

void spmv(int num_threads, Matrix const & A, value_array_type const & x, value_array_type & y,
          value_array_type & workspace, index_type chunk_size)
{
    if (chunk_size <= 0)
        chunk_size = std::max<index_type>(1, (A.rows + num_threads - 1) / num_threads);
    index_type const L = A.ell_row_length;
    index_type const * const ej = A.ell_column_index.data();
    value_type const * const ea = A.ell_value.data();
    value_type const * const xv = x.data();
    value_type * const yv = y.data();
    bool const stop = A.ell_skip_padding;
#pragma omp for nowait schedule(static, chunk_size)
    for (index_type i = 0; i < A.rows; ++i) {
        std::size_t const base = (std::size_t) i * (std::size_t) L;
        value_type z = 0.0;
        for (index_type l = 0; l < L; ++l) {
            if (stop && ej[base + l] == std::numeric_limits<index_type>::max())
                break;
            z += ea[base + l] * xv[ej[base + l]];
        }
        yv[i] += z;
    }

    index_type const * const cr = A.coo_row_index.data();
    index_type const * const cc = A.coo_column_index.data();
    value_type const * const cv = A.coo_value.data();
    if (num_threads == 1) {
        for (size_type k = 0; k < A.num_coo_entries; ++k)
            yv[cr[k]] += cv[k] * xv[cc[k]];
        return;
    }
#ifdef _OPENMP
    std::size_t const me = (std::size_t) omp_get_thread_num();
#else
    std::size_t const me = 0;
#endif
    value_type * const mine = workspace.data() + me * (std::size_t) A.rows;
#pragma omp for schedule(static, chunk_size)
    for (size_type k = 0; k < A.num_coo_entries; ++k)
        mine[cr[k]] += cv[k] * xv[cc[k]];
#pragma omp for schedule(static, chunk_size)
    for (index_type i = 0; i < A.rows; ++i)
        for (int t = 0; t < num_threads; ++t)
            yv[i] += workspace[(std::size_t) t * (std::size_t) A.rows + (std::size_t) i];
}

} // namespace hybrid_matrix
