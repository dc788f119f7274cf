/*
 * spmv_oracle.c -- CPU oracle for the SpMV hot path (TEST INFRASTRUCTURE ONLY).
 *
 * See spmv_oracle.h for scope, parity status and the "who may call this" rule.
 * Build: gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/).
 */
#include "spmv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * SpMV loops
 * ---------------------------------------------------------------------- */

/* src/matrix/csr-matrix-spmv.cpp:21-33: z accumulates left to right over the
 * row's entries, then y[i] += z (accumulate, not overwrite). */
static inline void csr_row(int32_t i, const int32_t *p, const int32_t *j,
                           const double *a, const double *x, double *y)
{
    double z = 0.0;
    for (int32_t k = p[i]; k < p[i + 1]; ++k)
        z += a[k] * x[j[k]];
    y[i] += z;
}

void oracle_csr_spmv(int32_t rows, const int32_t *row_ptr,
                     const int32_t *column_index, const double *value,
                     const double *x, double *y, int num_threads)
{
    if (num_threads < 1)
        num_threads = 1;
    /* src/matrix/csr-matrix-spmv.cpp:154-161 */
    int32_t chunk = (rows + num_threads - 1) / num_threads;
    if (chunk < 1)
        chunk = 1;
    if (num_threads == 1) {
        for (int32_t i = 0; i < rows; ++i)
            csr_row(i, row_ptr, column_index, value, x, y);
        return;
    }
    /* src/matrix/csr-matrix-spmv.cpp:72-75, run from inside the caller's
     * parallel region (src/profile-kernel.cpp:227). */
#pragma omp parallel num_threads(num_threads)
    {
#pragma omp for nowait schedule(static, chunk)
        for (int32_t i = 0; i < rows; ++i)
            csr_row(i, row_ptr, column_index, value, x, y);
    }
}

void oracle_coo_spmv(int num_threads, int32_t rows, int32_t num_entries,
                     const int32_t *row_index, const int32_t *column_index,
                     const double *value, const double *x, double *y,
                     double *workspace)
{
    if (num_threads <= 1) {
        /* src/matrix/coo-matrix.cpp:266-269 */
        for (int32_t k = 0; k < num_entries; ++k)
            y[row_index[k]] += value[k] * x[column_index[k]];
        return;
    }
    /* src/matrix/coo-matrix.cpp:321-323 */
    int32_t chunk = (num_entries + num_threads - 1) / num_threads;
    if (chunk < 1)
        chunk = 1;
#pragma omp parallel num_threads(num_threads)
    {
#ifdef _OPENMP
        size_t thread = (size_t)omp_get_thread_num();
#else
        size_t thread = 0;
#endif
        /* src/matrix/coo-matrix.cpp:272-275: private partial vector */
#pragma omp for schedule(static, chunk)
        for (int32_t k = 0; k < num_entries; ++k)
            workspace[thread * (size_t)rows + row_index[k]] +=
                value[k] * x[column_index[k]];
        /* src/matrix/coo-matrix.cpp:278-283: row-parallel reduction; the
         * reference reuses the nnz chunk size for this loop too. */
#pragma omp for schedule(static, chunk)
        for (int32_t i = 0; i < rows; i++)
            for (int32_t t = 0; t < num_threads; t++)
                y[i] += workspace[(size_t)t * (size_t)rows + i];
    }
}

void oracle_coo_spmv_atomic(int num_threads, int32_t rows, int32_t num_entries,
                            const int32_t *row_index,
                            const int32_t *column_index, const double *value,
                            const double *x, double *y)
{
    (void)rows;
    if (num_threads <= 1) {
        /* src/matrix/coo-matrix.cpp:298-301 */
        for (int32_t k = 0; k < num_entries; ++k)
            y[row_index[k]] += value[k] * x[column_index[k]];
        return;
    }
    int32_t chunk = (num_entries + num_threads - 1) / num_threads;
    if (chunk < 1)
        chunk = 1;
    /* src/matrix/coo-matrix.cpp:303-307 */
#pragma omp parallel for num_threads(num_threads) schedule(static, chunk)
    for (int32_t k = 0; k < num_entries; ++k) {
        double t = value[k] * x[column_index[k]];
#pragma omp atomic
        y[row_index[k]] += t;
    }
}

/* src/matrix/ell-matrix.cpp:243-258 */
static inline void ell_row(int32_t i, int32_t row_length, const int32_t *j,
                           const double *a, const double *x, double *y)
{
    double z = 0.0;
    for (int32_t l = 0; l < row_length; ++l) {
        /* the reference forms k = i*row_length + l in int32; the converter
         * guarantees rows*row_length fits, so size_t gives the same k. */
        size_t k = (size_t)i * (size_t)row_length + (size_t)l;
        z += a[k] * x[j[k]];
    }
    y[i] += z;
}

void oracle_ell_spmv(int32_t rows, int32_t row_length,
                     const int32_t *column_index, const double *value,
                     const double *x, double *y, int num_threads)
{
    if (num_threads < 1)
        num_threads = 1;
    /* src/matrix/ell-matrix.cpp:317-325 */
    int32_t chunk = (rows + num_threads - 1) / num_threads;
    if (chunk < 1)
        chunk = 1;
    if (num_threads == 1) {
        for (int32_t i = 0; i < rows; ++i)
            ell_row(i, row_length, column_index, value, x, y);
        return;
    }
    /* src/matrix/ell-matrix.cpp:269-272 */
#pragma omp parallel num_threads(num_threads)
    {
#pragma omp for nowait schedule(static, chunk)
        for (int32_t i = 0; i < rows; ++i)
            ell_row(i, row_length, column_index, value, x, y);
    }
}

/* ------------------------------------------------------------------------
 * Converters
 * ---------------------------------------------------------------------- */

typedef struct {
    int32_t i, j, k;
} sort_key;

static int cmp_key(const void *pa, const void *pb)
{
    const sort_key *a = (const sort_key *)pa, *b = (const sort_key *)pb;
    if (a->i != b->i)
        return a->i < b->i ? -1 : 1;
    if (a->j != b->j)
        return a->j < b->j ? -1 : 1;
    /* tie-break on the original position => stable */
    return a->k < b->k ? -1 : (a->k > b->k ? 1 : 0);
}

/* src/matrix/matrix-market.cpp:897-929 (comparison std::tie(i,j)). */
void oracle_sort_row_major(int32_t num_entries, const int32_t *i,
                           const int32_t *j, int32_t *perm)
{
    sort_key *keys = (sort_key *)malloc(sizeof(sort_key) *
                                        (size_t)(num_entries > 0 ? num_entries : 1));
    for (int32_t k = 0; k < num_entries; ++k) {
        keys[k].i = i[k];
        keys[k].j = j[k];
        keys[k].k = k;
    }
    qsort(keys, (size_t)num_entries, sizeof(sort_key), cmp_key);
    for (int32_t k = 0; k < num_entries; ++k)
        perm[k] = keys[k].k;
    free(keys);
}

int32_t oracle_csr_from_coordinate(int32_t rows, int32_t num_entries,
                                   const int32_t *i, const int32_t *j,
                                   const double *a, int32_t row_alignment,
                                   int32_t *row_ptr, int32_t *column_index,
                                   double *value)
{
    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) *
                                      (size_t)(num_entries > 0 ? num_entries : 1));
    /* src/matrix/csr-matrix.cpp:200-204 */
    oracle_sort_row_major(num_entries, i, j, perm);

    /* src/matrix/csr-matrix.cpp:206-217: row lengths incl. alignment padding */
    int32_t k = 0, l = 0;
    row_ptr[0] = 0;
    for (int32_t r = 0; r < rows; ++r) {
        while (l < num_entries && i[perm[l]] - 1 == r) {
            l++;
            k++;
        }
        k = ((k + (row_alignment - 1)) / row_alignment) * row_alignment;
        row_ptr[r + 1] = k;
    }

    if (column_index && value) {
        /* src/matrix/csr-matrix.cpp:219-237 */
        k = 0;
        l = 0;
        for (int32_t r = 0; r < rows; ++r) {
            while (l < num_entries && i[perm[l]] - 1 == r) {
                column_index[k] = j[perm[l]] - 1;
                value[k] = a[perm[l]];
                ++k;
                ++l;
            }
            while (k < row_ptr[r + 1]) {
                column_index[k] = 0;
                value[k] = 0.0;
                ++k;
            }
        }
    }
    free(perm);
    return row_ptr[rows];
}

void oracle_coo_from_coordinate(int32_t num_entries, const int32_t *i,
                                const int32_t *j, const double *a,
                                int32_t *row_index, int32_t *column_index,
                                double *value)
{
    /* src/matrix/coo-matrix.cpp:226-239 */
    for (int32_t k = 0; k < num_entries; k++) {
        row_index[k] = i[k] - 1;
        column_index[k] = j[k] - 1;
        value[k] = a[k];
    }
}

int32_t oracle_max_row_length(int32_t rows, int32_t num_entries,
                              const int32_t *i)
{
    /* src/matrix/matrix-market.cpp:282-307 */
    int32_t *len = (int32_t *)calloc((size_t)(rows > 0 ? rows : 1), sizeof(int32_t));
    for (int32_t k = 0; k < num_entries; ++k)
        ++len[i[k] - 1];
    int32_t m = 0;
    for (int32_t r = 0; r < rows; ++r)
        if (len[r] > m)
            m = len[r];
    free(len);
    return m;
}

int oracle_ell_from_coordinate(int32_t rows, int32_t num_entries,
                               const int32_t *i, const int32_t *j,
                               const double *a, int skip_padding,
                               int32_t row_length, int32_t *column_index,
                               double *value)
{
    /* src/matrix/ell-matrix.cpp:199-205 */
    int32_t padded;
    if (__builtin_mul_overflow(rows, row_length, &padded))
        return -1;

    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) *
                                      (size_t)(num_entries > 0 ? num_entries : 1));
    /* src/matrix/ell-matrix.cpp:208-212 */
    oracle_sort_row_major(num_entries, i, j, perm);

    /* src/matrix/ell-matrix.cpp:214-233 */
    int32_t k = 0, l = 0;
    for (int32_t r = 0; r < rows; ++r) {
        while (k < num_entries && i[perm[k]] - 1 == r) {
            column_index[l] = j[perm[k]] - 1;
            value[l] = a[perm[k]];
            ++k;
            ++l;
        }
        while (l < (r + 1) * row_length) {
            if (skip_padding) {
                column_index[l] = INT32_MAX;
            } else {
                if (k == 0) { /* reference reads column_indices[-1] here */
                    free(perm);
                    return -2;
                }
                column_index[l] = j[perm[k - 1]] - 1;
            }
            value[l] = 0.0;
            ++l;
        }
    }
    free(perm);
    return 0;
}

/* ------------------------------------------------------------------------
 * Hybrid ELLPACK + COO
 * ---------------------------------------------------------------------- */

int oracle_hybrid_from_coordinate(int32_t rows, int32_t num_entries, const int32_t *i,
                                  const int32_t *j, const double *a, int skip_padding,
                                  int32_t *sizes, int32_t *ell_column_index, double *ell_value,
                                  int32_t *coo_row_index, int32_t *coo_column_index,
                                  double *coo_value)
{
    /* src/matrix/hybrid-matrix.cpp:329-335: histogram of row lengths */
    int32_t *len = (int32_t *)calloc((size_t)(rows > 0 ? rows : 1), sizeof(int32_t));
    for (int32_t k = 0; k < num_entries; ++k)
        ++len[i[k] - 1];
    int32_t max_len = 0;
    for (int32_t r = 0; r < rows; ++r)
        if (len[r] > max_len)
            max_len = len[r];
    int32_t *hist = (int32_t *)calloc((size_t)max_len + 1, sizeof(int32_t));
    for (int32_t r = 0; r < rows; ++r)
        hist[len[r]]++;
    /* :337-344: smallest length such that at least 2/3 of the rows are not longer */
    int32_t median = 0, below = 0;
    while (below < (2 * rows) / 3) {
        below += hist[median];
        median++;
    }
    median = (median == 0) ? 0 : median - 1;
    int32_t L = median, n_ell;
    if (__builtin_mul_overflow(rows, L, &n_ell)) {
        free(len);
        free(hist);
        return -1;
    }
    /* :357-360 */
    int32_t n_coo = 0;
    for (int32_t l = L + 1; l <= max_len; l++)
        n_coo += hist[l] * (l - L);
    free(hist);
    sizes[0] = L;
    sizes[1] = n_ell;
    sizes[2] = n_coo;
    if (!ell_column_index) {
        free(len);
        return 0;
    }

    int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_entries > 0 ? num_entries : 1));
    oracle_sort_row_major(num_entries, i, j, perm);
    /* :376-409 */
    int32_t k = 0, e = 0, c = 0;
    for (int32_t r = 0; r < rows; ++r) {
        if (len[r] < L) {
            for (int32_t q = 0; q < len[r]; q++) {
                ell_column_index[e] = j[perm[k]] - 1;
                ell_value[e] = a[perm[k]];
                e++;
                k++;
            }
            for (int32_t q = len[r]; q < L; q++) {
                ell_column_index[e] = skip_padding ? INT32_MAX : (k > 0 ? j[perm[k - 1]] - 1 : 0);
                ell_value[e] = 0.0;
                e++;
            }
        } else {
            for (int32_t q = 0; q < L; q++) {
                ell_column_index[e] = j[perm[k]] - 1;
                ell_value[e] = a[perm[k]];
                e++;
                k++;
            }
            for (int32_t q = L; q < len[r]; q++) {
                coo_row_index[c] = i[perm[k]] - 1;
                coo_column_index[c] = j[perm[k]] - 1;
                coo_value[c] = a[perm[k]];
                c++;
                k++;
            }
        }
    }
    free(perm);
    free(len);
    return 0;
}

void oracle_hybrid_spmv(int num_threads, int32_t rows, int32_t L,
                        const int32_t *ej, const double *ea, int skip_padding,
                        int32_t n_coo, const int32_t *cr, const int32_t *cc, const double *cv,
                        const double *x, double *y, double *workspace)
{
    if (num_threads < 1)
        num_threads = 1;
    /* src/matrix/hybrid-matrix.cpp:543-545: ONE chunk size, from the rows, for all loops */
    int32_t chunk = (rows + num_threads - 1) / num_threads;
    if (chunk < 1)
        chunk = 1;
#pragma omp parallel num_threads(num_threads)
    {
#ifdef _OPENMP
        size_t thread = (size_t)omp_get_thread_num();
#else
        size_t thread = 0;
#endif
        /* :547-555 (inner loops :422-437 / :453-470) */
#pragma omp for nowait schedule(static, chunk)
        for (int32_t r = 0; r < rows; ++r) {
            double z = 0.0;
            for (int32_t l = 0; l < L; ++l) {
                size_t k = (size_t)r * (size_t)L + (size_t)l;
                if (skip_padding && ej[k] == INT32_MAX)
                    break;
                z += ea[k] * x[ej[k]];
            }
            y[r] += z;
        }
        /* :557-566 -> :490-527 */
        if (num_threads == 1) {
            for (int32_t k = 0; k < n_coo; ++k)
                y[cr[k]] += cv[k] * x[cc[k]];
        } else {
#pragma omp for schedule(static, chunk)
            for (int32_t k = 0; k < n_coo; ++k)
                workspace[thread * (size_t)rows + cr[k]] += cv[k] * x[cc[k]];
#pragma omp for schedule(static, chunk)
            for (int32_t r = 0; r < rows; r++)
                for (int32_t t = 0; t < num_threads; t++)
                    y[r] += workspace[(size_t)t * (size_t)rows + r];
        }
    }
}

/* ------------------------------------------------------------------------
 * Sample statistics (src/util/sample.hpp)
 * ---------------------------------------------------------------------- */

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static double moment(const int64_t *v, int64_t n, double mu, int order)
{
    /* src/util/sample.hpp:56-93: population moments, divisor n */
    double m = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double d = (double)v[i] - mu;
        double t = d;
        for (int o = 1; o < order; o++)
            t = t * d;
        m = m + t;
    }
    return m / (double)n;
}

void oracle_sample_stats(const int64_t *v, int64_t n, double *out)
{
    /* src/util/sample.hpp:11-29 */
    int64_t mn = INT64_MAX, mx = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        if (v[i] < mn)
            mn = v[i];
        if (v[i] > mx)
            mx = v[i];
    }
    out[0] = (double)mn;
    out[1] = (double)mx;
    if (n == 0) {
        for (int q = 2; q < 8; q++)
            out[q] = NAN;
        return;
    }
    /* src/util/sample.hpp:31-41 */
    double mu = 0.0;
    for (int64_t i = 0; i < n; i++)
        mu = mu + (double)v[i];
    mu = mu / (double)n;
    out[2] = mu;
    /* src/util/sample.hpp:43-54: `n % 1 == 0` is always true => sorted[n/2] */
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(s, v, sizeof(int64_t) * (size_t)n);
    qsort(s, (size_t)n, sizeof(int64_t), cmp_i64);
    out[3] = (double)s[n / 2];
    free(s);
    /* src/util/sample.hpp:95-106: divisor n-1 (NaN for n == 1) */
    double ss = 0.0;
    for (int64_t i = 0; i < n; i++)
        ss = ss + ((double)v[i] - mu) * ((double)v[i] - mu);
    double var = ss / (double)(n - 1);
    out[4] = var;
    /* :108-115 */
    out[5] = sqrt(var);
    /* :117-125 */
    out[6] = moment(v, n, mu, 3) / sqrt(var * var * var);
    /* :127-135 */
    double m2 = moment(v, n, mu, 2);
    out[7] = moment(v, n, mu, 4) / (m2 * m2);
}
