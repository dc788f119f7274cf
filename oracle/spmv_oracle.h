/*
 * spmv_oracle.h -- CPU oracle for the SpMV hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference's CSR / COO / ELLPACK
 * `y += A*x` loops and of the Matrix-Market -> CSR/COO/ELL converters, written
 * to be bit-identical with the reference library when both are built without
 * FMA contraction (the reference Makefile builds with plain `-O3`, x86-64
 * baseline, i.e. multiply-then-add; this file is built with
 * `-ffp-contract=off`).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It is the checker, never the product: nothing under
 * spmv-cache-trace_amd/ links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against (a) the reference's own known-answer tests and the poisson2D golden
 * fixture (tests/golden/), and (b) when oracle/_ref/libref_spmv.so is present,
 * against the reference library itself compiled from /root/reference.
 *
 * All indices are int32 and values are double, as in the reference
 * (src/matrix/csr-matrix.hpp:15-17).  Citations are relative to
 * /root/reference/.
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SpMV loops ------------------------------------------------------- */

/* src/matrix/csr-matrix-spmv.cpp:21-33 (inner loop), :63-76 (row loop),
 * :148-167 (chunk = ceil(rows / num_threads), static schedule). */
void oracle_csr_spmv(int32_t rows, const int32_t *row_ptr,
                     const int32_t *column_index, const double *value,
                     const double *x, double *y, int num_threads);

/* src/matrix/coo-matrix.cpp:248-285 and :313-335.  `workspace` must hold
 * num_threads*rows doubles when num_threads > 1 and is NOT zeroed here
 * (the reference zeroes it once at allocation only). */
void oracle_coo_spmv(int num_threads, int32_t rows, int32_t num_entries,
                     const int32_t *row_index, const int32_t *column_index,
                     const double *value, const double *x, double *y,
                     double *workspace);

/* src/matrix/coo-matrix.cpp:287-309 (num_threads==1 branch is the serial
 * loop; with threads the adds are atomic and the order is unspecified). */
void oracle_coo_spmv_atomic(int num_threads, int32_t rows, int32_t num_entries,
                            const int32_t *row_index,
                            const int32_t *column_index, const double *value,
                            const double *x, double *y);

/* src/matrix/ell-matrix.cpp:243-258 (inner loop, k = i*row_length + l),
 * :260-273 (row loop), :311-335 (chunking).  Row-major padded storage. */
void oracle_ell_spmv(int32_t rows, int32_t row_length,
                     const int32_t *column_index, const double *value,
                     const double *x, double *y, int num_threads);

/* ---- Converters (1-based coordinate entries in, 0-based formats out) --- */

/* Stable row-major sort permutation of coordinate entries by (i, j)
 * (src/matrix/matrix-market.cpp:897-929 uses std::sort, which leaves the
 * relative order of duplicate (i,j) unspecified; the oracle is stable). */
void oracle_sort_row_major(int32_t num_entries, const int32_t *i,
                           const int32_t *j, int32_t *perm);

/* src/matrix/csr-matrix.cpp:193-243.  row_ptr has rows+1 entries.  Call once
 * with column_index == NULL to get row_ptr (and thereby the padded entry
 * count row_ptr[rows]), then again with arrays of that size.
 * Returns row_ptr[rows]. */
int32_t oracle_csr_from_coordinate(int32_t rows, int32_t num_entries,
                                   const int32_t *i, const int32_t *j,
                                   const double *a, int32_t row_alignment,
                                   int32_t *row_ptr, int32_t *column_index,
                                   double *value);

/* src/matrix/coo-matrix.cpp:220-243: file order kept, indices made 0-based. */
void oracle_coo_from_coordinate(int32_t num_entries, const int32_t *i,
                                const int32_t *j, const double *a,
                                int32_t *row_index, int32_t *column_index,
                                double *value);

/* src/matrix/matrix-market.cpp:282-307: longest row. */
int32_t oracle_max_row_length(int32_t rows, int32_t num_entries,
                              const int32_t *i);

/* src/matrix/ell-matrix.cpp:190-238.  Returns 0 on success, -1 when
 * rows*row_length overflows int32 (:199-205), -2 when row 0 is empty (the
 * reference reads column_indices[-1] there: undefined behaviour, which the
 * oracle refuses to imitate).  column_index/value hold rows*row_length
 * entries; padding value 0.0, padding column = last real column of the row
 * seen so far (:226-229), or INT32_MAX with skip_padding. */
int oracle_ell_from_coordinate(int32_t rows, int32_t num_entries,
                               const int32_t *i, const int32_t *j,
                               const double *a, int skip_padding,
                               int32_t row_length, int32_t *column_index,
                               double *value);

/* ---- Hybrid ELLPACK + COO ---------------------------------- */

/* src/matrix/hybrid-matrix.cpp:316-417.  ELL row length = the "2/3 median" of the row
 * lengths (:337-344); rows shorter than that are padded (value 0.0, column = column of the
 * entry consumed last, 0 if none yet, or INT32_MAX with skip_padding), the entries of longer
 * rows beyond it go to a COO remainder in (row, column) order.
 * Call with ell_column_index == NULL to get the sizes only.
 * sizes[0] = ell_row_length, sizes[1] = rows*ell_row_length, sizes[2] = num_coo_entries.
 * Returns 0, or -1 when rows*ell_row_length overflows int32 (:348-353). */
int oracle_hybrid_from_coordinate(int32_t rows, int32_t num_entries, const int32_t *i,
                                  const int32_t *j, const double *a, int skip_padding,
                                  int32_t *sizes, int32_t *ell_column_index, double *ell_value,
                                  int32_t *coo_row_index, int32_t *coo_column_index,
                                  double *coo_value);

/* src/matrix/hybrid-matrix.cpp:535-567: the ELL part (static row blocks), then the COO
 * remainder exactly like coo_spmv but with chunk = ceil(rows / num_threads); workspace as for
 * oracle_coo_spmv (never re-zeroed). */
void oracle_hybrid_spmv(int num_threads, int32_t rows, int32_t ell_row_length,
                        const int32_t *ell_column_index, const double *ell_value,
                        int skip_padding, int32_t num_coo_entries, const int32_t *coo_row_index,
                        const int32_t *coo_column_index, const double *coo_value,
                        const double *x, double *y, double *workspace);

/* ---- Sample statistics (src/util/sample.hpp:11-135) -------------------- */
/* out[0..7] = min, max, mean, median, variance, standard_deviation,
 * skewness, kurtosis of n int64 samples (NaN where the reference gives NaN). */
void oracle_sample_stats(const int64_t *v, int64_t n, double *out);

#ifdef __cplusplus
}
#endif

#endif
