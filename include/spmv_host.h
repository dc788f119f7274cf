/*
 * spmv_host.h -- C ABI of the host library (libspmv_host.so): the input side of the SpMV path.
 *
 * The reference reaches its matrices through matrix_market::load_matrix
 * (src/matrix/matrix-market.cpp:777-861) and the format converters
 * csr_matrix::from_matrix_market (src/matrix/csr-matrix.cpp:187-243),
 * coo_matrix::from_matrix_market (src/matrix/coo-matrix.cpp:220-243),
 * ell_matrix::from_matrix_market (src/matrix/ell-matrix.cpp:190-238) and
 * hybrid_matrix::from_matrix_market (src/matrix/hybrid-matrix.cpp:316-417), all called from
 * Kernel::init (src/kernels/csr-spmv.cpp:26-46).  These entry points give the same arrays to
 * callers that are not C++ (bench.py --matrix, the tests), so that a file goes through the
 * same loader and converter whichever front end multiplies it.
 *
 * `path` is a Matrix Market file (.mtx, .gz, .tgz, .tar.gz, optional __RCM / __GP<n> suffix as in the reference -- without METIS __GP<n> reorders nothing -- or __GPX<n>, this build's own partitioner),
 * or "synthetic:<family>[:<parameters>]" (host/matrix/synthetic.hpp) for a generated matrix.
 * Symmetric files are NOT expanded (the reference multiplies the stored triangle only) unless
 * SPMV_HOST_EXPAND_SYMMETRIC is passed -- an extension, off by default.
 *
 * Conventions: plain C types, opaque handle, 0 = OK / negative = error with the message in
 * spmv_host_last_error(); nothing throws across this boundary.  Arrays belong to the handle and
 * stay valid until spmv_host_matrix_free.
 */
#ifndef SPMV_HOST_H
#define SPMV_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_HOST_OK 0
#define SPMV_HOST_ERR_INVALID (-1) /* bad argument */
#define SPMV_HOST_ERR_MATRIX (-2)  /* the loader or a converter refused the input (matrix_error) */
#define SPMV_HOST_ERR_SYSTEM (-3)  /* I/O or allocation failure */

#define SPMV_HOST_FORMAT_CSR 1
#define SPMV_HOST_FORMAT_COO 2
#define SPMV_HOST_FORMAT_ELL 3
#define SPMV_HOST_FORMAT_HYBRID 4

#define SPMV_HOST_EXPAND_SYMMETRIC 0x1u /* EXTENSION: mirror the entries of a symmetric file */

typedef struct spmv_host_matrix spmv_host_matrix;

const char *spmv_host_last_error(void);

/* Load (or generate) a matrix and convert it to `format`. */
int spmv_host_load(const char *path, int format, unsigned flags, spmv_host_matrix **out);

/* Rows [row_begin, row_end) of a matrix as their own CSR matrix (row_ptr rebased to 0, column
 * indices global): one rank's share under the reference's static row partition
 * (src/matrix/csr-matrix.cpp:77-95).  Generated matrices only produce the rows asked for where
 * the family allows it (poisson2d, kkt); files are loaded whole and cut. */
int spmv_host_load_csr_rows(const char *path, unsigned flags, int64_t row_begin, int64_t row_end,
                            spmv_host_matrix **out);

void spmv_host_matrix_free(spmv_host_matrix *m);

/* out[] receives up to n of:
 *  [0] format  [1] rows  [2] columns  [3] num_entries (entries of the file; padding not counted)
 *  [4] stored entries of the main arrays (CSR: row_ptr[rows]; COO: num_entries; ELL / hybrid:
 *      rows * row_length)
 *  [5] ELL row length (ELL, hybrid)  [6] COO remainder entries (hybrid)
 *  [7] rows of the whole matrix (differs from [1] only after spmv_host_load_csr_rows)
 *  [8] the reference's "matrix_size" in bytes (Matrix::size())  [9] 1 if the file was symmetric and was expanded */
int spmv_host_matrix_info(const spmv_host_matrix *m, int64_t *out, int n);

/* The arrays, exactly as the reference's Matrix structs hold them:
 *   CSR:    0 row_ptr int32[rows+1]     1 column_index int32[stored]  2 value f64[stored]
 *   COO:    0 row_index int32[stored]   1 column_index                2 value      (file order, 0-based)
 *   ELL:    1 column_index int32[rows*row_length] (row-major, k = i*row_length + l)  2 value
 *   HYBRID: 1, 2 the ELL part;  3 coo_row_index  4 coo_column_index  5 coo_value
 * Returns NULL for an array the format does not have. */
const void *spmv_host_matrix_array(const spmv_host_matrix *m, int which);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_HOST_H */
