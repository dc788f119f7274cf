// spmv-kernels.hpp -- the SpMV kernel types behind the Kernel interface.
//
//   csr_spmv_kernel / coo_spmv_kernel / coo_spmv_atomic_kernel / ell_spmv_kernel
//       the reference's CPU kernels (src/kernels/{csr,coo,coo-spmv-atomic,ell}-spmv.cpp):
//       OpenMP loops over host arrays.  They keep the CLI's CPU path (BASELINE configs[0]).
//   hip_csr_spmv_kernel / hip_coo_spmv_kernel / hip_ell_spmv_kernel
//       the same three formats multiplied on an MI355X through the C ABI (include/spmv_hip.h).
//       No fallback: if the device library cannot run, init() throws kernel_error.
//
// All of them load the matrix in init() exactly as the reference does (Matrix Market file ->
// format conversion, x = 1.0, y = 0.0, errors rewrapped as "<path>: <what>") and print the same
// JSON object (name, matrix_path, matrix_format, rows, columns, nonzeros, matrix_size, x_size,
// y_size); the HIP kernels add a "device" object.
#pragma once

#include "kernel.hpp"

#include "../matrix/coo-matrix.hpp"
#include "../matrix/csr-matrix.hpp"
#include "../matrix/ell-matrix.hpp"
#include "../matrix/hybrid-matrix.hpp"

#include <memory>
#include <string>

struct spmv_hip_ctx;

// Options shared by every SpMV kernel type (all default to the reference's behaviour).
struct SpmvOptions
{
    bool expand_symmetric = false; // EXTENSION: mirror symmetric files
    int device = 0;                // HIP device index
    int num_gpus = 0;              // > 0: rows partitioned over devices 0..num_gpus-1 (spmv_hip_create_multi; CSR only)
    int csr_algorithm = 0;         // SPMV_HIP_CSR_*
    int csr_lanes_per_row = 0;
    unsigned hip_flags = 0;        // SPMV_HIP_FLAG_*
};

enum class SpmvFormat { csr, coo, coo_atomic, ell, hybrid };

// Factory: `hip` selects the GPU implementation of the format.
std::unique_ptr<Kernel> make_spmv_kernel(SpmvFormat format, bool hip, std::string const & matrix_path,
                                         SpmvOptions const & options = SpmvOptions());
