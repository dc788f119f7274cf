// spmv_kernels.hpp -- every hand-written gfx950 kernel of y += A*x (fp64 values, int32 indices), by file:
//   tile_common.hpp       descriptor format, shared helpers
//   csr_wavetile.hpp      default CSR path (wave tiles and their classes)
//   csr_segtile.hpp       balanced tiles for skewed rows
//   csr_blockwin.hpp      one-ring block window (unstructured bands)
//   csr_segwin.hpp        segment windows (meshes in natural ordering, KKT systems)
//   csr_panels.hpp        column panels, plan time
//   csr_plan_kernels.hpp  tile classification, 16-bit columns, patterns
//   csr_basic.hpp         scalar / vector / adaptive CSR kernels
//   coo_kernels.hpp, ell_kernels.hpp, upload_kernels.hpp, triad_kernels.hpp
#pragma once

#include "tile_common.hpp"
#include "csr_basic.hpp"
#include "csr_wavetile.hpp"
#ifdef SPMV_HIP_EXPERIMENTS
#include "csr_rowgroup.hpp" // retired from the product library (internal.hpp)
#endif
#include "csr_segtile.hpp"
#include "csr_panels.hpp"
#include "csr_blockwin.hpp"
#include "csr_plan_kernels.hpp"
#include "coo_kernels.hpp"
#include "ell_kernels.hpp"
#include "upload_kernels.hpp"
#include "triad_kernels.hpp"
