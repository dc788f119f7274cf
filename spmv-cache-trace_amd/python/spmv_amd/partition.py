"""Row-range partitioning of a CSR matrix across the GPUs of one node.

The reference already cuts the rows into one contiguous block per OpenMP thread,
``chunk = ceil(rows / T)`` (src/matrix/csr-matrix-spmv.cpp:154-161, the same rule as
``Matrix::spmv_rows_per_thread``, src/matrix/csr-matrix.cpp:77-84).  Ranks take the
place of threads here: rank g owns rows ``[g*chunk, min(rows, (g+1)*chunk))``, holds
the full x, computes its y segment, and one all-gather assembles y.

The rule itself lives in ONE place, the C library (spmv_hip_partition_rows, csrc/multi_gpu.hip): the drop-in's
single-process context (spmv_hip_create_multi) and these one-process-per-GPU helpers cut by the same function.
"""
import ctypes as C

import numpy as np


def _boundaries(rows, parts, row_ptr=None):
    from . import capi
    out = np.zeros(parts + 1, dtype=np.int32)
    rp = None if row_ptr is None else np.ascontiguousarray(row_ptr, dtype=np.int32)
    capi.check(capi.load().spmv_hip_partition_rows(int(rows), int(parts), None if rp is None else rp.ctypes.data_as(C.c_void_p),
                                                   0 if rp is None else 1, out.ctypes.data_as(C.c_void_p)))
    return out


def row_chunk(rows, parts):
    """ceil(rows / parts): the reference's chunk size (the longest block of the static rule)."""
    if parts <= 0:
        return rows
    b = _boundaries(rows, parts)
    return max(1, int(np.max(np.diff(b)))) if rows > 0 else 0


def row_range(rows, part, parts):
    """Rows [begin, end) of partition `part` under the reference's static rule."""
    b = _boundaries(rows, parts)
    return int(b[part]), int(b[part + 1])


def nnz_balanced_ranges(row_ptr, parts):
    """Alternative split on row boundaries with ~equal stored entries per part:
    boundaries[g] = first row whose row_ptr >= g*nnz/parts."""
    b = _boundaries(len(row_ptr) - 1, parts, row_ptr)
    return [(int(b[g]), int(b[g + 1])) for g in range(parts)]


def csr_slice(row_ptr, col_idx, val, begin, end):
    """Rows [begin, end) as their own CSR matrix: row_ptr rebased to start at 0,
    column indices unchanged (x is replicated, so they stay global)."""
    row_ptr = np.asarray(row_ptr)
    k0, k1 = int(row_ptr[begin]), int(row_ptr[end])
    p = (row_ptr[begin:end + 1].astype(np.int64) - k0).astype(np.int32)
    return p, np.ascontiguousarray(col_idx[k0:k1]), np.ascontiguousarray(val[k0:k1])
