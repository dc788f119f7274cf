"""Row-range partitioning of a CSR matrix across the GPUs of one node.

The reference already cuts the rows into one contiguous block per OpenMP thread,
``chunk = ceil(rows / T)`` (src/matrix/csr-matrix-spmv.cpp:154-161, the same rule as
``Matrix::spmv_rows_per_thread``, src/matrix/csr-matrix.cpp:77-84).  Ranks take the
place of threads here: rank g owns rows ``[g*chunk, min(rows, (g+1)*chunk))``, holds
the full x, computes its y segment, and one all-gather assembles y.
"""
import numpy as np


def row_chunk(rows, parts):
    """ceil(rows / parts): the reference's chunk size."""
    return (rows + parts - 1) // parts if parts > 0 else rows


def row_range(rows, part, parts):
    """Rows [begin, end) of partition `part` under the reference's static rule."""
    chunk = row_chunk(rows, parts)
    return min(rows, part * chunk), min(rows, (part + 1) * chunk)


def nnz_balanced_ranges(row_ptr, parts):
    """Alternative split on row boundaries with ~equal stored entries per part:
    boundaries[g] = first row whose row_ptr >= g*nnz/parts (binary search)."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    rows = len(row_ptr) - 1
    nnz = int(row_ptr[-1] - row_ptr[0])
    targets = row_ptr[0] + (np.arange(parts + 1, dtype=np.int64) * nnz) // parts
    b = np.searchsorted(row_ptr, targets, side="left")
    b[0], b[-1] = 0, rows
    b = np.maximum.accumulate(np.minimum(b, rows))
    return [(int(b[g]), int(b[g + 1])) for g in range(parts)]


def csr_slice(row_ptr, col_idx, val, begin, end):
    """Rows [begin, end) as their own CSR matrix: row_ptr rebased to start at 0,
    column indices unchanged (x is replicated, so they stay global)."""
    row_ptr = np.asarray(row_ptr)
    k0, k1 = int(row_ptr[begin]), int(row_ptr[end])
    p = (row_ptr[begin:end + 1].astype(np.int64) - k0).astype(np.int32)
    return p, np.ascontiguousarray(col_idx[k0:k1]), np.ascontiguousarray(val[k0:k1])
