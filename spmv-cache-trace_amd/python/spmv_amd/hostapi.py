"""ctypes binding of include/spmv_host.h (libspmv_host.so): the repo's own Matrix Market
loader, format converters and matrix generators, for front ends that are not C++
(bench.py --matrix / --workload, the tests).

    A = hostapi.load("Queen_4147.tar.gz", "csr", expand_symmetric=True)
    A = hostapi.load("synthetic:webbase", "coo")
    A.rows, A.cols, A.num_entries, A.row_ptr, A.column_index, A.value ...

The arrays are numpy views of the library's memory; they stay valid as long as the
HostMatrix object is alive (keep a reference).
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.path.join(PKG, "libspmv_host.so")

FORMATS = {"csr": 1, "coo": 2, "ell": 3, "hybrid": 4}
EXPAND_SYMMETRIC = 0x1

_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        from . import capi
        capi._share_torch_hip_runtime()  # libspmv_host.so links libspmv_hip.so: one HIP runtime per process
        L = C.CDLL(LIB_PATH)
        L.spmv_host_last_error.restype = C.c_char_p
        L.spmv_host_load.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
        L.spmv_host_load_csr_rows.argtypes = [C.c_char_p, C.c_uint, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
        L.spmv_host_matrix_free.argtypes = [C.c_void_p]
        L.spmv_host_matrix_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.spmv_host_matrix_array.argtypes = [C.c_void_p, C.c_int]
        L.spmv_host_matrix_array.restype = C.c_void_p
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class HostMatrix:
    """A matrix held by the host library, in one of the reference's four formats."""

    def __init__(self, handle):
        self._h = handle
        L = lib()
        info = (C.c_int64 * 10)()
        L.spmv_host_matrix_info(handle, info, 10)
        (self.format_id, self.rows, self.cols, self.num_entries, self.stored, self.row_length,
         self.num_coo_entries, self.rows_total, self.matrix_size, expanded) = [int(v) for v in info]
        self.expanded = bool(expanded)
        self.format = {v: k for k, v in FORMATS.items()}[self.format_id]
        arr = lambda which, n, dt: _view(L.spmv_host_matrix_array(handle, which), n, dt)
        if self.format == "csr":
            self.row_ptr = arr(0, self.rows + 1, np.int32)
            self.column_index = arr(1, self.stored, np.int32)
            self.value = arr(2, self.stored, np.float64)
        elif self.format == "coo":
            self.row_index = arr(0, self.stored, np.int32)
            self.column_index = arr(1, self.stored, np.int32)
            self.value = arr(2, self.stored, np.float64)
        else:
            self.column_index = arr(1, self.stored, np.int32)
            self.value = arr(2, self.stored, np.float64)
            if self.format == "hybrid":
                self.coo_row_index = arr(3, self.num_coo_entries, np.int32)
                self.coo_column_index = arr(4, self.num_coo_entries, np.int32)
                self.coo_value = arr(5, self.num_coo_entries, np.float64)

    def close(self):
        if self._h:
            for name in ("row_ptr", "row_index", "column_index", "value", "coo_row_index", "coo_column_index", "coo_value"):
                if hasattr(self, name):
                    delattr(self, name)
            lib().spmv_host_matrix_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check(rc):
    if rc != 0:
        raise HostError("%s (code %d)" % (lib().spmv_host_last_error().decode(), rc))


def load(path, fmt="csr", expand_symmetric=False):
    """Load a Matrix Market file (or generate "synthetic:<spec>") and convert it to `fmt`."""
    h = C.c_void_p()
    _check(lib().spmv_host_load(os.fsencode(path), FORMATS[fmt], EXPAND_SYMMETRIC if expand_symmetric else 0, C.byref(h)))
    return HostMatrix(h)


def load_csr_rows(path, row_begin, row_end, expand_symmetric=False):
    """Rows [row_begin, row_end) as their own CSR matrix (row_ptr rebased, columns global)."""
    h = C.c_void_p()
    _check(lib().spmv_host_load_csr_rows(os.fsencode(path), EXPAND_SYMMETRIC if expand_symmetric else 0,
                                         int(row_begin), int(row_end), C.byref(h)))
    return HostMatrix(h)
