"""ctypes bindings for the CPU checker (TEST INFRASTRUCTURE ONLY).

Two libraries are exposed:

* ``Oracle``  -- oracle/liboracle_spmv.so, the plain-C restatement of the
  reference loops (oracle/spmv_oracle.c).
* ``RefLib``  -- oracle/_ref/libref_spmv.so, the reference library itself
  compiled from /root/reference by oracle/Makefile (absent if it was never
  built; callers must cope with ``RefLib.available() == False``).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (spmv-cache-trace_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle_spmv.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_spmv.so")

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile the checker libraries (gcc only; no GPU involved)."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", _HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Plain-C restatement (oracle/spmv_oracle.c)."""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_csr_spmv.argtypes = [C.c_int32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_csr_spmv.restype = None
        L.oracle_coo_spmv.argtypes = [C.c_int, C.c_int32, C.c_int32, _i32p, _i32p, _f64p,
                                      _f64p, _f64p, _f64p]
        L.oracle_coo_spmv.restype = None
        L.oracle_coo_spmv_atomic.argtypes = [C.c_int, C.c_int32, C.c_int32, _i32p, _i32p,
                                             _f64p, _f64p, _f64p]
        L.oracle_coo_spmv_atomic.restype = None
        L.oracle_ell_spmv.argtypes = [C.c_int32, C.c_int32, _i32p, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_ell_spmv.restype = None
        L.oracle_sort_row_major.argtypes = [C.c_int32, _i32p, _i32p, _i32p]
        L.oracle_sort_row_major.restype = None
        L.oracle_csr_from_coordinate.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _f64p,
                                                 C.c_int32, _i32p, C.c_void_p, C.c_void_p]
        L.oracle_csr_from_coordinate.restype = C.c_int32
        L.oracle_coo_from_coordinate.argtypes = [C.c_int32, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.oracle_coo_from_coordinate.restype = None
        L.oracle_max_row_length.argtypes = [C.c_int32, C.c_int32, _i32p]
        L.oracle_max_row_length.restype = C.c_int32
        L.oracle_ell_from_coordinate.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _f64p,
                                                 C.c_int, C.c_int32, _i32p, _f64p]
        L.oracle_ell_from_coordinate.restype = C.c_int
        L.oracle_sample_stats.argtypes = [_i64p, C.c_int64, _f64p]
        L.oracle_sample_stats.restype = None
        L.oracle_hybrid_from_coordinate.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _f64p, C.c_int, _i32p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_hybrid_from_coordinate.restype = C.c_int
        L.oracle_hybrid_spmv.argtypes = [C.c_int, C.c_int32, C.c_int32, _i32p, _f64p, C.c_int, C.c_int32,
                                         _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]
        L.oracle_hybrid_spmv.restype = None

    # -- SpMV: all of these ACCUMULATE into (a copy of) y, like the reference --
    def csr_spmv(self, rows, row_ptr, col, val, x, y=None, num_threads=1, runs=1):
        y = np.zeros(rows) if y is None else _f64(y).copy()
        row_ptr, col, val, x = _i32(row_ptr), _i32(col), _f64(val), _f64(x)
        for _ in range(runs):
            self.lib.oracle_csr_spmv(rows, row_ptr, col, val, x, y, num_threads)
        return y

    def csr_spmv_inplace(self, rows, row_ptr, col, val, x, y, num_threads=1):
        """No copies: for timing (arrays must already be contiguous and typed)."""
        self.lib.oracle_csr_spmv(rows, row_ptr, col, val, x, y, num_threads)

    def coo_spmv(self, rows, row_idx, col, val, x, y=None, num_threads=1, runs=1):
        y = np.zeros(rows) if y is None else _f64(y).copy()
        row_idx, col, val, x = _i32(row_idx), _i32(col), _f64(val), _f64(x)
        ws = np.zeros(max(1, num_threads * rows))
        for _ in range(runs):
            self.lib.oracle_coo_spmv(num_threads, rows, len(val), row_idx, col, val, x, y, ws)
        return y

    def coo_spmv_atomic(self, rows, row_idx, col, val, x, y=None, num_threads=1, runs=1):
        y = np.zeros(rows) if y is None else _f64(y).copy()
        row_idx, col, val, x = _i32(row_idx), _i32(col), _f64(val), _f64(x)
        for _ in range(runs):
            self.lib.oracle_coo_spmv_atomic(num_threads, rows, len(val), row_idx, col, val, x, y)
        return y

    def ell_spmv(self, rows, row_length, col, val, x, y=None, num_threads=1, runs=1):
        y = np.zeros(rows) if y is None else _f64(y).copy()
        col, val, x = _i32(col), _f64(val), _f64(x)
        for _ in range(runs):
            self.lib.oracle_ell_spmv(rows, row_length, col, val, x, y, num_threads)
        return y

    # -- converters (1-based coordinate entries in) --
    def csr_from_coordinate(self, rows, i, j, a, row_alignment=1):
        i, j, a = _i32(i), _i32(j), _f64(a)
        row_ptr = np.zeros(rows + 1, dtype=np.int32)
        n = self.lib.oracle_csr_from_coordinate(rows, len(a), i, j, a, row_alignment,
                                                row_ptr, None, None)
        col = np.zeros(max(n, 1), dtype=np.int32)
        val = np.zeros(max(n, 1), dtype=np.float64)
        self.lib.oracle_csr_from_coordinate(rows, len(a), i, j, a, row_alignment, row_ptr,
                                            col.ctypes.data_as(C.c_void_p),
                                            val.ctypes.data_as(C.c_void_p))
        return row_ptr, col[:n], val[:n]

    def coo_from_coordinate(self, i, j, a):
        i, j, a = _i32(i), _i32(j), _f64(a)
        r = np.zeros(len(a), dtype=np.int32)
        c = np.zeros(len(a), dtype=np.int32)
        v = np.zeros(len(a), dtype=np.float64)
        self.lib.oracle_coo_from_coordinate(len(a), i, j, a, r, c, v)
        return r, c, v

    def max_row_length(self, rows, i):
        i = _i32(i)
        return self.lib.oracle_max_row_length(rows, len(i), i)

    def ell_from_coordinate(self, rows, i, j, a, skip_padding=False):
        i, j, a = _i32(i), _i32(j), _f64(a)
        L = self.max_row_length(rows, i)
        n = rows * L
        if n > 2**31 - 1:
            return -1, L, None, None
        col = np.zeros(max(n, 1), dtype=np.int32)
        val = np.zeros(max(n, 1), dtype=np.float64)
        rc = self.lib.oracle_ell_from_coordinate(rows, len(a), i, j, a, int(skip_padding), L, col, val)
        return rc, L, col[:n], val[:n]

    def hybrid_from_coordinate(self, rows, i, j, a, skip_padding=False):
        """-> dict(row_length, ell_col, ell_val, coo_row, coo_col, coo_val) or None on int32 overflow."""
        i, j, a = _i32(i), _i32(j), _f64(a)
        sizes = np.zeros(3, dtype=np.int32)
        if self.lib.oracle_hybrid_from_coordinate(rows, len(a), i, j, a, int(skip_padding), sizes,
                                                  None, None, None, None, None) != 0:
            return None
        L, ne, nc = (int(t) for t in sizes)
        ej, ea = np.zeros(max(1, ne), dtype=np.int32), np.zeros(max(1, ne))
        cr, cc, cv = np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc))
        vp = lambda arr: arr.ctypes.data_as(C.c_void_p)
        self.lib.oracle_hybrid_from_coordinate(rows, len(a), i, j, a, int(skip_padding), sizes,
                                               vp(ej), vp(ea), vp(cr), vp(cc), vp(cv))
        return dict(row_length=L, ell_col=ej[:ne], ell_val=ea[:ne], coo_row=cr[:nc], coo_col=cc[:nc],
                    coo_val=cv[:nc], skip_padding=bool(skip_padding))

    def hybrid_spmv(self, rows, H, x, y=None, num_threads=1, runs=1):
        y = np.zeros(rows) if y is None else _f64(y).copy()
        ws = np.zeros(max(1, num_threads * rows))
        pad = lambda arr, dt: np.ascontiguousarray(arr if len(arr) else np.zeros(1), dtype=dt)
        for _ in range(runs):
            self.lib.oracle_hybrid_spmv(num_threads, rows, H["row_length"], pad(H["ell_col"], np.int32),
                                        pad(H["ell_val"], np.float64), int(H["skip_padding"]), len(H["coo_val"]),
                                        pad(H["coo_row"], np.int32), pad(H["coo_col"], np.int32),
                                        pad(H["coo_val"], np.float64), _f64(x), y, ws)
        return y

    def sample_stats(self, v):
        v = np.ascontiguousarray(v, dtype=np.int64)
        out = np.zeros(8)
        self.lib.oracle_sample_stats(v, len(v), out)
        return dict(zip(["min", "max", "mean", "median", "variance", "standard_deviation",
                         "skewness", "kurtosis"], out.tolist()))


class RefLib:
    """The reference library itself (oracle/_ref/libref_spmv.so)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO):
        L = self.lib = C.CDLL(path)
        vp = C.c_void_p
        L.ref_last_error.restype = C.c_char_p
        L.ref_mm_from_string.argtypes = [C.c_char_p, C.c_int64]
        L.ref_mm_from_string.restype = vp
        L.ref_mm_load.argtypes = [C.c_char_p]
        L.ref_mm_load.restype = vp
        L.ref_mm_free.argtypes = [vp]
        L.ref_mm_info.argtypes = [vp, _i32p]
        L.ref_mm_entries.argtypes = [vp, _i32p, _i32p, _f64p]
        L.ref_mm_max_row_length.argtypes = [vp]
        L.ref_mm_max_row_length.restype = C.c_int32
        L.ref_csr_from_mm.argtypes = [vp, C.c_int32]
        L.ref_csr_from_mm.restype = vp
        L.ref_csr_from_arrays.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f64p]
        L.ref_csr_from_arrays.restype = vp
        L.ref_csr_free.argtypes = [vp]
        L.ref_csr_info.argtypes = [vp, _i64p]
        L.ref_csr_arrays.argtypes = [vp, _i32p, _i32p, _f64p]
        L.ref_csr_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.ref_csr_spmv.restype = C.c_int
        L.ref_csr_spmv_timed.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int, _i64p]
        L.ref_csr_spmv_timed.restype = C.c_int
        L.ref_coo_from_mm.argtypes = [vp]
        L.ref_coo_from_mm.restype = vp
        L.ref_coo_from_arrays.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f64p]
        L.ref_coo_from_arrays.restype = vp
        L.ref_coo_free.argtypes = [vp]
        L.ref_coo_info.argtypes = [vp, _i64p]
        L.ref_coo_arrays.argtypes = [vp, _i32p, _i32p, _f64p]
        L.ref_coo_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.ref_coo_spmv.restype = C.c_int
        L.ref_ell_from_mm.argtypes = [vp, C.c_int]
        L.ref_ell_from_mm.restype = vp
        L.ref_ell_free.argtypes = [vp]
        L.ref_ell_info.argtypes = [vp, _i64p]
        L.ref_ell_arrays.argtypes = [vp, _i32p, _f64p]
        L.ref_ell_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.ref_ell_spmv.restype = C.c_int
        L.ref_print_sample.argtypes = [_i64p, C.c_int64, C.c_char_p, C.c_int64]
        L.ref_print_sample.restype = C.c_int64
        L.ref_hybrid_from_mm.argtypes = [vp, C.c_int]
        L.ref_hybrid_from_mm.restype = vp
        L.ref_hybrid_free.argtypes = [vp]
        L.ref_hybrid_info.argtypes = [vp, _i64p]
        L.ref_hybrid_arrays.argtypes = [vp, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.ref_hybrid_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.ref_hybrid_spmv.restype = C.c_int
        L.ref_trace_config_echo.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, _i32p]
        L.ref_trace_config_echo.restype = C.c_int64

    def error(self):
        return self.lib.ref_last_error().decode()

    # -- Matrix Market --
    def mm_from_string(self, text):
        b = text if isinstance(text, bytes) else text.encode()
        h = self.lib.ref_mm_from_string(b, len(b))
        if not h:
            raise RuntimeError(self.error())
        return h

    def mm_load(self, path):
        h = self.lib.ref_mm_load(path.encode())
        if not h:
            raise RuntimeError(self.error())
        return h

    def mm_free(self, h):
        self.lib.ref_mm_free(h)

    def mm_info(self, h):
        out = np.zeros(6, dtype=np.int32)
        self.lib.ref_mm_info(h, out)
        return dict(zip(["rows", "columns", "num_entries", "format", "field", "symmetry"],
                        out.tolist()))

    def mm_entries(self, h):
        n = self.mm_info(h)["num_entries"]
        i = np.zeros(n, dtype=np.int32)
        j = np.zeros(n, dtype=np.int32)
        a = np.zeros(n)
        self.lib.ref_mm_entries(h, i, j, a)
        return i, j, a

    def mm_max_row_length(self, h):
        return self.lib.ref_mm_max_row_length(h)

    # -- CSR --
    def csr_from_mm(self, h, row_alignment=1):
        A = self.lib.ref_csr_from_mm(h, row_alignment)
        if not A:
            raise RuntimeError(self.error())
        return A

    def csr_from_arrays(self, rows, cols, row_ptr, col, val):
        row_ptr, col, val = _i32(row_ptr), _i32(col), _f64(val)
        return self.lib.ref_csr_from_arrays(rows, cols, len(val), row_ptr, col, val)

    def csr_free(self, A):
        self.lib.ref_csr_free(A)

    def csr_info(self, A):
        out = np.zeros(6, dtype=np.int64)
        self.lib.ref_csr_info(A, out)
        return dict(zip(["rows", "columns", "num_entries", "row_alignment", "stored", "size"],
                        out.tolist()))

    def csr_arrays(self, A):
        info = self.csr_info(A)
        p = np.zeros(info["rows"] + 1, dtype=np.int32)
        j = np.zeros(max(1, info["stored"]), dtype=np.int32)
        a = np.zeros(max(1, info["stored"]))
        self.lib.ref_csr_arrays(A, p, j, a)
        return p, j[:info["stored"]], a[:info["stored"]]

    def csr_spmv(self, A, x, y=None, num_threads=1, runs=1):
        info = self.csr_info(A)
        y = np.zeros(info["rows"]) if y is None else _f64(y).copy()
        if self.lib.ref_csr_spmv(A, _f64(x), y, num_threads, runs) != 0:
            raise RuntimeError(self.error())
        return y

    def csr_spmv_timed(self, A, x, num_threads, runs):
        info = self.csr_info(A)
        y = np.zeros(info["rows"])
        ns = np.zeros(runs, dtype=np.int64)
        if self.lib.ref_csr_spmv_timed(A, _f64(x), y, num_threads, runs, ns) != 0:
            raise RuntimeError(self.error())
        return ns, y

    # -- COO --
    def coo_from_mm(self, h):
        A = self.lib.ref_coo_from_mm(h)
        if not A:
            raise RuntimeError(self.error())
        return A

    def coo_from_arrays(self, rows, cols, row_idx, col, val):
        row_idx, col, val = _i32(row_idx), _i32(col), _f64(val)
        return self.lib.ref_coo_from_arrays(rows, cols, len(val), row_idx, col, val)

    def coo_free(self, A):
        self.lib.ref_coo_free(A)

    def coo_info(self, A):
        out = np.zeros(4, dtype=np.int64)
        self.lib.ref_coo_info(A, out)
        return dict(zip(["rows", "columns", "num_entries", "size"], out.tolist()))

    def coo_arrays(self, A):
        n = self.coo_info(A)["num_entries"]
        r = np.zeros(max(1, n), dtype=np.int32)
        c = np.zeros(max(1, n), dtype=np.int32)
        v = np.zeros(max(1, n))
        self.lib.ref_coo_arrays(A, r, c, v)
        return r[:n], c[:n], v[:n]

    def coo_spmv(self, A, x, y=None, num_threads=1, runs=1):
        info = self.coo_info(A)
        y = np.zeros(info["rows"]) if y is None else _f64(y).copy()
        if self.lib.ref_coo_spmv(A, _f64(x), y, num_threads, runs) != 0:
            raise RuntimeError(self.error())
        return y

    # -- ELL --
    def ell_from_mm(self, h, skip_padding=False):
        A = self.lib.ref_ell_from_mm(h, int(skip_padding))
        if not A:
            raise RuntimeError(self.error())
        return A

    def ell_free(self, A):
        self.lib.ref_ell_free(A)

    def ell_info(self, A):
        out = np.zeros(6, dtype=np.int64)
        self.lib.ref_ell_info(A, out)
        return dict(zip(["rows", "columns", "num_entries", "row_length", "stored", "size"],
                        out.tolist()))

    def ell_arrays(self, A):
        n = self.ell_info(A)["stored"]
        c = np.zeros(max(1, n), dtype=np.int32)
        v = np.zeros(max(1, n))
        self.lib.ref_ell_arrays(A, c, v)
        return c[:n], v[:n]

    def ell_spmv(self, A, x, y=None, num_threads=1, runs=1):
        info = self.ell_info(A)
        y = np.zeros(info["rows"]) if y is None else _f64(y).copy()
        if self.lib.ref_ell_spmv(A, _f64(x), y, num_threads, runs) != 0:
            raise RuntimeError(self.error())
        return y

    # -- HYBRID --
    def hybrid_from_mm(self, h, skip_padding=False):
        A = self.lib.ref_hybrid_from_mm(h, int(skip_padding))
        if not A:
            raise RuntimeError(self.error())
        return A

    def hybrid_free(self, A):
        self.lib.ref_hybrid_free(A)

    def hybrid_info(self, A):
        out = np.zeros(7, dtype=np.int64)
        self.lib.ref_hybrid_info(A, out)
        return dict(zip(["rows", "columns", "num_entries", "row_length", "ell_stored", "coo_entries", "size"],
                        out.tolist()))

    def hybrid_arrays(self, A):
        info = self.hybrid_info(A)
        ne, nc = info["ell_stored"], info["coo_entries"]
        ej, ea = np.zeros(max(1, ne), dtype=np.int32), np.zeros(max(1, ne))
        cr, cc, cv = np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc))
        self.lib.ref_hybrid_arrays(A, ej, ea, cr, cc, cv)
        return ej[:ne], ea[:ne], cr[:nc], cc[:nc], cv[:nc]

    def hybrid_spmv(self, A, x, y=None, num_threads=1, runs=1):
        info = self.hybrid_info(A)
        y = np.zeros(info["rows"]) if y is None else _f64(y).copy()
        if self.lib.ref_hybrid_spmv(A, _f64(x), y, num_threads, runs) != 0:
            raise RuntimeError(self.error())
        return y

    def print_sample(self, v):
        v = np.ascontiguousarray(v, dtype=np.int64)
        buf = C.create_string_buffer(4096)
        n = self.lib.ref_print_sample(v, len(v), buf, 4096)
        assert n >= 0
        return buf.value.decode()

    def trace_config_echo(self, path):
        """(json text, info dict) of the reference's parse + print of a trace-config file;
        raises RuntimeError with the reference's message on a trace_config_error."""
        buf = C.create_string_buffer(1 << 16)
        info = np.zeros(4, dtype=np.int32)
        n = self.lib.ref_trace_config_echo(path.encode(), buf, 1 << 16, info)
        if n == -1:
            raise RuntimeError(self.error())
        assert n >= 0
        return buf.value.decode(), dict(zip(["threads", "numa_domains", "caches", "max_cache_size"],
                                            info.tolist()))
