"""ctypes access to the host library's test hooks (spmv-cache-trace_amd/host/test-hooks.cpp)
and a helper to run the CLI binary."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spmv-cache-trace_amd")
HOST_SO = os.path.join(PKG, "libspmv_host_test.so")  # the test hooks; links libspmv_host.so
CLI = os.path.join(PKG, "spmv-cache-trace-hip")

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
vp = C.c_void_p


class HostError(RuntimeError):
    pass


class Host:
    def __init__(self):
        from spmv_amd import capi
        capi._share_torch_hip_runtime()  # one HIP runtime per process (see capi.load)
        L = self.lib = C.CDLL(HOST_SO)
        L.host_last_error.restype = C.c_char_p
        for name in ("host_mm_from_buffer", "host_mm_load", "host_mm_load_tar_gz_member", "host_mm_expand_symmetry",
                     "host_csr_from_mm", "host_coo_from_mm", "host_ell_from_mm", "host_hybrid_from_mm"):
            getattr(L, name).restype = vp
        L.host_mm_from_buffer.argtypes = [C.c_char_p, C.c_longlong]
        L.host_mm_load.argtypes = [C.c_char_p]
        L.host_mm_load_tar_gz_member.argtypes = [C.c_char_p, C.c_char_p]
        L.host_mm_expand_symmetry.argtypes = [vp]
        L.host_mm_free.argtypes = [vp]
        L.host_mm_info.argtypes = [vp, _i32p]
        L.host_mm_entries.argtypes = [vp, _i32p, _i32p, _f64p]
        L.host_mm_comment.argtypes = [vp, C.c_int, C.c_char_p, C.c_longlong]
        L.host_mm_comment.restype = C.c_longlong
        L.host_mm_max_row_length.argtypes = [vp]
        L.host_mm_sorted_entries.argtypes = [vp, C.c_int, _i32p, _i32p, _f64p]
        L.host_csr_from_mm.argtypes = [vp, C.c_int32]
        L.host_csr_free.argtypes = [vp]
        L.host_csr_info.argtypes = [vp, _i64p]
        L.host_csr_arrays.argtypes = [vp, _i32p, _i32p, _f64p]
        L.host_csr_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.host_coo_from_mm.argtypes = [vp]
        L.host_coo_free.argtypes = [vp]
        L.host_coo_info.argtypes = [vp, _i64p]
        L.host_coo_arrays.argtypes = [vp, _i32p, _i32p, _f64p]
        L.host_coo_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int, C.c_int]
        L.host_ell_from_mm.argtypes = [vp, C.c_int]
        L.host_ell_free.argtypes = [vp]
        L.host_ell_info.argtypes = [vp, _i64p]
        L.host_ell_arrays.argtypes = [vp, _i32p, _f64p]
        L.host_ell_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.host_hybrid_from_mm.argtypes = [vp, C.c_int]
        L.host_hybrid_free.argtypes = [vp]
        L.host_hybrid_info.argtypes = [vp, _i64p]
        L.host_hybrid_arrays.argtypes = [vp, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.host_hybrid_spmv.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_int]
        L.host_print_sample.argtypes = [_i64p, C.c_longlong, C.c_char_p, C.c_longlong]
        L.host_print_sample.restype = C.c_longlong
        L.host_trace_config_echo.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong, _i32p]
        L.host_trace_config_echo.restype = C.c_longlong

    def _h(self, h):
        if not h:
            raise HostError(self.lib.host_last_error().decode())
        return h

    def _rc(self, rc):
        if rc != 0:
            raise HostError(self.lib.host_last_error().decode())

    # ---- Matrix Market
    def mm_from_text(self, text):
        b = text if isinstance(text, bytes) else text.encode()
        return self._h(self.lib.host_mm_from_buffer(b, len(b)))

    def mm_load(self, path):
        return self._h(self.lib.host_mm_load(path.encode()))

    def mm_load_tar_gz_member(self, path, member):
        return self._h(self.lib.host_mm_load_tar_gz_member(path.encode(), member.encode()))

    def mm_expand_symmetry(self, h):
        return self._h(self.lib.host_mm_expand_symmetry(h))

    def mm_free(self, h):
        self.lib.host_mm_free(h)

    def mm_info(self, h):
        out = np.zeros(7, dtype=np.int32)
        self.lib.host_mm_info(h, out)
        return dict(zip(["rows", "columns", "num_entries", "format", "field", "symmetry", "comments"], out.tolist()))

    def mm_entries(self, h):
        n = self.mm_info(h)["num_entries"]
        i, j, a = np.zeros(max(1, n), dtype=np.int32), np.zeros(max(1, n), dtype=np.int32), np.zeros(max(1, n))
        self._rc(self.lib.host_mm_entries(h, i, j, a))
        return i[:n], j[:n], a[:n]

    def mm_comment(self, h, k):
        buf = C.create_string_buffer(4096)
        assert self.lib.host_mm_comment(h, k, buf, 4096) >= 0
        return buf.value.decode()

    def mm_max_row_length(self, h):
        r = self.lib.host_mm_max_row_length(h)
        if r < 0:
            raise HostError(self.lib.host_last_error().decode())
        return r

    def mm_sorted(self, h, column_major=False):
        n = self.mm_info(h)["num_entries"]
        i, j, a = np.zeros(max(1, n), dtype=np.int32), np.zeros(max(1, n), dtype=np.int32), np.zeros(max(1, n))
        self._rc(self.lib.host_mm_sorted_entries(h, int(column_major), i, j, a))
        return i[:n], j[:n], a[:n]

    # ---- formats
    def csr(self, h, row_alignment=1):
        A = self._h(self.lib.host_csr_from_mm(h, row_alignment))
        info = np.zeros(6, dtype=np.int64)
        self.lib.host_csr_info(A, info)
        rows, cols, nnz, align, stored, size = info.tolist()
        p = np.zeros(rows + 1, dtype=np.int32)
        j, a = np.zeros(max(1, stored), dtype=np.int32), np.zeros(max(1, stored))
        self.lib.host_csr_arrays(A, p, j, a)
        return A, dict(rows=rows, columns=cols, num_entries=nnz, row_alignment=align, size=size), p, j[:stored], a[:stored]

    def csr_spmv(self, A, rows, x, y=None, threads=1, runs=1):
        y = np.zeros(rows) if y is None else np.array(y, dtype=np.float64)
        self._rc(self.lib.host_csr_spmv(A, np.ascontiguousarray(x, dtype=np.float64), y, threads, runs))
        return y

    def coo(self, h):
        A = self._h(self.lib.host_coo_from_mm(h))
        info = np.zeros(4, dtype=np.int64)
        self.lib.host_coo_info(A, info)
        rows, cols, nnz, size = info.tolist()
        r, c, v = np.zeros(max(1, nnz), dtype=np.int32), np.zeros(max(1, nnz), dtype=np.int32), np.zeros(max(1, nnz))
        self.lib.host_coo_arrays(A, r, c, v)
        return A, dict(rows=rows, columns=cols, num_entries=nnz, size=size), r[:nnz], c[:nnz], v[:nnz]

    def coo_spmv(self, A, rows, x, y=None, threads=1, runs=1, atomic=False):
        y = np.zeros(rows) if y is None else np.array(y, dtype=np.float64)
        self._rc(self.lib.host_coo_spmv(A, np.ascontiguousarray(x, dtype=np.float64), y, threads, runs, int(atomic)))
        return y

    def ell(self, h, skip_padding=False):
        A = self._h(self.lib.host_ell_from_mm(h, int(skip_padding)))
        info = np.zeros(6, dtype=np.int64)
        self.lib.host_ell_info(A, info)
        rows, cols, nnz, L, stored, size = info.tolist()
        c, v = np.zeros(max(1, stored), dtype=np.int32), np.zeros(max(1, stored))
        self.lib.host_ell_arrays(A, c, v)
        return A, dict(rows=rows, columns=cols, num_entries=nnz, row_length=L, size=size), c[:stored], v[:stored]

    def ell_spmv(self, A, rows, x, y=None, threads=1, runs=1):
        y = np.zeros(rows) if y is None else np.array(y, dtype=np.float64)
        self._rc(self.lib.host_ell_spmv(A, np.ascontiguousarray(x, dtype=np.float64), y, threads, runs))
        return y

    def hybrid(self, h, skip_padding=False):
        A = self._h(self.lib.host_hybrid_from_mm(h, int(skip_padding)))
        info = np.zeros(7, dtype=np.int64)
        self.lib.host_hybrid_info(A, info)
        rows, cols, nnz, L, ne, nc, size = info.tolist()
        ej, ea = np.zeros(max(1, ne), dtype=np.int32), np.zeros(max(1, ne))
        cr, cc, cv = np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc), dtype=np.int32), np.zeros(max(1, nc))
        self.lib.host_hybrid_arrays(A, ej, ea, cr, cc, cv)
        return A, dict(rows=rows, columns=cols, num_entries=nnz, row_length=L, size=size), \
            ej[:ne], ea[:ne], cr[:nc], cc[:nc], cv[:nc]

    def hybrid_spmv(self, A, rows, x, y=None, threads=1, runs=1):
        y = np.zeros(rows) if y is None else np.array(y, dtype=np.float64)
        self._rc(self.lib.host_hybrid_spmv(A, np.ascontiguousarray(x, dtype=np.float64), y, threads, runs))
        return y

    # ---- statistics / JSON
    def print_sample(self, v):
        v = np.ascontiguousarray(v, dtype=np.int64)
        buf = C.create_string_buffer(4096)
        assert self.lib.host_print_sample(v, len(v), buf, 4096) >= 0
        return buf.value.decode()

    def trace_config_echo(self, path):
        buf = C.create_string_buffer(1 << 16)
        info = np.zeros(4, dtype=np.int32)
        n = self.lib.host_trace_config_echo(path.encode(), buf, 1 << 16, info)
        if n == -1:
            raise HostError(self.lib.host_last_error().decode())
        return buf.value.decode(), dict(zip(["threads", "numa_domains", "caches", "max_cache_size"], info.tolist()))


def run_cli(*args, timeout=120):
    """Run the CLI; returns (exit code, stdout, stderr)."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "2")
    r = subprocess.run([CLI] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout, env=env)
    return r.returncode, r.stdout, r.stderr
