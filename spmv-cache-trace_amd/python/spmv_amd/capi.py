"""ctypes binding of include/spmv_hip.h, spmv_hip_tuning.h and spmv_hip_plan.h (the C ABI of libspmv_hip.so).

This is plumbing: it loads the in-tree shared library and turns negative return
codes into ``SpmvHipError``.  There is deliberately no fallback of any kind: if
the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.path.join(PKG_ROOT, "libspmv_hip.so")
# tools/ only: SPMV_HIP_EXPERIMENTS=1 loads the build with the timing experiments compiled in
EXPERIMENTS_LIB_PATH = os.path.join(PKG_ROOT, "libspmv_hip_experiments.so")
if os.environ.get("SPMV_HIP_EXPERIMENTS") == "1":
    LIB_PATH = EXPERIMENTS_LIB_PATH
elif os.environ.get("SPMV_HIP_EXPERIMENTS", "").endswith(".so"):  # an ablation build of tools/ablate.sh
    LIB_PATH = os.path.abspath(os.environ["SPMV_HIP_EXPERIMENTS"])
# the drop-in boundary (what an adapter of the reference binds) and the two headers that include it (tuning switches; Level 2)
HEADER_PATH = os.path.join(os.path.dirname(PKG_ROOT), "include", "spmv_hip.h")
HEADER_PATHS = [HEADER_PATH] + [os.path.join(os.path.dirname(PKG_ROOT), "include", n) for n in ("spmv_hip_tuning.h", "spmv_hip_plan.h")]

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_STATE, ERR_OVERFLOW, ERR_ALIGN = -1, -2, -3, -4, -5, -6, -7
CSR_AUTO, CSR_SCALAR, CSR_VECTOR, CSR_ADAPTIVE, CSR_WAVETILE = 0, 1, 2, 3, 4
FLAG_XCD_REMAP, FLAG_EXACT_ORDER, FLAG_BIG_TILE, FLAG_NO_INDEX_COMPRESSION, FLAG_COO_KEEP_ORDER, FLAG_READ_ROW_PTR, FLAG_ROWS64, FLAG_ROWS128, FLAG_ELL_COLUMN_MAJOR = 0x1, 0x2, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200
FLAG_NO_SHIFTED_TILES = 0x400
FLAG_PIPELINE_GATHER = 0x4  # create_multi: gather k on a second stream beside multiply k + 1 (two alternating copies of y)
FLAG_NO_X_WINDOW = 0x800
FLAG_NO_COLUMN_PANELS = 0x1000
FLAG_VERIFY_PLAN = 0x8000
FLAG_NO_BALANCED_TILES = 0x40000
FLAG_NO_RUN_EVENTS = 0x80000
FLAG_NO_VALUE_INDEX = 0x100000
FLAG_PEER_GATHER = 0x200000
FLAG_BALANCE_ENTRIES = 0x400000
FLAG_NO_SEGMENT_WINDOW = 0x800000
FLAG_FUSED_PEER_STORE = 0x1000000
FLAG_NO_BLOCK_TILES = 0x2000000
FLAG_HUB_COLUMNS = 0x4000000  # libspmv_hip_experiments.so only (retired from the product: csrc/internal.hpp)
FLAG_NO_MULTI_WINDOW = 0x8000000
FLAG_NO_MASKED_BLOCKS = 0x20000000
FLAG_ROW_GROUPS = 0x10000000  # libspmv_hip_experiments.so only (retired from the product: csrc/internal.hpp)
CSR_ALGORITHM_NAMES = {1: "scalar", 2: "vector", 3: "adaptive", 4: "wavetile"}

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/spmv_hip*.h declare
SIGNATURES = {
    "spmv_hip_version": (C.c_int, []),
    "spmv_hip_strerror": (C.c_char_p, [C.c_int]),
    "spmv_hip_last_error": (C.c_char_p, []),
    "spmv_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "spmv_hip_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_uint]),
    "spmv_hip_create_multi": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_uint]),
    "spmv_hip_destroy": (None, [_vp]),
    "spmv_hip_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "spmv_hip_set_csr_algorithm": (C.c_int, [_vp, C.c_int, C.c_int]),
    "spmv_hip_upload_csr": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f64p]),
    "spmv_hip_upload_coo": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f64p]),
    "spmv_hip_upload_ell": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _i32p, _f64p]),
    "spmv_hip_upload_hybrid": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _i32p, _f64p, C.c_int32, _i32p, _i32p, _f64p]),
    "spmv_hip_set_x": (C.c_int, [_vp, _f64p]),
    "spmv_hip_set_y": (C.c_int, [_vp, _f64p]),
    "spmv_hip_get_y": (C.c_int, [_vp, _f64p]),
    "spmv_hip_run": (C.c_int, [_vp]),
    "spmv_hip_sync": (C.c_int, [_vp]),
    "spmv_hip_flush_caches": (C.c_int, [_vp]),
    "spmv_hip_last_run_ns": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "spmv_hip_last_run_times": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "spmv_hip_ctx_info": (C.c_int, [_vp, _i64p, C.c_int]),
    "spmv_hip_plan_csr": (C.c_int, [C.POINTER(_vp), C.c_int32, C.c_int32, _i32p, C.c_int, C.c_int, C.c_uint]),
    "spmv_hip_plan_csr_compress": (C.c_int, [_vp, _vp, _vp]),
    "spmv_hip_plan_verify": (C.c_int, [_vp, _vp, _vp]),
    "spmv_hip_plan_csr_repack": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_plan_csr_refresh_values": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_plan_csr_index_values": (C.c_int, [_vp, _vp, _vp]),
    "spmv_hip_plan_destroy": (None, [_vp]),
    "spmv_hip_plan_info": (C.c_int, [_vp, _i64p, C.c_int]),
    "spmv_hip_plan_csr_confirm_blocks": (C.c_int, [_vp, _vp, _vp, C.c_void_p, _vp]),
    "spmv_hip_csr_spmv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_csr_spmv_out": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_csr_spmv_out_peers": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), C.c_int, C.POINTER(C.c_int), _vp]),
    "spmv_hip_partition_rows": (C.c_int, [C.c_int32, C.c_int, _vp, C.c_int, _vp]),
    "spmv_hip_ipc_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t, C.c_char_p]),
    "spmv_hip_ipc_open": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "spmv_hip_ipc_close": (C.c_int, [_vp]),
    "spmv_hip_ipc_free": (C.c_int, [_vp]),
    "spmv_hip_peer_push": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int64, _vp]),
    "spmv_hip_coo_spmv": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_coo_sort_by_row": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "spmv_hip_ell_to_column_major": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_ell_spmv": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "spmv_hip_triad": (C.c_int, [C.c_int64, _vp, _vp, _vp, C.c_double, _vp]),
}


class SpmvHipError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__("%s (%d): %s" % (what, code, detail))
        self.code = code


_lib = None
hip_runtime_path = None  # which libamdhip64 the process ended up with (diagnostic)


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and
    ask for it by file name, so a process that first loads /opt/rocm's copy through
    libspmv_hip.so and then imports torch ends up with TWO HIP runtimes, and the second
    one sees no GPU.  Loading torch's copy first makes the dynamic loader satisfy our
    DT_NEEDED libamdhip64.so.7 with it (SONAME match), and torch later finds the same
    file already mapped.  Set SPMV_HIP_RUNTIME=system to skip this (no torch in the
    process)."""
    global hip_runtime_path
    if os.environ.get("SPMV_HIP_RUNTIME", "torch") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        hip_runtime_path = cand


def load():
    """Load libspmv_hip.so (in-tree).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `make -C %s lib` (there is no CPU fallback)"
                % (LIB_PATH, PKG_ROOT))
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(lib, name) and os.environ.get("SPMV_HIP_EXPERIMENTS", "").endswith(".so"):
                continue  # an A/B against an OLDER build (tools/ab_two_libs.sh): entry points added since are simply not there
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        lib = load()
        raise SpmvHipError(rc, lib.spmv_hip_strerror(rc).decode(), lib.spmv_hip_last_error().decode())


def device_count():
    n = C.c_int(0)
    check(load().spmv_hip_device_count(C.byref(n)))
    return n.value


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_EMPTY_I32 = np.zeros(1, dtype=np.int32)
_EMPTY_F64 = np.zeros(1, dtype=np.float64)


class Context:
    """Level-1 API: host arrays in, host arrays out (what the C++ adapters use)."""

    def __init__(self, device=0, flags=0, num_gpus=None):
        """One device (`device`), or, with num_gpus, a multi-GPU context over devices 0..num_gpus-1
        (row blocks + one in-place RCCL all-gather per run; CSR only)."""
        self.lib = load()
        h = _vp()
        if num_gpus is None:
            check(self.lib.spmv_hip_create(C.byref(h), device, flags))
        else:
            check(self.lib.spmv_hip_create_multi(C.byref(h), num_gpus, flags))
        self.h = h
        self.rows = self.cols = 0

    def close(self):
        if self.h:
            self.lib.spmv_hip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream=None):
        """Launch on the caller's stream (a raw hipStream_t, e.g. torch's cuda_stream); None = the
        context's own stream again."""
        check(self.lib.spmv_hip_set_stream(self.h, stream or 0, 1 if stream is None else 0))

    def set_csr_algorithm(self, algorithm, lanes_per_row=0):
        check(self.lib.spmv_hip_set_csr_algorithm(self.h, algorithm, lanes_per_row))

    def upload_csr(self, rows, cols, row_ptr, col, val):
        row_ptr, col, val = _i32(row_ptr), _i32(col), _f64(val)
        nnz = int(row_ptr[rows]) if len(row_ptr) > rows else -1
        if len(col) == 0:
            col, val = _EMPTY_I32, _EMPTY_F64
        check(self.lib.spmv_hip_upload_csr(self.h, rows, cols, nnz, row_ptr, col, val))
        self.rows, self.cols = rows, cols

    def upload_coo(self, rows, cols, row_idx, col, val):
        row_idx, col, val = _i32(row_idx), _i32(col), _f64(val)
        nnz = len(val)
        if nnz == 0:
            row_idx, col, val = _EMPTY_I32, _EMPTY_I32, _EMPTY_F64
        check(self.lib.spmv_hip_upload_coo(self.h, rows, cols, nnz, row_idx, col, val))
        self.rows, self.cols = rows, cols

    def upload_ell(self, rows, cols, row_length, col, val):
        col, val = _i32(col), _f64(val)
        if len(col) == 0:
            col, val = _EMPTY_I32, _EMPTY_F64
        check(self.lib.spmv_hip_upload_ell(self.h, rows, cols, row_length, col, val))
        self.rows, self.cols = rows, cols

    def upload_hybrid(self, rows, cols, row_length, ell_col, ell_val, coo_row, coo_col, coo_val):
        ell_col, ell_val = _i32(ell_col), _f64(ell_val)
        coo_row, coo_col, coo_val = _i32(coo_row), _i32(coo_col), _f64(coo_val)
        n = len(coo_val)
        if len(ell_col) == 0:
            ell_col, ell_val = _EMPTY_I32, _EMPTY_F64
        if n == 0:
            coo_row, coo_col, coo_val = _EMPTY_I32, _EMPTY_I32, _EMPTY_F64
        check(self.lib.spmv_hip_upload_hybrid(self.h, rows, cols, row_length, ell_col, ell_val, n,
                                              coo_row, coo_col, coo_val))
        self.rows, self.cols = rows, cols

    def set_x(self, x):
        x = _f64(x)
        assert len(x) == self.cols
        check(self.lib.spmv_hip_set_x(self.h, x if len(x) else _EMPTY_F64))

    def set_y(self, y):
        y = _f64(y)
        assert len(y) == self.rows
        check(self.lib.spmv_hip_set_y(self.h, y if len(y) else _EMPTY_F64))

    def get_y(self):
        y = np.zeros(max(1, self.rows))
        check(self.lib.spmv_hip_get_y(self.h, y))
        return y[:self.rows]

    def run(self, runs=1, sync=True):
        for _ in range(runs):
            check(self.lib.spmv_hip_run(self.h))
        if sync:
            check(self.lib.spmv_hip_sync(self.h))

    def flush_caches(self):
        """Evict the device's L2 and Infinity Cache (the device side of --flush-caches)."""
        check(self.lib.spmv_hip_flush_caches(self.h))

    def last_run_ns(self):
        ns = C.c_uint64(0)
        check(self.lib.spmv_hip_last_run_ns(self.h, C.byref(ns)))
        return ns.value

    def last_run_times(self):
        """(kernel_ns, gather_ns) of the last run."""
        k, g = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.spmv_hip_last_run_times(self.h, C.byref(k), C.byref(g)))
        return k.value, g.value

    def info(self):
        out = np.zeros(20, dtype=np.int64)
        check(self.lib.spmv_hip_ctx_info(self.h, out, 20))
        keys = ["format", "rows", "cols", "stored", "algorithm", "lanes_per_row", "workgroups",
                "row_blocks", "long_blocks", "device_bytes", "narrow_tiles", "shifted_tiles", "xwin_tiles",
                "blockwin_tiles", "panel_tiles", "streamed_bytes", "devices", "ell_path", "rccl_ranks", "pipelined"]
        return dict(zip(keys, out.tolist()))


class CsrPlan:
    """Level-2 launch plan for caller-owned device arrays (torch tensors)."""

    def __init__(self, rows, cols, host_row_ptr, algorithm=CSR_AUTO, lanes_per_row=0, flags=0):
        self.lib = load()
        h = _vp()
        check(self.lib.spmv_hip_plan_csr(C.byref(h), rows, cols, _i32(host_row_ptr), algorithm,
                                         lanes_per_row, flags))
        self.h = h

    def close(self):
        if self.h:
            self.lib.spmv_hip_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = np.zeros(38, dtype=np.int64)
        check(self.lib.spmv_hip_plan_info(self.h, out, 38))
        keys = ["algorithm", "lanes_per_row", "workgroups", "row_blocks", "long_blocks", "rows",
                "nnz", "meta_bytes", "narrow_tiles", "uniform_tiles", "shifted_tiles", "xwin_tiles", "blockwin_tiles", "panel_tiles",
                "streamed_bytes", "shifted_entries", "narrow_entries", "uniform_rows", "value_snapshot", "balanced", "indexed_values",
                "segwin_tiles", "segwin_slots", "value_row_tiles", "dictionary_launch_tiles", "block_tiles", "block_entries", "hub_columns", "hub_entries", "multi_window_tiles", "row_group_tiles",
                "masked_block_tiles", "masked_block_entries", "stencil_mask_tiles", "stencil_mask_entries",
                "group_tiles", "group_entries", "group_rows"]
        return dict(zip(keys, out.tolist()))

    def confirm_blocks(self, d_row_ptr, d_col, host_row_ptr=None, stream=0):
        """Before compress: a candidate for (masked) block tiles gets its tiles cut on the row groups found in the columns."""
        hp = None if host_row_ptr is None else np.ascontiguousarray(host_row_ptr, dtype=np.int32)
        check(self.lib.spmv_hip_plan_csr_confirm_blocks(self.h, d_row_ptr, d_col, None if hp is None else hp.ctypes.data, stream))

    def compress(self, d_col, stream=0):
        """16-bit column offsets for the tiles that allow it (wave-tile algorithm only)."""
        check(self.lib.spmv_hip_plan_csr_compress(self.h, d_col, stream))

    def repack(self, d_row_ptr, d_col, d_val, stream=0):
        """Column panels for scattered matrices (after compress; a no-op when the matrix does not
        qualify).  The plan then owns a snapshot of the values."""
        check(self.lib.spmv_hip_plan_csr_repack(self.h, d_row_ptr, d_col, d_val, stream))

    def spmv(self, d_row_ptr, d_col, d_val, d_x, d_y, stream=0):
        """All arguments are raw device addresses (ints), e.g. tensor.data_ptr()."""
        check(self.lib.spmv_hip_csr_spmv(self.h, d_row_ptr, d_col, d_val, d_x, d_y, stream))

    def spmv_out(self, d_row_ptr, d_col, d_val, d_x, d_y_in, d_y_out, stream=0):
        """y_out = y_in + A*x (two different arrays, or the same one)."""
        check(self.lib.spmv_hip_csr_spmv_out(self.h, d_row_ptr, d_col, d_val, d_x, d_y_in, d_y_out, stream))

    def verify(self, d_col, stream=0):
        """Raises SpmvHipError (ERR_STATE) if d_col no longer has the contents the plan was compressed from."""
        check(self.lib.spmv_hip_plan_verify(self.h, d_col, stream))

    def index_values(self, d_val, stream=0):
        """Value dictionary for matrices with at most 128 distinct values (a no-op otherwise).  The caller keeps
        d_val unchanged while the plan lives, or calls refresh_values after changing it."""
        check(self.lib.spmv_hip_plan_csr_index_values(self.h, d_val, stream))

    def refresh_values(self, d_row_ptr, d_col, d_val, stream=0):
        """Re-copy the values into the plan's column-panel copy (no-op without panels)."""
        check(self.lib.spmv_hip_plan_csr_refresh_values(self.h, d_row_ptr, d_col, d_val, stream))


def ipc_alloc(nbytes):
    """(device address, 64-byte handle) of zeroed device memory other processes can map (ipc_open)."""
    h = C.create_string_buffer(64)
    p = _vp()
    check(load().spmv_hip_ipc_alloc(C.byref(p), nbytes, h))
    return p.value, h.raw


def ipc_open(handle):
    p = _vp()
    check(load().spmv_hip_ipc_open(handle, C.byref(p)))
    return p.value


def ipc_close(addr):
    check(load().spmv_hip_ipc_close(addr))


def ipc_free(addr):
    check(load().spmv_hip_ipc_free(addr))


def peer_push(d_src, d_dst_list, n, stream=0):
    arr = (_vp * len(d_dst_list))(*d_dst_list)
    check(load().spmv_hip_peer_push(d_src, arr, len(d_dst_list), n, stream))


def coo_spmv(rows, nnz, d_row, d_col, d_val, d_x, d_y, stream=0):
    check(load().spmv_hip_coo_spmv(rows, nnz, d_row, d_col, d_val, d_x, d_y, stream))


def coo_variant(variant):
    """Sweep hook of libspmv_hip_experiments.so (SPMV_HIP_EXPERIMENTS=1), not part of the C ABI:
    1 = always the 64-entries-per-wave COO kernel."""
    lib = load()
    if not hasattr(lib, "spmv_hip_coo_variant"):
        raise RuntimeError("spmv_hip_coo_variant needs the experiments build (SPMV_HIP_EXPERIMENTS=1)")
    lib.spmv_hip_coo_variant.argtypes = [C.c_int]
    lib.spmv_hip_coo_variant.restype = None
    lib.spmv_hip_coo_variant(variant)


def coo_sort_by_row(rows, nnz, d_row, d_col, d_val, stream=0):
    """Stable in-place sort of device COO triplets by row index."""
    check(load().spmv_hip_coo_sort_by_row(rows, nnz, d_row, d_col, d_val, stream))


def ell_to_column_major(rows, row_length, d_col_rm, d_val_rm, d_col_cm, d_val_cm, stream=0):
    check(load().spmv_hip_ell_to_column_major(rows, row_length, d_col_rm, d_val_rm, d_col_cm,
                                              d_val_cm, stream))


def ell_spmv(rows, row_length, d_col_cm, d_val_cm, d_x, d_y, stream=0):
    check(load().spmv_hip_ell_spmv(rows, row_length, d_col_cm, d_val_cm, d_x, d_y, stream))


def triad(n, d_a, d_b, d_c, q=3.1, stream=0):
    """a = b + q*c on device arrays (STREAM triad, the empirical bandwidth roofline)."""
    check(load().spmv_hip_triad(n, d_a, d_b, d_c, q, stream))
