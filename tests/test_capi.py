"""The C-ABI library loads and exports every symbol include/spmv_hip.h, spmv_hip_tuning.h and spmv_hip_plan.h declare; argument
validation that needs no device works; and with no GPU the compute entry points FAIL
(there is no CPU fallback).  No compute calls here."""
import ctypes as C
import re

import numpy as np
import pytest

from spmv_amd import capi


def _declared_symbols(paths=None):
    out = set()
    for path in (paths or capi.HEADER_PATHS):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(spmv_hip_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_header_symbols_all_exported():
    lib = C.CDLL(capi.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), "libspmv_hip.so does not export %s" % s
    # and the Python binding covers exactly the declared set
    assert sorted(capi.SIGNATURES.keys()) == syms


def test_the_boundary_header_is_the_small_one():
    """include/spmv_hip.h is the SURVEY 8(b) surface -- the context functions an adapter binds, the error
    codes and seven flags -- and nothing of the plan API or the tuning switches; the adapter of INTEGRATION.md compiles against
    a directory that holds that header ALONE."""
    import os
    import shutil
    import subprocess
    import tempfile
    small = _declared_symbols([capi.HEADER_PATH])
    want = {"spmv_hip_create", "spmv_hip_create_multi", "spmv_hip_destroy", "spmv_hip_upload_csr", "spmv_hip_upload_coo", "spmv_hip_upload_ell",
            "spmv_hip_upload_hybrid", "spmv_hip_set_x", "spmv_hip_set_y", "spmv_hip_get_y", "spmv_hip_run", "spmv_hip_sync", "spmv_hip_flush_caches",
            "spmv_hip_last_run_ns", "spmv_hip_last_run_times", "spmv_hip_strerror", "spmv_hip_last_error", "spmv_hip_version", "spmv_hip_device_count"}
    assert set(small) == want, sorted(set(small) ^ want)
    text = open(capi.HEADER_PATH).read()
    flags = sorted(set(re.findall(r"#define (SPMV_HIP_FLAG_[A-Z0-9_]+)", text)))
    assert flags == ["SPMV_HIP_FLAG_BALANCE_ENTRIES", "SPMV_HIP_FLAG_COO_KEEP_ORDER", "SPMV_HIP_FLAG_EXACT_ORDER", "SPMV_HIP_FLAG_FUSED_PEER_STORE",
                     "SPMV_HIP_FLAG_NO_RUN_EVENTS", "SPMV_HIP_FLAG_PEER_GATHER", "SPMV_HIP_FLAG_PIPELINE_GATHER"], flags
    assert "spmv_hip_plan" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    # every header is plain C99 on its own
    root = os.path.dirname(os.path.dirname(capi.HEADER_PATH))
    for h in capi.HEADER_PATHS:
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(h), "-fsyntax-only", "-x", "c", "-"],
                           input='#include "%s"\n' % os.path.basename(h), text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout
    # the reference-side adapter sees nothing but the small header
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference tree is not here: the adapter's other includes cannot be resolved")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(capi.HEADER_PATH, tmp)
        src = os.path.join(root, "integration", "src")
        r = subprocess.run(["g++", "-std=c++14", "-fopenmp", "-DUSE_OPENMP", "-DUSE_POSIX_MEMALIGN", "-DUSE_SPMV_HIP", "-include", "cstdint", "-I", tmp, "-I", "/root/reference/src", "-I", "/root/reference/src/kernels", "-I", src,
                            "-fsyntax-only", os.path.join(src, "kernels", "hip-spmv.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-3000:]


def test_version_and_strerror():
    lib = capi.load()
    assert lib.spmv_hip_version() == 130
    assert lib.spmv_hip_strerror(0) == b"success"
    # the ELL overflow message is the reference's (src/matrix/ell-matrix.cpp:202-204)
    assert b"Integer overflow when computing number of non-zeros" in lib.spmv_hip_strerror(capi.ERR_OVERFLOW)
    for code in range(-7, 0):
        assert lib.spmv_hip_strerror(code) not in (b"", b"unknown error")
    assert lib.spmv_hip_strerror(-99) == b"unknown error"


def test_device_count_is_reported():
    assert capi.device_count() >= 0


def test_plan_argument_validation_needs_no_device():
    lib = capi.load()
    h = C.c_void_p()
    bad = np.array([0, 3, 2, 5], dtype=np.int32)  # decreasing
    assert lib.spmv_hip_plan_csr(C.byref(h), 3, 3, bad, capi.CSR_ADAPTIVE, 0, 0) == capi.ERR_INVALID
    assert b"non-decreasing" in lib.spmv_hip_last_error()
    ok = np.array([0, 1, 2, 3], dtype=np.int32)
    assert lib.spmv_hip_plan_csr(C.byref(h), 3, 3, ok, 9, 0, 0) == capi.ERR_INVALID
    assert lib.spmv_hip_plan_csr(C.byref(h), 3, 3, ok, capi.CSR_VECTOR, 3, 0) == capi.ERR_INVALID
    assert lib.spmv_hip_plan_csr(C.byref(h), -1, 3, ok, capi.CSR_VECTOR, 0, 0) == capi.ERR_INVALID
    assert lib.spmv_hip_coo_spmv(-1, 0, None, None, None, None, None, None) == capi.ERR_INVALID
    assert lib.spmv_hip_ell_spmv(70000, 40000, None, None, None, None, None) == capi.ERR_OVERFLOW


def test_peer_entry_points_validate_their_arguments_without_a_device():
    """The one-process-per-GPU entry points (spmv_hip_ipc_*, spmv_hip_peer_push, spmv_hip_csr_spmv_out_peers) refuse bad
    arguments with a code before they touch HIP -- and, like everything else, never fall back to anything on a box
    without a GPU."""
    lib = capi.load()
    p = C.c_void_p()
    assert lib.spmv_hip_ipc_alloc(C.byref(p), 0, C.create_string_buffer(64)) == capi.ERR_INVALID  # zero bytes
    assert lib.spmv_hip_ipc_alloc(None, 64, C.create_string_buffer(64)) == capi.ERR_INVALID
    assert lib.spmv_hip_ipc_open(None, C.byref(p)) == capi.ERR_INVALID
    assert lib.spmv_hip_ipc_close(None) == 0 and lib.spmv_hip_ipc_free(None) == 0  # nothing to close / free
    assert lib.spmv_hip_peer_push(None, None, -1, 8, None) == capi.ERR_INVALID
    assert lib.spmv_hip_peer_push(None, None, 0, 8, None) == 0  # no peers: nothing to do
    assert lib.spmv_hip_peer_push(None, None, 2, 8, None) == capi.ERR_INVALID  # peers but no pointers
    fused = C.c_int(7)
    assert lib.spmv_hip_csr_spmv_out_peers(None, None, None, None, None, None, None, None, 3, C.byref(fused), None) == capi.ERR_INVALID
    assert lib.spmv_hip_csr_spmv_out_peers(None, None, None, None, None, None, None, None, 0, C.byref(fused), None) == capi.ERR_INVALID  # null plan
    if capi.device_count() == 0:
        assert lib.spmv_hip_ipc_alloc(C.byref(p), 4096, C.create_string_buffer(64)) in (capi.ERR_NO_DEVICE, capi.ERR_HIP)
        assert not p.value


def test_no_gpu_means_failure_not_fallback():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present; this test covers the no-device behaviour")
    with pytest.raises(capi.SpmvHipError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.SpmvHipError):
        capi.CsrPlan(3, 3, np.array([0, 1, 2, 3], dtype=np.int32), capi.CSR_ADAPTIVE)
