"""The C-ABI library loads and exports every symbol include/spmv_hip.h declares; argument
validation that needs no device works; and with no GPU the compute entry points FAIL
(there is no CPU fallback).  No compute calls here."""
import ctypes as C
import re

import numpy as np
import pytest

from spmv_amd import capi


def _declared_symbols():
    text = open(capi.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spmv_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = C.CDLL(capi.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), "libspmv_hip.so does not export %s" % s
    # and the Python binding covers exactly the declared set
    assert sorted(capi.SIGNATURES.keys()) == syms


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
