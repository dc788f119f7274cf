"""The drop-in proven on the reference tree itself (src/kernels/kernel.hpp:18-45, src/main.cpp:209-232).

Where /root/reference exists (the build container): a temporary copy of its src/ + Makefile gets
integration/reference.patch, is built by the PATCHED Makefile (`make NO_LIBPFM=1 SPMV_HIP_ROOT=<this repo>`) against
libspmv_hip.so, and the resulting program is run: --help names the new formats, `--spmv-format hip-csr` without a device
ends in kernel_error's message (no CPU fallback), and -- because the patch also makes --profile reachable in a build
without libpfm -- the reference's own timed loop and JSON writer run here for the first time, so its document is compared
field by field with the one this repo's CLI writes for the same input.  Nothing of this travels to the GPU box except the
two binaries oracle/Makefile builds into oracle/_ref/ (tests/test_gpu_reference_binary.py)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "reference.patch")
MTX = os.path.join(ROOT, "tests", "golden", "poisson2D.mtx")
LIB = os.path.join(ROOT, "spmv-cache-trace_amd", "libspmv_hip.so")
CLI = os.path.join(ROOT, "spmv-cache-trace_amd", "spmv-cache-trace-hip")

CONFIG = {"caches": {"L1": {"size": 32768, "line_size": 64, "bandwidth": None, "bandwidth_per_numa_domain": None,
                            "cache_miss_event": None, "parent": None}},
          "num_numa_domains": 1,
          "thread_affinities": [{"cpu": 0, "cache": "L1", "numa_domain": 0, "event_groups": []}]}

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")), reason="the reference tree is not here")


def test_patch_carries_the_adapter_files_verbatim():
    """The new files inside the patch are integration/src/kernels/hip-spmv.{hpp,cpp}: one source of truth."""
    text = open(PATCH).read()
    for name in ("hip-spmv.hpp", "hip-spmv.cpp"):
        body = open(os.path.join(ROOT, "integration", "src", "kernels", name)).read()
        start = text.index("+++ b/src/kernels/" + name)
        hunk = text[start:].split("\n", 2)[2]
        end = hunk.find("\ndiff -urN")
        added = [l[1:] for l in (hunk if end < 0 else hunk[:end]).splitlines() if l.startswith("+")]
        assert "\n".join(added).strip() == body.strip(), name


@pytest.fixture(scope="module")
def patched_tree(tmp_path_factory):
    tree = tmp_path_factory.mktemp("reference_patched")
    shutil.copytree(os.path.join(REFERENCE, "src"), tree / "src")
    shutil.copy(os.path.join(REFERENCE, "Makefile"), tree / "Makefile")
    r = subprocess.run(["patch", "-p1", "-d", str(tree), "-i", PATCH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "fuzz" not in r.stdout and "FAILED" not in r.stdout, r.stdout
    assert os.path.exists(LIB), "build libspmv_hip.so first (__graft_entry__.build())"
    r = subprocess.run(["make", "-C", str(tree), "-j8", "NO_LIBPFM=1", "CXX=g++ -include cstdint", "SPMV_HIP_ROOT=" + ROOT, "spmv-cache-trace"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    cfg = tree / "config.json"
    cfg.write_text(json.dumps(CONFIG))
    return tree


def _run(tree, *args):
    return subprocess.run([str(tree / "spmv-cache-trace")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@needs_reference
def test_patched_reference_builds_and_lists_the_formats(patched_tree):
    r = _run(patched_tree, "--help")
    assert r.returncode == 0
    flat = " ".join(r.stdout.split())
    assert "hip-csr, hip-coo and hip-ell" in flat and "coo, coo-atomic, csr, ell, mkl-csr and hybrid" in flat
    ldd = subprocess.run(["ldd", str(patched_tree / "spmv-cache-trace")], stdout=subprocess.PIPE, text=True).stdout
    assert "libspmv_hip.so" in ldd and "not found" not in ldd


@needs_reference
@pytest.mark.parametrize("fmt", ["hip-csr", "hip-coo", "hip-ell"])
def test_patched_reference_without_a_device_fails_like_kernel_error(patched_tree, fmt):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = _run(patched_tree, "-c", str(patched_tree / "config.json"), "-m", MTX, "--spmv-format", fmt, "--profile=3")
    assert r.returncode == 1 and r.stdout == ""
    # main.cpp:264-266: "<kernel name>: <what>", what = "<path>: <reason>" as csr-spmv.cpp:37-45 builds it
    assert r.stderr.strip() == "%s-spmv: %s: no HIP device available (this library has no CPU fallback)" % (fmt, MTX)


@needs_reference
def test_patched_reference_without_the_build_switch_is_the_reference(tmp_path):
    """No SPMV_HIP_ROOT: the patched tree builds exactly what the reference builds (the new code is behind USE_SPMV_HIP)."""
    shutil.copytree(os.path.join(REFERENCE, "src"), tmp_path / "src")
    shutil.copy(os.path.join(REFERENCE, "Makefile"), tmp_path / "Makefile")
    assert subprocess.run(["patch", "-s", "-p1", "-d", str(tmp_path), "-i", PATCH]).returncode == 0
    r = subprocess.run(["make", "-C", str(tmp_path), "-j8", "NO_LIBPFM=1", "CXX=g++ -include cstdint", "spmv-cache-trace"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    h = subprocess.run([str(tmp_path / "spmv-cache-trace"), "--help"], stdout=subprocess.PIPE, text=True).stdout
    assert "hip-csr" not in h
    ldd = subprocess.run(["ldd", str(tmp_path / "spmv-cache-trace")], stdout=subprocess.PIPE, text=True).stdout
    assert "libspmv_hip" not in ldd
    bad = subprocess.run([str(tmp_path / "spmv-cache-trace"), "-c", "x", "--spmv-format", "hip-csr"], stderr=subprocess.PIPE, text=True)
    assert bad.returncode != 0 and "invalid argument" in bad.stderr


def _shape(doc):
    """Keys and value types of a JSON document, numbers folded together."""
    if isinstance(doc, dict):
        return {k: _shape(v) for k, v in doc.items()}
    if isinstance(doc, list):
        return [_shape(v) for v in doc]
    if isinstance(doc, bool) or doc is None or isinstance(doc, str):
        return type(doc).__name__
    return "number"


@needs_reference
@pytest.mark.parametrize("fmt", ["csr", "coo", "ell"])
def test_the_reference_timed_loop_writes_the_document_this_repo_writes(patched_tree, fmt):
    """src/profile-kernel.cpp:340-391 + src/util/sample.hpp, run for real (the patch makes --profile reachable without
    libpfm): same keys in the same order, same kernel block, as host/main.cpp's CPU selection for the same input."""
    cfg = str(patched_tree / "config.json")
    ref = _run(patched_tree, "-c", cfg, "-m", MTX, "--spmv-format", fmt, "--profile=5")
    assert ref.returncode == 0, ref.stderr
    own = subprocess.run([CLI, "-c", cfg, "-m", MTX, "--spmv-format", fmt, "--profile=5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert own.returncode == 0, own.stderr
    a, b = json.loads(ref.stdout), json.loads(own.stdout)
    assert list(a) == ["trace_config", "kernel", "execution_time", "profiling_events"]
    for key in a:
        assert key in b
    assert a["trace_config"] == b["trace_config"]
    assert a["kernel"] == {k: b["kernel"][k] for k in a["kernel"]}
    assert list(a["execution_time"]) == [k for k in b["execution_time"] if k in a["execution_time"]]
    assert _shape(a["execution_time"]) == _shape({k: b["execution_time"][k] for k in a["execution_time"]})
    assert a["execution_time"]["samples"] == b["execution_time"]["samples"] == 5 and a["execution_time"]["unit"] == "ns"
