"""Identity of the device library a measurement was taken with.

A rocprofv3 summary under profiles/ is only evidence for the binary it profiled.  Every bench line
therefore carries `build` = {source_sha256, lib_sha256, git_head}: the hash of the sources
libspmv_hip.so is compiled from (csrc/ + include/spmv_hip*.h; reproducible wherever the library is
rebuilt), the hash of the .so the process really loaded, and the commit (from .git where the tree
has one, else from the stamp file __graft_entry__.build() leaves next to the library: the GPU box
gets a snapshot without .git).  bench.py reports PMC traffic from a committed summary only when its
source hash equals the running one.
"""
import glob
import hashlib
import os
import subprocess

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPO_ROOT = os.path.dirname(PKG_ROOT)
HEAD_STAMP = os.path.join(PKG_ROOT, ".build_head")  # git-ignored, travels with gpurun


def _sha_files(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.relpath(p, REPO_ROOT).encode())
        h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def device_sources():
    """What libspmv_hip.so is compiled from (Makefile: CSRC + CHDR)."""
    src = sorted(glob.glob(os.path.join(PKG_ROOT, "csrc", "*.hip")) + glob.glob(os.path.join(PKG_ROOT, "csrc", "*.hpp")))
    return src + sorted(glob.glob(os.path.join(REPO_ROOT, "include", "spmv_hip*.h")))  # spmv_hip.h, spmv_hip_plan.h, spmv_hip_tuning.h


def source_sha256():
    return _sha_files(device_sources())


def lib_sha256(path):
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def git_head():
    """Commit of the working tree ('+dirty' when tracked files differ), or the stamp of the last build, or None."""
    try:
        r = subprocess.run(["git", "-C", REPO_ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            head = r.stdout.strip()
            d = subprocess.run(["git", "-C", REPO_ROOT, "status", "--porcelain", "--untracked-files=no"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=30)
            return head + ("+dirty" if d.returncode == 0 and d.stdout.strip() else "")
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        return open(HEAD_STAMP).read().strip() or None
    except OSError:
        return None


def write_head_stamp():
    """Called by __graft_entry__.build(): remember the commit for trees that travel without .git."""
    head = git_head()
    if head:
        with open(HEAD_STAMP, "w") as f:
            f.write(head + "\n")
    return head


def build_info(lib_path):
    return {"source_sha256": source_sha256()[:16], "lib_sha256": (lib_sha256(lib_path) or "")[:16] or None,
            "git_head": git_head()}
