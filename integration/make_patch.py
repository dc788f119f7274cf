#!/usr/bin/env python3
"""Regenerates integration/reference.patch: the edits a maintainer of spmv-cache-trace makes to bind libspmv_hip.so.

    python integration/make_patch.py [/root/reference]

Works on two temporary copies of the reference's src/ + Makefile; nothing of the reference is written into this
repository except the patch's own context lines.  The new files of the patch (src/kernels/hip-spmv.{hpp,cpp}) are
integration/src/kernels/hip-spmv.{hpp,cpp} verbatim (tests/test_reference_integration.py checks that).

Edits (anchored on short unique strings, so a moved line does not silently mis-apply):
  src/kernels.hpp            includes kernels/hip-spmv.hpp                              (#ifdef USE_SPMV_HIP)
  src/main.cpp:27-36         kernel_type gets kernel_hip_csr / _coo / _ell
  src/main.cpp:142-150       --spmv-format hip-csr | hip-coo | hip-ell
  src/main.cpp:187           the option's help text names them
  src/main.cpp:209-232       the factory makes hip_{csr,coo,ell}_spmv_kernel
  src/util/perf-events.cpp:35-45  ONLY under USE_SPMV_HIP: a libpfm_context can be constructed in a NO_LIBPFM build; asking it for an event group
                             or the event list still fails with "Please re-build with libpfm enabled" -- so that
                             main.cpp:247 no longer makes --profile unreachable without libpfm
  Makefile                   SPMV_HIP_ROOT=<engine checkout>: -DUSE_SPMV_HIP, -I.../include, -lspmv_hip + rpath, hip-spmv.cpp
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old, s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)


def apply_edits(root):
    for f in ("hip-spmv.hpp", "hip-spmv.cpp"):
        shutil.copy(os.path.join(HERE, "src", "kernels", f), os.path.join(root, "src", "kernels", f))
    edit(os.path.join(root, "src/kernels.hpp"), [
        ('#include "kernels/hybrid-spmv.hpp"\n',
         '#include "kernels/hybrid-spmv.hpp"\n#ifdef USE_SPMV_HIP\n#include "kernels/hip-spmv.hpp"\n#endif\n')])
    edit(os.path.join(root, "src/main.cpp"), [
        ('    kernel_hybrid,\n};', '    kernel_hybrid,\n    kernel_hip_csr,\n    kernel_hip_coo,\n    kernel_hip_ell,\n};'),
        ('        else if (strcmp(arg, "hybrid") == 0) args.kernel_type = kernel_hybrid;\n',
         '        else if (strcmp(arg, "hybrid") == 0) args.kernel_type = kernel_hybrid;\n'
         '#ifdef USE_SPMV_HIP\n'
         '        else if (strcmp(arg, "hip-csr") == 0) args.kernel_type = kernel_hip_csr;\n'
         '        else if (strcmp(arg, "hip-coo") == 0) args.kernel_type = kernel_hip_coo;\n'
         '        else if (strcmp(arg, "hip-ell") == 0) args.kernel_type = kernel_hip_ell;\n'
         '#endif\n'),
        ('"choose one of: coo, coo-atomic, csr, ell, mkl-csr and hybrid", 0},',
         '"choose one of: coo, coo-atomic, csr, ell, mkl-csr and hybrid"\n'
         '#ifdef USE_SPMV_HIP\n'
         '         "; on an AMD GPU (libspmv_hip): hip-csr, hip-coo and hip-ell"\n'
         '#endif\n'
         '         , 0},'),
        ('        kernel = std::make_unique<hybrid_spmv_kernel>(args.matrix_path);\n        break;\n',
         '        kernel = std::make_unique<hybrid_spmv_kernel>(args.matrix_path);\n        break;\n'
         '#ifdef USE_SPMV_HIP\n'
         '    case kernel_hip_csr:\n'
         '        kernel = std::make_unique<hip_csr_spmv_kernel>(args.matrix_path);\n'
         '        break;\n'
         '    case kernel_hip_coo:\n'
         '        kernel = std::make_unique<hip_coo_spmv_kernel>(args.matrix_path);\n'
         '        break;\n'
         '    case kernel_hip_ell:\n'
         '        kernel = std::make_unique<hip_ell_spmv_kernel>(args.matrix_path);\n'
         '        break;\n'
         '#endif\n'
         '    default:\n'
         '        break;\n')])
    edit(os.path.join(root, "src/util/perf-events.cpp"), [
        # (only in a build that binds the engine: without SPMV_HIP_ROOT the reference behaves as before, throw included)
        ('#else\n    throw perf_error("Please re-build with libpfm enabled");\n#endif\n}\n\nlibpfm_context::~libpfm_context()',
         '#elif !defined(USE_SPMV_HIP)\n    throw perf_error("Please re-build with libpfm enabled");\n#endif\n}\n\nlibpfm_context::~libpfm_context()')])
    edit(os.path.join(root, "Makefile"), [
        ('# Default\n',
         '# MI355X kernels: make SPMV_HIP_ROOT=<checkout of the engine> (its include/spmv_hip.h and built libspmv_hip.so)\n'
         'ifdef SPMV_HIP_ROOT\n'
         'SPMV_HIP_LIBDIR ?= $(SPMV_HIP_ROOT)/spmv-cache-trace_amd\n'
         'CXXFLAGS += -DUSE_SPMV_HIP\n'
         'INCLUDES += -I$(SPMV_HIP_ROOT)/include\n'
         'LDFLAGS += -L$(SPMV_HIP_LIBDIR) -lspmv_hip -Wl,-rpath,$(SPMV_HIP_LIBDIR)\n'
         'endif\n\n# Default\n'),
        ('\tsrc/kernels/kernel.cpp\n', '\tsrc/kernels/kernel.cpp\nifdef SPMV_HIP_ROOT\nkernels_sources += src/kernels/hip-spmv.cpp\nendif\n'),
        ('\tsrc/kernels/kernel.hpp\n', '\tsrc/kernels/kernel.hpp\nifdef SPMV_HIP_ROOT\nkernels_headers += src/kernels/hip-spmv.hpp\nendif\n')])


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    with tempfile.TemporaryDirectory() as tmp:
        for side in ("a", "b"):
            os.makedirs(os.path.join(tmp, side))
            shutil.copytree(os.path.join(ref, "src"), os.path.join(tmp, side, "src"))
            shutil.copy(os.path.join(ref, "Makefile"), os.path.join(tmp, side, "Makefile"))
        apply_edits(os.path.join(tmp, "b"))
        r = subprocess.run(["diff", "-urN", "-U2", "a", "b"], cwd=tmp, stdout=subprocess.PIPE, text=True)
        assert r.returncode == 1, "diff found nothing"
        # no time stamps in the headers: the file is the same whenever it is regenerated
        lines = []
        for l in r.stdout.splitlines(keepends=True):
            if l.startswith(("--- ", "+++ ")):
                l = l.split("\t")[0].rstrip("\n") + "\n"
            lines.append(l)
        open(os.path.join(HERE, "reference.patch"), "w").write("".join(lines))
    print("wrote", os.path.join(HERE, "reference.patch"))


if __name__ == "__main__":
    main()
