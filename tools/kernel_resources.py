#!/usr/bin/env python3
"""Registers, occupancy, scratch and LDS of the kernels named on the command line (substrings), from
spmv-cache-trace_amd/build/resource_usage.txt (`make -C spmv-cache-trace_amd asm`)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    want = sys.argv[1:] or ["wavetile", "segtile"]
    text = open(os.path.join(ROOT, "spmv-cache-trace_amd", "build", "resource_usage.txt")).read()
    seen = set()
    for block in re.split(r"(?=remark: [^\n]*Function Name)", text):
        m = re.search(r"Function Name: (\S+)", block)
        if not m or not any(w in m.group(1) for w in want) or m.group(1) in seen:
            continue
        seen.add(m.group(1))
        get = lambda pat: int(re.search(pat, block).group(1))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        print("%3d VGPR %2d waves/SIMD %4d B scratch %6d B LDS  %s" % (
            get(r"VGPRs: (\d+)"), get(r"Occupancy \[waves/SIMD\]: (\d+)"), get(r"ScratchSize \[bytes/lane\]: (\d+)"),
            get(r"LDS Size \[bytes/block\]: (\d+)"), re.sub(r"^void spmv::", "", name)[:120]))


if __name__ == "__main__":
    main()
