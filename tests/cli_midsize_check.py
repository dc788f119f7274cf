"""End-to-end check of the CLI on mid-size generated files (not part of the pytest suite: it writes
~100 MB of Matrix Market text): Poisson 1024^2 and a scattered matrix through every hip-* kernel with
--check, which compares the device result with the CPU kernel of the same format."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python"))
from spmv_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "spmv-cache-trace_amd", "spmv-cache-trace-hip")
TC = os.path.join(ROOT, "tests", "golden", "trace_config_1thread.json")


def main():
    d = tempfile.mkdtemp()
    cases = {"poisson1024": synth.poisson2d(1024), "random": synth.random_uniform(200000, 800000, 12, seed=5)}
    for name, (rows, cols, p, c, v) in cases.items():
        path = os.path.join(d, name + ".mtx")
        i, j, a = synth.csr_to_coordinate(rows, p, c, v)
        synth.write_mtx(path, rows, cols, i, j, a)
        for fmt in ("hip-csr", "hip-coo", "hip-ell", "hip-hybrid"):
            if name == "random" and fmt == "hip-ell":
                pass
            r = subprocess.run([CLI, "-c", TC, "--spmv-format", fmt, "--matrix", path, "--profile=5", "--check"],
                               capture_output=True, text=True)
            ok = r.returncode == 0
            dev = {}
            if ok:
                doc = json.loads(r.stdout)
                dev = doc.get("kernel", {}).get("device", {})
                t = doc.get("execution_time", {}).get("median")
            print("%-12s %-10s rc=%d median=%s ns device=%s" % (name, fmt, r.returncode, t if ok else "-", json.dumps(dev)))
            if not ok:
                print(r.stderr[-600:])
                sys.exit(1)
    print("all ok")


if __name__ == "__main__":
    main()
