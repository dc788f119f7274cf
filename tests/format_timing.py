"""Ad-hoc timing of the context-level CSR / COO / hybrid paths on a power-law matrix (webbase-like,
BASELINE configs[4]); builds the hybrid layout with the oracle, hence lives under tests/."""
import sys, os, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "spmv-cache-trace_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py
from spmv_amd import capi, synth
O = oracle_py.Oracle()
rows, cols, p, c, v = synth.powerlaw(1000005, 1000005, seed=4)
i, j, a = synth.csr_to_coordinate(rows, p, c, v)
H = O.hybrid_from_coordinate(rows, i, j, a)
print("hybrid: L=%d ell entries %d coo entries %d" % (H["row_length"], rows * H["row_length"], len(H["coo_val"])))
x = synth.x_vector(cols)
ctx = capi.Context(0)
def timeit(n=50):
    ts = []
    for _ in range(n):
        ctx.run(); ts.append(ctx.last_run_ns())
    return np.median(ts) / 1e3
ctx.upload_hybrid(rows, cols, H["row_length"], H["ell_col"], H["ell_val"], H["coo_row"], H["coo_col"], H["coo_val"]); ctx.set_x(x)
print("hip-hybrid median %.1f us" % timeit())
r = (i - 1).astype(np.int32)
ctx.upload_coo(rows, cols, r, (j - 1).astype(np.int32), a); ctx.set_x(x)
print("hip-coo    median %.1f us" % timeit())
ctx.upload_csr(rows, cols, p, c, v); ctx.set_x(x)
print("hip-csr    median %.1f us" % timeit())
ctx.close()

# scattered triplets: 2 M rows x 24 random columns, with and without column panels
rows, cols, p, c, v = synth.random_uniform(2000000, 2000000, 24, seed=3)
i, j, a = synth.csr_to_coordinate(rows, p, c, v)
x = synth.x_vector(cols)
for flags, label in ((0, "column panels"), (capi.FLAG_NO_COLUMN_PANELS, "no panels")):
    ctx = capi.Context(0, flags=flags)
    ctx.upload_coo(rows, cols, (i - 1).astype(np.int32), (j - 1).astype(np.int32), a); ctx.set_x(x)
    print("random 2M x 24  hip-coo (%s) median %.1f us" % (label, timeit(20)))
    ctx.close()
rows, cols, p, c, v = synth.powerlaw(1000005, 1000005, seed=4)
i, j, a = synth.csr_to_coordinate(rows, p, c, v)
x = synth.x_vector(cols)
for flags, label in ((0, "column panels"), (capi.FLAG_NO_COLUMN_PANELS, "no panels")):
    ctx = capi.Context(0, flags=flags)
    ctx.upload_coo(rows, cols, (i - 1).astype(np.int32), (j - 1).astype(np.int32), a); ctx.set_x(x)
    print("power law       hip-coo (%s) median %.1f us  (panel blocks %d)" % (label, timeit(50), ctx.info()["panel_tiles"]))
    ctx.close()
