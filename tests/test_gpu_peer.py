"""One process per GPU without a collective (spmv_amd/peer.py): every rank's multiply stores its rows of y into the
other ranks' vectors, which live in inter-process device memory (spmv_hip_ipc_*).  A one-GPU box cannot give every rank
its own device, so the ranks share device 0 -- the handles, the mappings, the fused and the pushed delivery, the
completion protocol and the C ABI calls are the real ones; only the xGMI hop is missing.  Checked against the oracle's
CSR loop (src/matrix/csr-matrix-spmv.cpp:21-33) on every rank, whole vector."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, spec, fused, balanced, need=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "spmv-cache-trace_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import oracle_py
    from spmv_amd import hostapi, partition, synth
    from spmv_amd.peer import PeerCsrSpmv

    verdict = "exception"
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        A = hostapi.load(spec, "csr")
        rows, cols, p, c, v = A.rows, A.cols, np.array(A.row_ptr), np.array(A.column_index), np.array(A.value)
        A.close()
        ranges = partition.nnz_balanced_ranges(p, world) if balanced else None
        b, e = ranges[rank] if balanced else partition.row_range(rows, rank, world)
        pl, cl, vl = partition.csr_slice(p, c, v, b, e)
        x = synth.x_vector(cols, seed=9)
        op = PeerCsrSpmv.on_gpu(rows, cols, rank, world, dev, pl, cl, vl, x, ranges=ranges, fused=fused)
        steps = 3
        for _ in range(steps):
            op.step()
        got = op.y().cpu().numpy()
        O = oracle_py.Oracle()
        want = O.csr_spmv(rows, p, c, v, x, num_threads=2, runs=steps)
        scale = steps * np.bincount(np.repeat(np.arange(rows), np.diff(p)), weights=np.abs(v) * np.abs(x[c]), minlength=rows)
        ok = bool(np.all(np.abs(got - want) <= 1e-10 * np.maximum(scale, np.abs(want)) + 1e-300))
        # the wave-tile kernel (default, value dictionary, x windows) and the segment-window kernel forward their row sums
        # themselves; the other plans (balanced tiles for skewed rows, the one-ring block window, column panels, split long
        # rows) push the segment with a second launch -- and fused=False always does
        info = op.plan.info()
        ring_window = info["blockwin_tiles"] > 0 and info["segwin_tiles"] == 0
        expect_fused = (fused and world > 1 and not info["balanced"] and not ring_window and info["panel_tiles"] == 0
                        and info["long_blocks"] == 0)
        ok = ok and (op.fused == expect_fused)
        if need is not None:  # the tile class this case is here for is in this rank's plan
            ok = ok and info[need] > 0 and op.fused
        op.zero()
        ok = ok and float(op.y().abs().max().item()) == 0.0
        op.finish()  # a vector may be read until ANY rank multiplies again: this rank is done reading
        op.step()
        got1 = op.y().cpu().numpy()
        want1 = O.csr_spmv(rows, p, c, v, x, num_threads=2)
        ok = ok and bool(np.all(np.abs(got1 - want1) <= 1e-10 * np.maximum(scale, np.abs(want1)) + 1e-300))
        op.close()
        verdict = "ok" if ok else "mismatch (fused %s, expected %s, %s %s)" % (op.fused, expect_fused, need, info.get(need))
    except Exception as ex:  # the parent reads the verdict; a silent hang would cost the whole GPU call
        verdict = "exception: %r" % (ex,)
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write(verdict)
    try:  # no rank tears its sockets down while another is still inside the last barrier (see tests/test_distributed.py)
        import time
        open(os.path.join(out_dir, "left%d" % rank), "w").write("x")
        t0 = time.time()
        while time.time() - t0 < 60 and not all(os.path.exists(os.path.join(out_dir, "left%d" % r)) for r in range(world)):
            time.sleep(0.01)
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.parametrize("world,spec,fused,balanced,need", [
    (2, "synthetic:poisson2d:300", True, False, None),       # value dictionary + lane-per-row tiles, row sums forwarded by the kernel
    (3, "synthetic:poisson2d:300,1", True, False, None),     # the same without a dictionary, three ranks, a short last block
    (2, "synthetic:queen:20,15,10", True, False, None),      # narrow tiles, several lanes per row
    (2, "synthetic:banded:60000,13", True, False, None),     # shifted tiles with x windows (the XW kernel variant)
    (2, "synthetic:kkt:44,50", True, False, None),           # segment windows + the launch over the leftover tiles, both forwarding
    (2, "synthetic:poisson2d:300", False, False, None),      # pushed by a second launch
    (3, "synthetic:webbase:30000,100000,300,75", True, True, None),  # skewed rows: balanced tiles (no forwarding variant) + uneven blocks
    # round 5's tile classes, each storing through the same forwarding y_store
    (2, "synthetic:queen:20,15,10,3,20,97", True, False, "masked_block_tiles"),  # 3 x 3 blocks with entries missing, odd nodes
    (2, "synthetic:poisson3d:40", True, False, "stencil_mask_tiles"),            # masked stencil tiles, values read
    (2, "synthetic:poisson3d:40,1", True, False, "stencil_mask_tiles"),          # ... and with the value dictionary
    (2, "synthetic:queen:20,15,10,3,0,0,2", True, False, "group_tiles"),         # 2 unknowns per node: group tiles (one column list per pair of rows)
    (3, "synthetic:queen:20,15,10,3,0,0,4", True, False, "group_tiles"),         # 4 per node, three ranks
    (2, "synthetic:random:3000,600", True, False, "multi_window_tiles"),         # rows of 600 entries sharing a tile through LDS
    (2, "synthetic:random:40000,600", True, False, "multi_window_tiles"),        # ... and, the matrix large enough, in registers
])
def test_peer_stores_between_processes_on_one_device(tmp_path, world, spec, fused, balanced, need):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), spec, fused, balanced, need), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "ok"
