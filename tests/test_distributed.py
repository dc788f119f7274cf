"""The N > 1 path: row-range partition + one all-gather of the y segments, exercised with
world_size 2 and 3 over gloo on CPU tensors.  The local multiply is injected here (the oracle
computes each rank's rows); on the GPU box the same class is wired to the HIP kernel by
DistributedCsrSpmv.on_gpu, which bench.py uses."""
import os
import socket

import numpy as np
import pytest

from spmv_amd import partition, synth


def test_row_range_is_the_reference_rule():
    # src/matrix/csr-matrix.cpp:77-84: chunk = ceil(rows/T); [min(rows,t*chunk), min(rows,(t+1)*chunk))
    for rows, parts in ((10, 3), (16777216, 8), (7, 8), (0, 4), (1000, 1), (1138, 2)):
        chunk = (rows + parts - 1) // parts
        covered = []
        for g in range(parts):
            b, e = partition.row_range(rows, g, parts)
            assert (b, e) == (min(rows, g * chunk), min(rows, (g + 1) * chunk))
            covered += list(range(b, e))
        assert covered == list(range(rows))
        assert partition.row_chunk(rows, parts) * parts >= rows


def test_nnz_balanced_ranges_and_slices():
    rows, cols, p, c, v = synth.powerlaw(5000, 5000, seed=2)
    for parts in (1, 2, 8):
        r = partition.nnz_balanced_ranges(p, parts)
        assert r[0][0] == 0 and r[-1][1] == rows and all(r[k][1] == r[k + 1][0] for k in range(parts - 1))
        nnz = [int(p[e] - p[b]) for b, e in r]
        assert sum(nnz) == int(p[-1])
        assert max(nnz) <= int(p[-1]) / parts + np.diff(p).max() + 1
    b, e = 1200, 3400
    ps, cs, vs = partition.csr_slice(p, c, v, b, e)
    assert ps[0] == 0 and ps[-1] == p[e] - p[b] and len(ps) == e - b + 1
    assert np.array_equal(cs, c[p[b]:p[e]]) and np.array_equal(vs, v[p[b]:p[e]])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, steps, out_dir, overlap=False, balanced=False, pingpong=True):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "spmv-cache-trace_amd", "python"))
    import torch
    import torch.distributed as dist
    import oracle_py
    from spmv_amd import partition, synth
    from spmv_amd.distributed import DistributedCsrSpmv

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = oracle_py.Oracle()
    if balanced:  # uneven rows: split by stored entries, segments of different lengths
        rows, cols, p, c, v = synth.powerlaw(700, 700, seed=4, max_len=300)
        ranges = partition.nnz_balanced_ranges(p, world)
    else:
        rows, cols, p, c, v = synth.stencil27_like(11, 9, 7, seed=5)  # 693 rows: not divisible by 2 or 3... evenly
        ranges = None
    x = synth.x_vector(cols, seed=9)
    b, e = ranges[rank] if balanced else partition.row_range(rows, rank, world)
    pl, cl, vl = partition.csr_slice(p, c, v, b, e)

    def local_spmv(y_local):  # test double for the HIP kernel: the oracle on this rank's rows
        y = y_local.numpy()
        y[:e - b] = O.csr_spmv(e - b, pl, cl, vl, x, y=y[:e - b])

    op = DistributedCsrSpmv(rows, cols, rank, world, torch.device("cpu"), e - b, local_spmv, overlap=overlap,
                            ranges=ranges, pingpong=pingpong)
    assert op.pingpong == (overlap and pingpong) and len(op.full) == (2 if op.pingpong else 1)
    for _ in range(steps):
        op.step()
    want = O.csr_spmv(rows, p, c, v, x, runs=steps)
    got = op.y().numpy()  # y() waits for the outstanding gather
    ok = np.array_equal(got, want) and (balanced or op.y_full.numel() == partition.row_chunk(rows, world) * world)
    if balanced:
        ok = ok and len({e2 - b2 for b2, e2 in ranges}) > 1  # really uneven
        if world == 3:
            ok = ok and not op.packed  # a short segment in front of a longer one: y() has to drop padding
    # every rank must hold the whole y after the gather
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap,balanced,pingpong", [(2, False, False, True), (3, False, False, True),
                                                             (2, True, False, True), (2, True, False, False),
                                                             (3, False, True, True), (2, True, True, True),
                                                             (3, True, True, False)])
def test_partitioned_spmv_with_allgather_gloo(tmp_path, world, overlap, balanced, pingpong):
    """overlap + pingpong: two segment buffers alternate (y_out = y_in + A*x), several gathers may be in
    flight; overlap without it: one buffer and a snapshot copy; 5 steps so both buffers are reused."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 5, str(tmp_path), overlap, balanced, pingpong), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "ok"
