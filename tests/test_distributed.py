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
                                                             (3, True, True, False),
                                                             # eight ranks, what the first 8-GPU run will launch: 693 rows in chunks of 87, the last one 84
                                                             (8, True, False, True), (8, False, True, True)])
def test_partitioned_spmv_with_allgather_gloo(tmp_path, world, overlap, balanced, pingpong):
    """overlap + pingpong: two segment buffers alternate (y_out = y_in + A*x), several gathers may be in
    flight; overlap without it: one buffer and a snapshot copy; 5 steps so both buffers are reused."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 5, str(tmp_path), overlap, balanced, pingpong), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() == "ok"


def _leave_together(out_dir, rank, world, timeout=60.0):
    """A rank that tears its sockets down while another is still inside the last barrier resets that one's read (gloo):
    every rank says on the file system that it is out of its last collective, and waits for the others to say so."""
    import time
    open(os.path.join(out_dir, "left%d" % rank), "w").write("x")
    t0 = time.time()
    while time.time() - t0 < timeout and not all(os.path.exists(os.path.join(out_dir, "left%d" % r)) for r in range(world)):
        time.sleep(0.01)


class _ShmVectors:
    """Test double of peer.HipPeerVectors: every rank's copy of y in POSIX shared memory, the others' mapped."""

    def __init__(self, n, rank, world, tag):
        import torch
        import torch.distributed as dist
        from multiprocessing import shared_memory
        self.rank, self.world = rank, world
        self.mine = shared_memory.SharedMemory(create=True, size=8 * n, name="%s_%d" % (tag, rank))
        np.ndarray((n,), dtype=np.float64, buffer=self.mine.buf)[:] = 0.0
        dist.barrier()
        self.theirs = {h: shared_memory.SharedMemory(name="%s_%d" % (tag, h)) for h in range(world) if h != rank}
        self.views = {h: np.ndarray((n,), dtype=np.float64, buffer=s.buf) for h, s in self.theirs.items()}
        self.own = torch.from_numpy(np.ndarray((n,), dtype=np.float64, buffer=self.mine.buf))

    def close(self):
        import torch.distributed as dist
        self.own = None
        self.views = {}
        dist.barrier()
        for s in self.theirs.values():
            s.close()
        dist.barrier()
        self.mine.close()
        self.mine.unlink()


def _peer_worker(rank, world, port, steps, out_dir, balanced):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "spmv-cache-trace_amd", "python"))
    import torch
    import torch.distributed as dist
    import oracle_py
    from spmv_amd import partition, synth
    from spmv_amd.peer import PeerCsrSpmv

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = oracle_py.Oracle()
    if balanced:
        rows, cols, p, c, v = synth.powerlaw(700, 700, seed=4, max_len=300)
        ranges = partition.nnz_balanced_ranges(p, world)
        chunk = max(e - b for b, e in ranges)
    else:
        rows, cols, p, c, v = synth.stencil27_like(11, 9, 7, seed=5)
        ranges = None
        chunk = partition.row_chunk(rows, world)
    x = synth.x_vector(cols, seed=9)
    b, e = ranges[rank] if balanced else partition.row_range(rows, rank, world)
    pl, cl, vl = partition.csr_slice(p, c, v, b, e)
    vec = _ShmVectors(chunk * world, rank, world, "spmvt%d" % port)

    def local_spmv(y_local):  # the oracle on this rank's rows, in place in its slot of its own vector
        y = y_local.numpy()
        y[:e - b] = O.csr_spmv(e - b, pl, cl, vl, x, y=y[:e - b])

    def deliver():  # what spmv_hip_peer_push / the fused kernel do: my slot into everybody else's vector
        mine = vec.own.numpy()[rank * chunk: rank * chunk + (e - b)]
        for h, view in vec.views.items():
            view[rank * chunk: rank * chunk + (e - b)] = mine

    op = PeerCsrSpmv(rows, cols, rank, world, torch.device("cpu"), e - b, local_spmv, vec, ranges=ranges, deliver=deliver)
    ok = not op.collective and not op.pingpong and len(op.full) == 1 and op.full[0] is vec.own
    for _ in range(steps):
        op.step()  # no collective inside: ranks may drift apart here
    want = O.csr_spmv(rows, p, c, v, x, runs=steps)
    ok = ok and np.array_equal(op.y().numpy(), want)  # y() = finish(): complete on EVERY rank
    op.zero()  # (begins with finish(): nobody multiplies on while another rank still compares)
    ok = ok and float(op.y().abs().max()) == 0.0
    op.finish()  # a vector may be read until ANY rank multiplies again: tell the others this rank is done reading
    op.step()
    ok = ok and np.array_equal(op.y().numpy(), O.csr_spmv(rows, p, c, v, x))
    open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("ok" if ok else "mismatch")
    try:
        op.close()  # ends with two barriers
    except Exception:  # say which rank failed first: the others only see their connections drop
        import traceback
        open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("close failed:\n" + traceback.format_exc())
    _leave_together(out_dir, rank, world)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (3, False), (3, True), (8, False), (8, True)])
def test_partitioned_spmv_with_peer_stores_gloo(tmp_path, world, balanced):
    """The gather as stores into the other ranks' vectors (spmv_amd/peer.py): partition, slots, completion by
    finish() (sync + barrier), zero(), uneven segments -- on CPU with the vectors in shared memory."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_peer_worker, args=(world, port, 4, str(tmp_path), balanced), nprocs=world, join=True)
    verdicts = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() for r in range(world)]
    assert verdicts == ["ok"] * world, verdicts
