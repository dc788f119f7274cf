"""Row-partitioned SpMV across ranks WITHOUT a collective: every rank stores its rows of y straight into the
other ranks' vectors (one process per GPU; xGMI is point to point, so a direct all-gather is G - 1 copies of the
segment leaving over G - 1 links at once -- SURVEY section 5, last row).

    every rank g:   y_g += A[rows_g, :] @ x     and the same doubles into slot g of every other rank's y

The partition is DistributedCsrSpmv's (the reference's static row chunks, src/matrix/csr-matrix.cpp:77-95, or any
contiguous split).  Two ways the doubles travel, both through include/spmv_hip.h:

* fused (default): spmv_hip_csr_spmv_out_peers -- the multiply kernel stores every row sum into all G copies as
  each tile finishes, so the transfer overlaps the SAME multiply and a step is one launch;
* push: the multiply, then spmv_hip_peer_push of the segment on the same stream (plans whose kernels have no
  forwarding variant do this by themselves; ``fused=False`` forces it for comparison).

Nothing is received by a kernel.  Row blocks are disjoint, so ranks may drift apart without ever touching the
same doubles; a rank's vector is complete -- and may be read -- after ``finish()``: every rank synchronises its
device, then the ranks meet at a barrier.  It stays valid until ANY rank multiplies again (that rank's stores land in
it without asking): a reader calls ``finish()`` once more when it is done, before the ranks go on -- the place an
all-gather's implicit hand-shake would have taken.  The vectors live in device memory the other processes map with HIP's
inter-process handles (``HipPeerVectors``); the partition / completion logic is exercised on CPU with vectors in
POSIX shared memory (``tests/test_distributed.py``).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .distributed import DistributedCsrSpmv, raw_stream_getter, upload_and_plan


class PeerUnavailable(RuntimeError):
    """Raised on EVERY rank (the ranks agree first) when the vectors cannot be shared: no inter-process handles on this
    driver, or no peer access between two devices.  The caller keeps the collective gather."""


class _DeviceArray:
    """Raw device memory as something torch.as_tensor can alias (no copy)."""

    def __init__(self, addr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (addr, False), "version": 2}


class HipPeerVectors:
    """This rank's copy of the gathered y in memory the other ranks can store into, and theirs mapped here."""

    def __init__(self, n, rank, world, device, group=None):
        self.rank, self.world, self.group, self.n = rank, world, group, n
        self.addr, self.peer_addr, self.closed = None, {}, False
        shared = world > 1 or dist.is_initialized()
        handle, why = None, None
        try:
            self.addr, handle = capi.ipc_alloc(8 * n)
        except capi.SpmvHipError as e:
            why = str(e)
        if shared:
            # every step is agreed on by all ranks before anyone depends on it: a rank that cannot share must not
            # leave the others waiting in a collective
            handles = [None] * world
            dist.all_gather_object(handles, handle, group=group)
            if any(h is None for h in handles):
                self._abandon("a rank could not allocate or export its vector" + (": " + why if why else ""))
            try:
                for h in range(world):
                    if h != rank:
                        self.peer_addr[h] = capi.ipc_open(handles[h])
            except capi.SpmvHipError as e:
                why = str(e)
            ok = [None] * world
            dist.all_gather_object(ok, why is None, group=group)
            if not all(ok):
                self._abandon("a rank could not map another rank's vector" + (": " + why if why else ""))
        elif handle is None:
            raise PeerUnavailable(why)
        self._holder = _DeviceArray(self.addr, n)
        self.own = torch.as_tensor(self._holder, device=device)
        assert self.own.data_ptr() == self.addr and self.own.dtype == torch.float64

    def _abandon(self, why):
        for a in self.peer_addr.values():
            capi.ipc_close(a)
        self.peer_addr = {}
        dist.barrier(group=self.group)  # every mapping is closed before anybody frees
        if self.addr is not None:
            capi.ipc_free(self.addr)
            self.addr = None
        self.closed = True
        raise PeerUnavailable(why)

    def close(self):
        """Collective: nobody frees memory another rank still has mapped."""
        if self.closed:
            return
        self.closed = True
        torch.cuda.synchronize()
        if self.peer_addr:
            dist.barrier(group=self.group)
        for a in self.peer_addr.values():
            capi.ipc_close(a)
        if self.peer_addr:
            dist.barrier(group=self.group)
        self.own = None
        capi.ipc_free(self.addr)


class PeerCsrSpmv(DistributedCsrSpmv):
    """DistributedCsrSpmv whose gather is stores into the other ranks' vectors.

    ``local_spmv(y_segment)`` accumulates this rank's rows into its segment of ``vectors.own`` (and, in the fused
    scheme, into the peers' copies); ``deliver()`` -- if given -- pushes the segment to the peers afterwards;
    ``sync()`` waits for this rank's device (a no-op for the CPU test double)."""

    def __init__(self, rows, cols, rank, world, device, local_rows, local_spmv, vectors, group=None, ranges=None,
                 deliver=None, sync=None):
        super().__init__(rows, cols, rank, world, device, local_rows, local_spmv, group, overlap=False, ranges=ranges,
                         full=[vectors.own])
        self.vectors = vectors
        self.deliver = deliver
        self.sync = sync or (lambda: None)
        self.collective = False

    @classmethod
    def on_gpu(cls, rows, cols, rank, world, device, p_local, c_local, v_local, x_host, algorithm=capi.CSR_AUTO,
               lanes_per_row=0, flags=0, group=None, ranges=None, fused=True, uploaded=None):
        """Product path: this rank's rows on `device`, the HIP kernel on torch's current stream, the y vectors in
        inter-process device memory.  Raises if the HIP library, the GPU or peer access is missing."""
        from . import partition
        local_rows = len(p_local) - 1
        plan, tp, tc, tv, tx = uploaded or upload_and_plan(local_rows, cols, device, p_local, c_local, v_local, x_host, algorithm,
                                                           lanes_per_row, flags)
        if ranges is None:
            chunk = partition.row_chunk(rows, world)
        else:
            chunk = max(1, max(int(e) - int(b) for b, e in ranges))
        vectors = HipPeerVectors(chunk * world, rank, world, device, group)
        peers = [vectors.peer_addr[h] + 8 * rank * chunk for h in sorted(vectors.peer_addr)]  # my slot in their vectors
        peer_arr = (C.c_void_p * max(1, len(peers)))(*peers)
        npeers = len(peers)
        lib, handle = plan.lib, plan.h
        fixed = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr())
        dev_index = torch.device(device).index
        if dev_index is None:
            dev_index = torch.cuda.current_device()
        raw_stream = raw_stream_getter()
        seg_addr = vectors.addr + 8 * rank * chunk
        was_fused = C.c_int(0)
        fn_peers, fn_plain, fn_push = lib.spmv_hip_csr_spmv_out_peers, lib.spmv_hip_csr_spmv_out, lib.spmv_hip_peer_push

        if fused:
            def local_spmv(y_local):  # one call: the multiply forwards its row sums, or pushes the segment itself
                rc = fn_peers(handle, fixed[0], fixed[1], fixed[2], fixed[3], seg_addr, seg_addr, peer_arr, npeers,
                              C.byref(was_fused), raw_stream(dev_index))
                if rc != 0:
                    capi.check(rc)
            deliver = None
        else:
            def local_spmv(y_local):
                rc = fn_plain(handle, fixed[0], fixed[1], fixed[2], fixed[3], seg_addr, seg_addr, raw_stream(dev_index))
                if rc != 0:
                    capi.check(rc)

            def deliver():
                rc = fn_push(seg_addr, peer_arr, npeers, local_rows, raw_stream(dev_index))
                if rc != 0:
                    capi.check(rc)
            if npeers == 0:
                deliver = None

        self = cls(rows, cols, rank, world, device, local_rows, local_spmv, vectors, group, ranges, deliver,
                   sync=torch.cuda.synchronize)
        assert self.seg[0].data_ptr() == seg_addr and self.chunk == chunk
        self.plan = plan
        self._keep = (tp, tc, tv, tx)
        self.uploaded = (plan, tp, tc, tv, tx)
        self._was_fused = was_fused
        self.scheme = "fused" if fused else "push"
        return self

    @property
    def fused(self):
        """Did the last multiply forward its row sums itself (True) or was the segment pushed by a second launch?"""
        return bool(getattr(self, "_was_fused", C.c_int(0)).value)

    def multiply_local(self):
        """y_local += A_local @ x, and the new segment on its way into the other ranks' vectors (enqueue only)."""
        self.local_spmv(self.seg[0])
        if self.deliver is not None:
            self.deliver()

    def gather(self):
        """Nothing to do: the stores are on their way since multiply_local; finish() says when they have all landed."""

    gather_async = gather

    def finish(self):
        """Every rank's device done with what it was given, then the ranks meet: the vectors are complete."""
        self.sync()
        if self.world > 1 or dist.is_initialized():
            dist.barrier(group=self.group)

    def zero(self):
        self.finish()  # nobody is storing into anybody's vector any more
        self.full[0].zero_()
        self.finish()  # ... and nobody starts again before every vector is zero

    def step(self):
        self.multiply_local()

    def close(self):
        self.finish()
        self.seg = self.full = None
        self.vectors.close()
