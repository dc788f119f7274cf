"""Row-partitioned SpMV across ranks: one process per GPU, one all-gather of the
y segments per multiply (RCCL over xGMI when the process group is "nccl").

    rank g:   y_g += A[rows_g, :] @ x            (HIP kernel, local rows only)
    all:      y = all_gather(y_0, ..., y_{G-1})  (the only collective on the path)

The default partition is the reference's static row chunking (partition.row_range).
Segments are padded to the common chunk length so the collective is a plain
equal-count all-gather; because chunks are contiguous and only the last one is
short, the first `rows` entries of the gathered buffer are y itself.  Any other
contiguous split (``ranges``, e.g. partition.nnz_balanced_ranges for matrices with
uneven rows) pads every segment to the longest one; y() then drops the padding.

With ``overlap=True`` the gather of multiply k runs on the collective's stream while
multiply k+1 runs on the compute stream.  The kernels accumulate (y += A*x), so the
segment that is being sent must not be the one the next multiply writes:

* ``pingpong=True`` (default): two gathered vectors alternate, each holding this rank's segment in
  its own slot, so every all-gather is IN PLACE (nothing is copied for the rank's own rows).
  Multiply k reads the old segment from one and writes the new one to the other (spmv_hip_csr_spmv_out:
  y_out = y_in + A*x), gather k sends what multiply k wrote, multiply k+1 only READS that
  buffer and writes the first one again -- which gather k-1 must have finished sending, the
  only wait on the compute stream.  No copy.
* ``pingpong=False``: one buffer, snapshotted into a send buffer before every gather
  (one extra read + write of the segment per step; round 1's scheme).

Every multiply is still gathered exactly once.

The local multiply is injected (``local_spmv``, optionally ``local_spmv_out``) so the
partition / gather logic can be exercised with gloo on CPU tensors; the product constructor
``DistributedCsrSpmv.on_gpu`` wires in the HIP path and has no other option.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi, partition


def upload_and_plan(local_rows, cols, device, p_local, c_local, v_local, x_host, algorithm, lanes_per_row, flags):
    """This rank's rows on its GPU and the launch plan for them (tile classes, column panels, value dictionary)."""
    plan = capi.CsrPlan(local_rows, cols, p_local, algorithm, lanes_per_row, flags)
    tp = torch.from_numpy(np.ascontiguousarray(p_local, dtype=np.int32)).to(device)
    tc = torch.from_numpy(np.ascontiguousarray(c_local, dtype=np.int32)).to(device)
    tv = torch.from_numpy(np.ascontiguousarray(v_local, dtype=np.float64)).to(device)
    tx = torch.from_numpy(np.ascontiguousarray(x_host, dtype=np.float64)).to(device)
    if not (flags & capi.FLAG_NO_INDEX_COMPRESSION):
        stream = torch.cuda.current_stream().cuda_stream
        plan.compress(tc.data_ptr(), stream)
        # column panels keep a snapshot of the values: the operator holds the tensors it was taken from for as
        # long as the plan lives, so it cannot go stale
        plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), stream)
        # ... and so does the value dictionary of a matrix with few distinct values (same lifetime argument)
        plan.index_values(tv.data_ptr(), stream)
    return plan, tp, tc, tv, tx


def raw_stream_getter():
    """torch's current stream of a device as a raw hipStream_t, looked up per call with one C call."""
    fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if fn is None:
        def fn(i):
            return torch.cuda.current_stream(i).cuda_stream
    return fn


class DistributedCsrSpmv:
    def __init__(self, rows, cols, rank, world, device, local_rows, local_spmv, group=None, overlap=False,
                 ranges=None, local_spmv_out=None, pingpong=True, full=None):
        self.rows, self.cols = rows, cols
        self.rank, self.world = rank, world
        if ranges is None:
            ranges = [partition.row_range(rows, g, world) for g in range(world)]
            self.chunk = partition.row_chunk(rows, world)
        else:
            ranges = [(int(b), int(e)) for b, e in ranges]
            assert len(ranges) == world and ranges[0][0] == 0 and ranges[-1][1] == rows
            assert all(ranges[g][1] == ranges[g + 1][0] for g in range(world - 1))
            self.chunk = max(1, max(e - b for b, e in ranges))
        self.ranges = ranges
        # the gathered buffer is y itself when every segment but the last fills its chunk
        self.packed = all(e - b == self.chunk for b, e in ranges[:-1] if e > b) and \
            all(b == g * self.chunk or e == b for g, (b, e) in enumerate(ranges))
        self.begin, self.end = ranges[rank]
        assert local_rows == self.end - self.begin
        self.device = device
        self.group = group
        self.local_spmv = local_spmv
        if local_spmv_out is None:
            def local_spmv_out(y_in, y_out):  # generic: copy, then accumulate in place
                y_out.copy_(y_in)
                local_spmv(y_out)
        self.local_spmv_out = local_spmv_out
        self.overlap = overlap
        # two alternating vectors only where a gather can be in flight: one rank without a process group
        # multiplies in place (measured: a separate y_out costs a cache-resident multiply 4 of its 27 us)
        self.pingpong = bool(pingpong and overlap and (world > 1 or dist.is_initialized()))
        # The gathered vector(s) (world * chunk >= rows); this rank's padded segment LIVES INSIDE at
        # [rank * chunk, (rank + 1) * chunk), so the all-gather is in place (send buffer = the rank's
        # slot of the receive buffer): the collective moves only what comes from other ranks and the
        # rank's own segment is never copied.  Two such vectors alternate with pingpong.
        if full is not None:  # the caller's vector(s): memory the other ranks can store into (PeerCsrSpmv)
            assert len(full) == (2 if self.pingpong else 1) and all(f.numel() == self.chunk * world for f in full)
            self.full = list(full)
        else:
            self.full = [torch.zeros(self.chunk * world, dtype=torch.float64, device=device) for _ in range(2 if self.pingpong else 1)]
        self.seg = [f[rank * self.chunk:(rank + 1) * self.chunk] for f in self.full]
        self.cur = 0  # the buffer that holds the current y_local (and, after its gather, the current y)
        self.send_buf = None
        if overlap and not self.pingpong:
            # snapshot scheme: the segment keeps accumulating while an older copy of it is being gathered, so it
            # must NOT live inside the receive buffer (the gather would write the old copy over it)
            self.seg = [torch.zeros(self.chunk, dtype=torch.float64, device=device)]
            self.send_buf = torch.zeros(self.chunk, dtype=torch.float64, device=device)
        self.inflight = []  # outstanding gathers, oldest first
        self.collective = True  # the gather is a collective call (PeerCsrSpmv: stores into the other ranks' vectors)

    @classmethod
    def on_gpu(cls, rows, cols, rank, world, device, p_local, c_local, v_local, x_host,
               algorithm=capi.CSR_AUTO, lanes_per_row=0, flags=0, group=None, overlap=False, ranges=None,
               pingpong=True, uploaded=None):
        """Product path: local slice uploaded to `device`, multiplied by the HIP kernel
        on torch's current stream.  Raises if the HIP library or the GPU is missing.
        `uploaded`: the (plan, row_ptr, column, value, x) of another operator on the same rows, to share."""
        local_rows = len(p_local) - 1
        plan, tp, tc, tv, tx = uploaded or upload_and_plan(local_rows, cols, device, p_local, c_local, v_local, x_host, algorithm,
                                                           lanes_per_row, flags)

        # The launch itself is one foreign call with everything resolved beforehand (device addresses, the plan
        # handle): a rank-local multiply of a partitioned matrix lasts 20-30 us, and ten attribute look-ups per
        # step on the host would leave the launch queue empty in between.  The STREAM is looked up on every call:
        # the collectives order themselves against torch's current stream at call time, so the multiply must be
        # enqueued on that same stream or a gather could send a segment that is not finished
        # (torch._C._cuda_getCurrentRawStream: one C call, no Python stream object).
        fn, handle = plan.lib.spmv_hip_csr_spmv_out, plan.h
        fixed = (tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr())
        dev_index = torch.device(device).index
        if dev_index is None:
            dev_index = torch.cuda.current_device()
        raw_stream = raw_stream_getter()
        # device addresses are cached for the segment views only (they live as long as this object, so neither
        # their addresses nor their ids can be reused); any other tensor is asked for its address every time
        seg_addr = {}

        def local_spmv_out(y_in, y_out):
            a_in = seg_addr.get(id(y_in)) or y_in.data_ptr()
            a_out = seg_addr.get(id(y_out)) or y_out.data_ptr()
            rc = fn(handle, fixed[0], fixed[1], fixed[2], fixed[3], a_in, a_out, raw_stream(dev_index))
            if rc != 0:
                capi.check(rc)

        def local_spmv(y_local):
            local_spmv_out(y_local, y_local)

        self = cls(rows, cols, rank, world, device, local_rows, local_spmv, group, overlap, ranges,
                   local_spmv_out=local_spmv_out, pingpong=pingpong)
        self.plan = plan
        self._keep = (tp, tc, tv, tx)
        self.uploaded = (plan, tp, tc, tv, tx)
        self._seg_keep = list(self.seg)  # pins the ids the cache is keyed by
        for t in self._seg_keep:
            seg_addr[id(t)] = t.data_ptr()
        return self

    @property
    def y_local(self):
        """The padded segment holding this rank's current rows of y (a view into y_full)."""
        return self.seg[self.cur]

    @property
    def y_full(self):
        """The vector the current segment is gathered into."""
        return self.full[self.cur]

    def zero(self):
        self.finish()
        for f in self.full + self.seg:
            f.zero_()

    def multiply_local(self):
        """y_local += A_local @ x (enqueue only)."""
        if self.pingpong:
            # the buffer about to be written was the source of the gather before last
            while len(self.inflight) >= 2:
                self.inflight.pop(0).wait()
            src, dst = self.seg[self.cur], self.seg[1 - self.cur]
            self.local_spmv_out(src, dst)
            self.cur = 1 - self.cur
        else:
            self.local_spmv(self.seg[0])

    def gather(self):
        """The one collective of the path: equal-count all-gather of the y segments."""
        self.finish()
        if self.world > 1 or dist.is_initialized():
            dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group)  # in place
        # one rank and no process group: the segment already is the gathered vector

    def gather_async(self):
        """Start the all-gather of the current segment without waiting for it."""
        if self.pingpong:
            self.inflight.append(dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group, async_op=True))
            return
        # one buffer: the previous gather must have finished with the send buffer, then snapshot
        self.finish()
        self.send_buf.copy_(self.seg[0])
        self.inflight.append(dist.all_gather_into_tensor(self.y_full, self.send_buf, group=self.group, async_op=True))

    def finish(self):
        """Wait for the outstanding gathers (no-op without overlap)."""
        while self.inflight:
            self.inflight.pop(0).wait()

    def step(self):
        self.multiply_local()
        if self.overlap and (self.world > 1 or dist.is_initialized()):
            self.gather_async()
        else:
            self.gather()

    def y(self):
        """The assembled y: the first `rows` entries of the gathered buffer, or, for a split with
        uneven segments, the segments without their padding."""
        self.finish()
        if self.packed:
            return self.y_full[:self.rows]
        return torch.cat([self.y_full[g * self.chunk: g * self.chunk + (e - b)] for g, (b, e) in enumerate(self.ranges)])
