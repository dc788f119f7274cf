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
multiply k+1 runs on the compute stream: the segment is snapshotted into a send buffer
(the kernels accumulate into y_local, so it cannot be sent from in place), and the next
snapshot waits for the previous gather.  Every multiply is still gathered exactly once.

The local multiply is injected (``local_spmv``) so the partition / gather logic can
be exercised with gloo on CPU tensors; the product constructor
``DistributedCsrSpmv.on_gpu`` wires in the HIP path and has no other option.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi, partition


class DistributedCsrSpmv:
    def __init__(self, rows, cols, rank, world, device, local_rows, local_spmv, group=None, overlap=False,
                 ranges=None):
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
        # padded local segment and the gathered vector (world * chunk >= rows)
        self.y_local = torch.zeros(self.chunk, dtype=torch.float64, device=device)
        self.y_full = torch.zeros(self.chunk * world, dtype=torch.float64, device=device)
        self.overlap = overlap
        self.send_buf = torch.zeros(self.chunk, dtype=torch.float64, device=device) if overlap else None
        self.pending = None

    @classmethod
    def on_gpu(cls, rows, cols, rank, world, device, p_local, c_local, v_local, x_host,
               algorithm=capi.CSR_AUTO, lanes_per_row=0, flags=0, group=None, overlap=False, ranges=None):
        """Product path: local slice uploaded to `device`, multiplied by the HIP kernel
        on torch's current stream.  Raises if the HIP library or the GPU is missing."""
        local_rows = len(p_local) - 1
        plan = capi.CsrPlan(local_rows, cols, p_local, algorithm, lanes_per_row, flags)
        tp = torch.from_numpy(np.ascontiguousarray(p_local, dtype=np.int32)).to(device)
        tc = torch.from_numpy(np.ascontiguousarray(c_local, dtype=np.int32)).to(device)
        tv = torch.from_numpy(np.ascontiguousarray(v_local, dtype=np.float64)).to(device)
        tx = torch.from_numpy(np.ascontiguousarray(x_host, dtype=np.float64)).to(device)
        if not (flags & capi.FLAG_NO_INDEX_COMPRESSION):
            plan.compress(tc.data_ptr(), torch.cuda.current_stream().cuda_stream)
            plan.repack(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), torch.cuda.current_stream().cuda_stream)

        def local_spmv(y_local):
            plan.spmv(tp.data_ptr(), tc.data_ptr(), tv.data_ptr(), tx.data_ptr(), y_local.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)

        self = cls(rows, cols, rank, world, device, local_rows, local_spmv, group, overlap, ranges)
        self.plan = plan
        self._keep = (tp, tc, tv, tx)
        return self

    def multiply_local(self):
        """y_local += A_local @ x (enqueue only)."""
        self.local_spmv(self.y_local)

    def gather(self):
        """The one collective of the path: equal-count all-gather of the y segments."""
        if self.world == 1 and not dist.is_initialized():
            self.y_full.copy_(self.y_local)
        else:
            dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group)

    def gather_async(self):
        """Snapshot y_local and start its all-gather without waiting for it; the previous one
        must have finished with the send buffer first."""
        if self.pending is not None:
            self.pending.wait()
        self.send_buf.copy_(self.y_local)
        self.pending = dist.all_gather_into_tensor(self.y_full, self.send_buf, group=self.group, async_op=True)

    def finish(self):
        """Wait for the last outstanding gather (no-op without overlap)."""
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

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
