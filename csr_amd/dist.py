"""
Row-partitioned SpMV across the GPUs of one node: one process per GPU, torch.distributed
for the exchange (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

This is the parallel form of what the reference does sequentially when a matrix exceeds a
kernel's max_nnz (csr/csr.py:584-590): split A into contiguous row ranges balanced by nnz
(split points = searchsorted(rowptrs, g*nnz/G), the primitive of _shard_rows, csr/csr.py:609),
multiply every range against the SAME x, and concatenate the pieces (np.concatenate, :590).
Here every rank owns one range, x is replicated, and the concatenation is the one exchange
step of the path:

  allgather  (default) every rank contributes its y slice; slices are padded to the longest
             one so a single all_gather_into_tensor moves them (xGMI is point-to-point: the
             7 peers' slices arrive over 7 links concurrently), then unpadded into y by one
             concatenation kernel.
  allreduce  the form BASELINE.json's north_star names: each rank writes its slice into a
             zeroed full-length y and the ranks sum.  Same result (the slices are disjoint, so
             every sum has one non-zero term and is exact), about twice the bytes per link.

torch is plumbing here (device buffers + the collective); the product kernels run behind
`local_spmv`, a callable that writes y[r0:r1] = A[r0:r1, :] x into the buffer it is given.
"""
import torch
import torch.distributed as dist


class RowPartitionedSpMV:
    def __init__(self, bounds, rank, world, local_spmv, device, mode='allgather', group=None):
        """
        bounds: world+1 row indices (rank g owns rows bounds[g]:bounds[g+1]).
        local_spmv(x, out): computes this rank's rows into `out` (a float64 tensor of
        bounds[rank+1]-bounds[rank] entries on `device`); asynchronous on the current stream.
        """
        assert len(bounds) == world + 1 and mode in ('allgather', 'allreduce')
        self.bounds = [int(b) for b in bounds]
        self.rank, self.world, self.mode, self.group = rank, world, mode, group
        self.local_spmv = local_spmv
        self.nrows = self.bounds[-1]
        self.r0, self.r1 = self.bounds[rank], self.bounds[rank + 1]
        self.lens = [self.bounds[g + 1] - self.bounds[g] for g in range(world)]
        self.y = torch.zeros(self.nrows, dtype=torch.float64, device=device)
        self.timing = False            # set True to record device events around the local product
        self._ev = []
        if world > 1 and mode == 'allgather':
            self.maxlen = max(self.lens)
            self.loc = torch.zeros(self.maxlen, dtype=torch.float64, device=device)
            self.gath = torch.zeros(world * self.maxlen, dtype=torch.float64, device=device)
            # the slices of `gath` that make up y, in rank order: unpadded by ONE concatenation kernel
            self.pieces = [self.gath[g * self.maxlen:g * self.maxlen + self.lens[g]] for g in range(world)
                           if self.lens[g]]

    def _local(self, x, out):
        if self.timing and out.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.local_spmv(x, out)
            e1.record()
            self._ev.append((e0, e1))
        else:
            self.local_spmv(x, out)

    def compute_ms(self):
        "mean device time of the local product over the steps timed so far (call after a synchronize)"
        if not self._ev:
            return 0.0
        ms = sum(a.elapsed_time(b) for a, b in self._ev) / len(self._ev)
        self._ev = []
        return ms

    def step(self, x):
        "y = A x, complete on every rank; returns the (reused) y tensor"
        if self.world == 1:
            self._local(x, self.y)
            return self.y
        if self.mode == 'allgather':
            self._local(x, self.loc[:self.r1 - self.r0])
            dist.all_gather_into_tensor(self.gath, self.loc, group=self.group)
            if self.pieces:
                torch.cat(self.pieces, out=self.y)
            return self.y
        # allreduce: zero what the previous step left in the other ranks' slices
        if self.r0 > 0:
            self.y[:self.r0].zero_()
        if self.r1 < self.nrows:
            self.y[self.r1:].zero_()
        self._local(x, self.y[self.r0:self.r1])
        dist.all_reduce(self.y, op=dist.ReduceOp.SUM, group=self.group)
        return self.y


def hip_local_spmv(handle):
    """
    local_spmv callable over a libcsrk handle (csrk_spmv_device on torch's current stream).
    `handle` is the raw csrk_handle_t (int) of this rank's row range.
    """
    from ._lib import lib, check

    def run(x, out):
        assert x.dtype == torch.float64 and out.dtype == torch.float64 and out.is_contiguous()
        stream = torch.cuda.current_stream(out.device).cuda_stream
        check(lib.csrk_spmv_device(handle, x.data_ptr(), out.data_ptr(), stream))
    return run
