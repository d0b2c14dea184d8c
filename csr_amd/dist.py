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
  allgatherv the slices of y themselves as the output list of one all_gather (RCCL: a group of broadcasts for
             slices of different lengths): no padding, no concatenation.
  allreduce  the form BASELINE.json's north_star names: each rank writes its slice into a
             zeroed full-length y and the ranks sum.  Same result (the slices are disjoint, so
             every sum has one non-zero term and is exact), about twice the bytes per link.

SplitPhaseRowPartitionedSpMV: point-to-point sends of the slices straight into y, hidden behind the tiers' part of one
product (csrk_spmv_device_part).  (Rounds 2-3 also carried chunk-pipelined, plain point-to-point and IPC-push forms; none
has ever run with two RCCL ranks, so they were removed: `git log -- csr_amd/dist.py`.)

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
        assert len(bounds) == world + 1 and mode in ('allgather', 'allgatherv', 'allreduce')
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

        if world > 1 and mode == 'allgatherv':
            # every rank's slice of y itself, as the output list of ONE all_gather: with slices of different lengths
            # RCCL runs it as a group of broadcasts straight into place (nothing padded, nothing concatenated);
            # gloo only takes equal lengths
            self.views = [self.y[self.bounds[g]:self.bounds[g + 1]] for g in range(world)]

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
        if self.mode == 'allgatherv':
            self._local(x, self.views[self.rank])
            dist.all_gather(self.views, self.views[self.rank], group=self.group)
            return self.y
        # allreduce: zero what the previous step left in the other ranks' slices
        if self.r0 > 0:
            self.y[:self.r0].zero_()
        if self.r1 < self.nrows:
            self.y[self.r1:].zero_()
        self._local(x, self.y[self.r0:self.r1])
        dist.all_reduce(self.y, op=dist.ReduceOp.SUM, group=self.group)
        return self.y


def _sync_if_host_backend(t, group):
    """
    gloo (the CPU tests, and bench.py's two-ranks-on-one-GPU test hook) sends device tensors from the host without
    waiting for the stream that produces them: wait here.  RCCL orders a send after the producing kernels by itself.
    """
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        torch.cuda.synchronize(t.device)


class SplitPhaseRowPartitionedSpMV:
    """
    The exchange hidden behind the part of the product that touches few rows.  A rank's SpMV has two parts
    (csrk_spmv_device_part): part 1 computes every row of the row-major path and writes 0.0 into the rows the plan
    cut out for its tiers -- a few thousand long rows holding most of the entries --, part 2 computes those rows.
    So: part 1, then the slice is sent to every peer (point to point, straight into y: one
    grouped batch_isend_irecv) WHILE part 2 runs, and afterwards the cut rows' values -- a few KB per rank -- follow
    in one small all-gather and are written over the stale entries on every rank.  (The big send may read a cut
    row's entry before or after part 2 stores it; either way the small exchange overwrites it with the final value.)
    No extra handles, x is read once per kernel as in the plain product.

    local_part(x, out, part): this rank's rows, part 1 / 2 (asynchronous on the current stream);
    cut_rows: ascending int64 LOCAL row indices part 2 writes (may be empty), on `device`.
    """

    def __init__(self, bounds, rank, world, local_part, cut_rows, device, group=None):
        assert len(bounds) == world + 1
        self.bounds = [int(b) for b in bounds]
        self.rank, self.world, self.group = rank, world, group
        self.local_part = local_part
        self.nrows = self.bounds[-1]
        self.r0, self.r1 = self.bounds[rank], self.bounds[rank + 1]
        self.y = torch.zeros(self.nrows, dtype=torch.float64, device=device)
        self.timing = False
        self._ev = []
        self.ops = []
        mine = self.y[self.r0:self.r1]
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world
            if self.r1 > self.r0:
                self.ops.append(dist.P2POp(dist.isend, mine, to, group))
            if self.bounds[frm + 1] > self.bounds[frm]:
                self.ops.append(dist.P2POp(dist.irecv, self.y[self.bounds[frm]:self.bounds[frm + 1]], frm, group))
        # the cut rows of every rank, as global row indices: exchanged once
        cut = cut_rows.to(device=device, dtype=torch.int64) + self.r0
        n_mine = int(cut.numel())
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[rank] = n_mine
        if world > 1:
            dist.all_reduce(counts, group=group)
        counts = [int(c) for c in counts.tolist()]
        self.n_mine, self.maxn = n_mine, max(max(counts), 1)
        self.total_cut = sum(counts)
        self.my_rows = cut
        self.hv_loc = torch.zeros(self.maxn, dtype=torch.float64, device=device)
        self.hv_all = torch.zeros(world * self.maxn, dtype=torch.float64, device=device)
        if world > 1 and self.total_cut:
            pad = torch.zeros(self.maxn, dtype=torch.int64, device=device)
            pad[:n_mine] = cut
            rows_all = torch.zeros(world * self.maxn, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(rows_all, pad, group=group)
            src = [torch.arange(g * self.maxn, g * self.maxn + counts[g], device=device) for g in range(world) if g != rank]
            self.src_pos = torch.cat(src) if src else torch.zeros(0, dtype=torch.int64, device=device)
            self.dst_rows = rows_all.index_select(0, self.src_pos)
            self.tmp = torch.zeros(int(self.src_pos.numel()), dtype=torch.float64, device=device)

    def compute_ms(self):
        "mean device time per step of the two local parts over the steps timed so far"
        if not self._ev:
            return 0.0
        ms = sum(a.elapsed_time(b) for a, b in self._ev) * 2 / len(self._ev)
        self._ev = []
        return ms

    def _part(self, x, out, part):
        if self.timing and out.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.local_part(x, out, part)
            e1.record()
            self._ev.append((e0, e1))
        else:
            self.local_part(x, out, part)

    def step(self, x):
        "y = A x, complete on every rank; returns the (reused) y tensor"
        mine = self.y[self.r0:self.r1]
        self._part(x, mine, 1)
        if self.world > 1 and self.ops:
            _sync_if_host_backend(self.y, self.group)
        works = dist.batch_isend_irecv(self.ops) if (self.world > 1 and self.ops) else []
        self._part(x, mine, 2)
        if self.world > 1 and self.total_cut:
            if self.n_mine:
                torch.index_select(self.y, 0, self.my_rows, out=self.hv_loc[:self.n_mine])
        for w in works:
            w.wait()
        if self.world > 1 and self.total_cut:
            dist.all_gather_into_tensor(self.hv_all, self.hv_loc, group=self.group)
            if self.tmp.numel():
                torch.index_select(self.hv_all, 0, self.src_pos, out=self.tmp)
                self.y.index_copy_(0, self.dst_rows, self.tmp)
        return self.y


def hip_local_spmv_parts(handle, device):
    """
    (local_part, cut_rows) over a libcsrk handle for SplitPhaseRowPartitionedSpMV: csrk_spmv_device_part on torch's
    current stream, and the plan's cut rows (csrk_spmv_cut_rows: builds the plan) as an int64 tensor on `device`.
    """
    import ctypes as C
    from ._lib import lib, check

    def run(x, out, part):
        assert x.dtype == torch.float64 and out.dtype == torch.float64 and out.is_contiguous()
        stream = torch.cuda.current_stream(out.device).cuda_stream
        check(lib.csrk_spmv_device_part(handle, x.data_ptr(), out.data_ptr(), stream, int(part)))

    n = C.c_int64(0)
    check(lib.csrk_spmv_cut_rows(handle, None, 0, C.byref(n)))
    rows = torch.zeros(max(n.value, 1), dtype=torch.int32, device=device)
    if n.value:
        check(lib.csrk_spmv_cut_rows(handle, rows.data_ptr(), n.value, C.byref(n)))
    return run, rows[:n.value].to(torch.int64)


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
