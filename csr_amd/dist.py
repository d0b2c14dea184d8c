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

PipelinedRowPartitionedSpMV: point-to-point exchange straight into y (no padding, no concatenation) and/or the
exchange pipelined behind the product (K chunks per rank).  SplitPhaseRowPartitionedSpMV: the exchange hidden
behind the tiers' part of one product (csrk_spmv_device_part).  IpcPushRowPartitionedSpMV: the same with the slices
pushed into the peers' (IPC-mapped) buffers by device copies over xGMI instead of collective kernels.

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


class PipelinedRowPartitionedSpMV:
    """
    The same product with the exchange (a) point-to-point and (b) pipelined behind the product.

    (a) exchange='p2p': every rank computes its rows straight into its slice of y and posts one send of that slice
        to each peer and one receive per peer into the peer's slice of y (dist.batch_isend_irecv: one grouped
        RCCL launch).  xGMI is a full mesh of point-to-point links, so the 7 slices leave over 7 links at once,
        nothing is padded and nothing is copied afterwards (the padded all-gather needs a concatenation kernel:
        80 MB read + 80 MB written per step).  exchange='allgather' keeps the padded all-gather per chunk.
    (b) every rank's row range is cut into K chunks (each a handle of its own); chunk c is exchanged
        asynchronously, on the collective's stream, while chunk c + 1 is being multiplied.  The exchange of the
        80 MB y is link-bound; what can be hidden is the product, behind it.

    Same result bit for bit (the slices are disjoint and every row is computed by the same kernels' rules).
    sub_bounds[g] = K + 1 absolute row indices of rank g's chunks (every rank passes the same table);
    local_spmvs[c](x, out) computes this rank's chunk c into `out`.
    """

    def __init__(self, sub_bounds, rank, world, local_spmvs, device, exchange='p2p', group=None):
        K = len(sub_bounds[0]) - 1
        assert K >= 1 and len(sub_bounds) == world and all(len(b) == K + 1 for b in sub_bounds)
        assert len(local_spmvs) == K and exchange in ('p2p', 'allgather')
        for g in range(world - 1):
            assert sub_bounds[g][-1] == sub_bounds[g + 1][0]
        self.sub = [[int(v) for v in b] for b in sub_bounds]
        assert self.sub[0][0] == 0 and all(b[c] <= b[c + 1] for b in self.sub for c in range(K))
        self.rank, self.world, self.K, self.group, self.exchange = rank, world, K, group, exchange
        self.local_spmvs = local_spmvs
        self.nrows = self.sub[-1][-1]
        self.y = torch.zeros(self.nrows, dtype=torch.float64, device=device)
        self.timing = False
        self._ev = []
        self.lens = [[self.sub[g][c + 1] - self.sub[g][c] for c in range(K)] for g in range(world)]
        if world > 1 and exchange == 'p2p':
            # chunk c: my slice to every peer, every peer's slice c into its place in y; peers in ring order from
            # my own position so that the sends of one chunk do not all start at rank 0's link
            self.ops = []
            for c in range(K):
                ops = []
                mine = self.y[self.sub[rank][c]:self.sub[rank][c + 1]]
                for d in range(1, world):
                    to, frm = (rank + d) % world, (rank - d) % world
                    if self.lens[rank][c]:
                        ops.append(dist.P2POp(dist.isend, mine, to, group))
                    if self.lens[frm][c]:
                        ops.append(dist.P2POp(dist.irecv, self.y[self.sub[frm][c]:self.sub[frm][c + 1]], frm, group))
                self.ops.append(ops)
        elif world > 1:
            self.maxlen = [max(max(self.lens[g][c] for g in range(world)), 1) for c in range(K)]
            self.loc = [torch.zeros(self.maxlen[c], dtype=torch.float64, device=device) for c in range(K)]
            self.gath = [torch.zeros(world * self.maxlen[c], dtype=torch.float64, device=device) for c in range(K)]
            # y = the slices in (rank, chunk) order
            self.pieces = [self.gath[c][g * self.maxlen[c]:g * self.maxlen[c] + self.lens[g][c]]
                           for g in range(world) for c in range(K) if self.lens[g][c]]

    def _local(self, c, x, out):
        if self.timing and out.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.local_spmvs[c](x, out)
            e1.record()
            self._ev.append((e0, e1))
        else:
            self.local_spmvs[c](x, out)

    def compute_ms(self):
        "mean device time per step of the local products (all chunks) over the steps timed so far"
        if not self._ev:
            return 0.0
        ms = sum(a.elapsed_time(b) for a, b in self._ev) * self.K / len(self._ev)
        self._ev = []
        return ms

    def step(self, x):
        "y = A x, complete on every rank; returns the (reused) y tensor"
        r, works = self.rank, []
        for c in range(self.K):
            if self.world == 1 or self.exchange == 'p2p':
                self._local(c, x, self.y[self.sub[r][c]:self.sub[r][c + 1]])
                if self.world > 1 and self.ops[c]:
                    _sync_if_host_backend(self.y, self.group)
                    works += dist.batch_isend_irecv(self.ops[c])
            else:
                self._local(c, x, self.loc[c][:self.lens[r][c]])
                works.append(dist.all_gather_into_tensor(self.gath[c], self.loc[c], group=self.group, async_op=True))
        for w in works:
            w.wait()
        if self.world > 1 and self.exchange == 'allgather' and self.pieces:
            torch.cat(self.pieces, out=self.y)
        return self.y


class SplitPhaseRowPartitionedSpMV:
    """
    The exchange hidden behind the part of the product that touches few rows.  A rank's SpMV has two parts
    (csrk_spmv_device_part): part 1 computes every row of the row-major path and writes 0.0 into the rows the plan
    cut out for its tiers -- a few thousand long rows holding most of the entries --, part 2 computes those rows.
    So: part 1, then the slice is sent to every peer (point to point, straight into y, as in
    PipelinedRowPartitionedSpMV) WHILE part 2 runs, and afterwards the cut rows' values -- a few KB per rank -- follow
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


class IpcPushRowPartitionedSpMV:
    """
    The bulk of the exchange as point-to-point DEVICE COPIES over xGMI instead of collective kernels.  Every rank
    maps its peers' y buffers into its own address space once (torch's IPC tensor sharing: hipIpcGetMemHandle /
    hipIpcOpenMemHandle underneath) and then, per step, PUSHES its slice into each peer's buffer with one asynchronous
    copy per peer, each on a stream of its own -- the copy engines drive the seven links of the full mesh, no CU is
    taken from the product, which matters here because the product's persistent workgroups hold nearly all of a CU's
    LDS and a collective's kernel can find itself waiting for them.  The product is split as in
    SplitPhaseRowPartitionedSpMV: the pushes start after part 1 and run beside part 2; the cut rows' values follow
    in one small all-gather, which is also what tells a rank that every peer's push into its buffer has completed
    (a rank contributes to it only after its own pushes, in stream order, and nobody's all-gather completes without
    everybody's contribution).  Two y buffers alternate, so a peer one step ahead writes into the buffer its
    neighbours are not reading; the per-step all-gather keeps ranks within one step of each other.
    """

    def __init__(self, bounds, rank, world, local_part, cut_rows, device, group=None):
        from torch.multiprocessing.reductions import reduce_tensor
        assert len(bounds) == world + 1 and world > 1
        self.bounds = [int(b) for b in bounds]
        self.rank, self.world, self.group = rank, world, group
        self.local_part = local_part
        self.nrows = self.bounds[-1]
        self.r0, self.r1 = self.bounds[rank], self.bounds[rank + 1]
        self.ys = [torch.zeros(self.nrows, dtype=torch.float64, device=device) for _ in range(2)]
        self.k = 0
        self.timing = False
        self._ev = []
        # the peers' buffers: handles travel as pickled (rebuild function, arguments) pairs
        metas = [reduce_tensor(y) for y in self.ys]
        gathered = [None] * world
        dist.all_gather_object(gathered, metas, group=group)
        self.peer = [[None] * world for _ in range(2)]
        err = None
        try:
            for g in range(world):
                if g != rank:
                    for b in range(2):
                        fn, args = gathered[g][b]
                        t = fn(*args)
                        assert t.numel() == self.nrows and t.dtype == torch.float64
                        self.peer[b][g] = t
        except Exception as e:                # every rank must learn of it before anyone enters a collective alone
            err = e
        bad = torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if bad.item() > 0:
            raise RuntimeError(f'peer buffers could not be mapped: {err!r}')
        self.peers = [(rank + d) % world for d in range(1, world)]
        self.streams = [torch.cuda.Stream(device=device) for _ in self.peers]
        # the cut rows of every rank (global indices), as in SplitPhaseRowPartitionedSpMV
        cut = cut_rows.to(device=device, dtype=torch.int64) + self.r0
        n_mine = int(cut.numel())
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[rank] = n_mine
        dist.all_reduce(counts, group=group)
        counts = [int(c) for c in counts.tolist()]
        self.n_mine, self.maxn = n_mine, max(max(counts), 1)
        self.my_rows = cut
        self.hv_loc = torch.zeros(self.maxn, dtype=torch.float64, device=device)
        self.hv_all = torch.zeros(world * self.maxn, dtype=torch.float64, device=device)
        pad = torch.zeros(self.maxn, dtype=torch.int64, device=device)
        pad[:n_mine] = cut
        rows_all = torch.zeros(world * self.maxn, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(rows_all, pad, group=group)
        src = [torch.arange(g * self.maxn, g * self.maxn + counts[g], device=device) for g in range(world) if g != rank]
        self.src_pos = torch.cat(src)
        self.dst_rows = rows_all.index_select(0, self.src_pos)
        self.tmp = torch.zeros(int(self.src_pos.numel()), dtype=torch.float64, device=device)

    def compute_ms(self):
        if not self._ev:
            return 0.0
        ms = sum(a.elapsed_time(b) for a, b in self._ev) * 2 / len(self._ev)
        self._ev = []
        return ms

    def _part(self, x, out, part):
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.local_part(x, out, part)
            e1.record()
            self._ev.append((e0, e1))
        else:
            self.local_part(x, out, part)

    def step(self, x):
        "y = A x, complete on every rank; returns this step's y buffer (the two buffers alternate)"
        b = self.k & 1
        self.k += 1
        y = self.ys[b]
        mine = y[self.r0:self.r1]
        cur = torch.cuda.current_stream(y.device)
        self._part(x, mine, 1)
        if self.r1 > self.r0:
            ready = torch.cuda.Event()
            ready.record(cur)
            for st, g in zip(self.streams, self.peers):
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    self.peer[b][g][self.r0:self.r1].copy_(mine, non_blocking=True)
        self._part(x, mine, 2)
        if self.n_mine:
            torch.index_select(y, 0, self.my_rows, out=self.hv_loc[:self.n_mine])
        for st in self.streams:
            cur.wait_stream(st)
        dist.all_gather_into_tensor(self.hv_all, self.hv_loc, group=self.group)      # values + "my pushes are done"
        if self.tmp.numel():
            torch.index_select(self.hv_all, 0, self.src_pos, out=self.tmp)
            y.index_copy_(0, self.dst_rows, self.tmp)
        return y


def chunk_cuts(rowptrs, K):
    """
    K + 1 local row indices cutting a rank's row range into K chunks balanced by nnz - the same rule as the rank
    boundaries (searchsorted on the row pointers: the primitive of _shard_rows, csr/csr.py:609).
    rowptrs: this rank's rebased row pointers (tensor or array, rowptrs[0] == 0).
    """
    rp = torch.as_tensor(rowptrs)
    n = rp.numel() - 1
    total = int(rp[-1])
    tg = torch.tensor([total * c // K for c in range(1, K)], dtype=rp.dtype, device=rp.device)
    mid = torch.searchsorted(rp, tg).tolist() if K > 1 else []
    cuts = [0] + [min(max(int(m), 0), n) for m in mid] + [n]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


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
