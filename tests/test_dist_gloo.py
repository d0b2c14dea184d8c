"""
The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row partition and
both exchange modes of csr_amd.dist.RowPartitionedSpMV.  The per-rank product is computed by
the oracle here (the HIP kernels need a GPU; they are covered by the -m gpu tests), so this
checks exactly what a multi-GPU run adds: the split points, the padded all-gather / the
all-reduce, and the reassembly.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from csr_amd import synth
        from csr_amd.dist import RowPartitionedSpMV
        from oracle import oracle as O
        nrows, ncols, nnz = 6000, 5000, 90000
        shard = synth.powerlaw_csr(nrows, ncols, nnz, device='cpu', rank=rank, world=world)
        x = synth.dense_vector(ncols, device='cpu')
        rp, ci, vs = (shard[k].numpy() for k in ('rowptrs', 'colinds', 'values'))
        n_loc = shard['row_end'] - shard['row_begin']

        def local_spmv(xt, out):
            out.copy_(torch.from_numpy(O.mult_vec(n_loc, ncols, rp, ci, vs, xt.numpy())))

        op = RowPartitionedSpMV(shard['bounds'], rank, world, local_spmv, 'cpu', mode=mode)
        y1 = op.step(x).clone()
        y2 = op.step(x).clone()          # buffers are reused: a second step must agree
        assert torch.equal(y1, y2)
        np.save(os.path.join(out_dir, f'y_{mode}_{rank}.npy'), y1.numpy())
        np.save(os.path.join(out_dir, f'bounds_{rank}.npy'), np.array(shard['bounds']))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('mode', ['allgather', 'allreduce'])
def test_row_partitioned_spmv_gloo(tmp_path, world, mode):
    from csr_amd import synth
    from oracle import oracle as O
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    full = synth.powerlaw_csr(6000, 5000, 90000, device='cpu')
    x = synth.dense_vector(5000, device='cpu').numpy()
    ref = O.mult_vec(6000, 5000, full['rowptrs'].numpy(), full['colinds'].numpy(), full['values'].numpy(), x)
    for r in range(world):
        y = np.load(tmp_path / f'y_{mode}_{r}.npy')
        # disjoint slices: every rank ends with the single-process result, bit for bit
        assert np.array_equal(y, ref)
        b = np.load(tmp_path / f'bounds_{r}.npy')
        assert b[0] == 0 and b[-1] == 6000 and np.all(np.diff(b) >= 0)
    # nnz balance of the split (searchsorted on rowptrs, csr/csr.py:609)
    rp = full['rowptrs'].numpy()
    per = [int(rp[b[g + 1]] - rp[b[g]]) for g in range(world)]
    assert max(per) - min(per) <= int(np.max(np.diff(rp))) + 1


def _worker_split(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from csr_amd import synth
        from csr_amd.dist import SplitPhaseRowPartitionedSpMV
        from oracle import oracle as O
        nrows, ncols, nnz = 6000, 5000, 90000
        shard = synth.powerlaw_csr(nrows, ncols, nnz, device='cpu', rank=rank, world=world)
        x = synth.dense_vector(ncols, device='cpu')
        rp, ci, vs = (shard[k].numpy() for k in ('rowptrs', 'colinds', 'values'))
        n_loc = shard['row_end'] - shard['row_begin']
        lens = np.diff(rp)
        # the stand-in for the plan's tiers: rows of 40 entries and more (rank 1 of 3 keeps none: an empty list must work)
        cut = np.flatnonzero(lens >= 40) if not (world == 3 and rank == 1) else np.zeros(0, dtype=np.int64)
        is_cut = np.zeros(n_loc, dtype=bool)
        is_cut[cut] = True

        def local_part(xt, out, part):
            full = O.mult_vec(n_loc, ncols, rp, ci, vs, xt.numpy())
            o = out.numpy()
            if part == 1:
                o[:] = np.where(is_cut, 0.0, full)          # the cut rows get 0.0
            else:
                o[is_cut] = full[is_cut]                    # ... and are overwritten by part 2

        op = SplitPhaseRowPartitionedSpMV(shard['bounds'], rank, world, local_part, torch.from_numpy(cut.astype(np.int64)), 'cpu')
        y1 = op.step(x).clone()
        y2 = op.step(x).clone()
        assert torch.equal(y1, y2)
        np.save(os.path.join(out_dir, f'ys_{rank}.npy'), y1.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_split_phase_row_partitioned_spmv_gloo(tmp_path, world):
    "the slice travels while the cut rows are computed, their values follow in a small all-gather: the single-process y"
    from csr_amd import synth
    from oracle import oracle as O
    mp.spawn(_worker_split, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = synth.powerlaw_csr(6000, 5000, 90000, device='cpu')
    x = synth.dense_vector(5000, device='cpu').numpy()
    ref = O.mult_vec(6000, 5000, full['rowptrs'].numpy(), full['colinds'].numpy(), full['values'].numpy(), x)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'ys_{r}.npy'), ref)


def _worker_equal(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from csr_amd import synth
        from csr_amd.dist import RowPartitionedSpMV
        from oracle import oracle as O
        full = synth.powerlaw_csr(6000, 5000, 90000, device='cpu')
        x = synth.dense_vector(5000, device='cpu')
        rp, ci, vs = (full[k].numpy() for k in ('rowptrs', 'colinds', 'values'))
        bounds = [6000 * g // world for g in range(world + 1)]      # equal slices: the only kind gloo's all_gather takes
        a, b = bounds[rank], bounds[rank + 1]

        def local_spmv(xt, out):
            out.copy_(torch.from_numpy(O.mult_vec(b - a, 5000, rp[a:b + 1] - rp[a], ci[rp[a]:rp[b]], vs[rp[a]:rp[b]], xt.numpy())))

        op = RowPartitionedSpMV(bounds, rank, world, local_spmv, 'cpu', mode='allgatherv')
        y1 = op.step(x).clone()
        assert torch.equal(y1, op.step(x))
        np.save(os.path.join(out_dir, f'yv_{rank}.npy'), y1.numpy())
    finally:
        dist.destroy_process_group()


def test_allgatherv_mode_gloo(tmp_path):
    "the slices of y as the output list of one all_gather (equal slices here: gloo; RCCL also takes unequal ones)"
    from csr_amd import synth
    from oracle import oracle as O
    mp.spawn(_worker_equal, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    full = synth.powerlaw_csr(6000, 5000, 90000, device='cpu')
    x = synth.dense_vector(5000, device='cpu').numpy()
    ref = O.mult_vec(6000, 5000, full['rowptrs'].numpy(), full['colinds'].numpy(), full['values'].numpy(), x)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'yv_{r}.npy'), ref)
