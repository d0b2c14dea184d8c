"""
Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle would need
seconds to minutes per case at these sizes, so it is used on sampled rows only):

  configs[1]  SpMV, 10M x 10M power-law, nnz = 2e8, fp64:
              linearity  A(a x + b z) = a A x + b A z,
              checksum   sum_i y_i = sum_k a_k x_col(k)   (formed independently, entry by entry),
              sampled rows against the oracle's sequential loop,
              bitwise reproducibility across launches.
  configs[2]  dense-panel SpMM, A 2M x 2M nnz 5e7, B 2M x 64: every panel column equals the SpMV
              with that column of B (checked for 3 columns).
  configs[4]  transpose, MovieLens-25M shape: transpose(transpose(A)) == A bit for bit, row pointers of
              the transpose == column histogram, entry multiset preserved.
The matrices are generated in HBM by csr_amd.synth (torch is plumbing); all products go through the
libcsrk C ABI.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(m, nrows, ncols):
    from csr_amd._lib import lib, check, handle_t
    h = handle_t(0)
    nnz = int(m['colinds'].numel())
    check(lib.csrk_create_device(nrows, ncols, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(),
                                 m['values'].data_ptr(), 2, C.byref(h)))
    return h


def _spmv(h, x, y):
    from csr_amd._lib import lib, check
    check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))


def test_spmv_config2_full_size():
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check
    from oracle import oracle as O
    dev = 'cuda'
    n, nnz = 10_000_000, 200_000_000
    m = synth.powerlaw_csr(n, n, nnz, device=dev)
    assert int(m['colinds'].numel()) == nnz and int(m['rowptrs'][-1]) == nnz
    h = _handle(m, n, n)
    habs = None
    try:
        x = synth.dense_vector(n, device=dev, stream=3)
        z = synth.dense_vector(n, device=dev, stream=4)
        y1, y2, y3, yb = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(4))
        _spmv(h, x, y1)                  # 1st launch: single merge path
        _spmv(h, x, y2)                  # 2nd launch: tiered kernels (lazy split)
        _spmv(h, x, y3)
        torch.cuda.synchronize()
        assert torch.equal(y2, y3)       # bitwise reproducible
        import ctypes as C
        st = (C.c_int64 * 20)()
        check(lib.csrk_spmv_plan_stats(h, st, 20))
        assert st[2] > 0 and st[10] > 0 and st[13] > 0     # both panel tiers are in use at this size ...
        assert st[16] > 0                                  # ... and so is the hot-column pack
        # error scale: sum_j |a_ij| |x_j| through the same kernels on |A|, |x|
        absv = m['values'].abs()
        mabs = dict(m, values=absv)
        habs = _handle(mabs, n, n)
        _spmv(habs, x.abs().contiguous(), yb)
        _spmv(habs, x.abs().contiguous(), yb)
        torch.cuda.synchronize()
        assert float(((y1 - y2).abs() / (yb + 1e-300)).max()) <= 1e-12     # both code paths agree
        # linearity
        a, b = 0.75, -1.5
        comb = (a * x + b * z).contiguous()
        yz, yc = torch.empty_like(y1), torch.empty_like(y1)
        _spmv(h, z, yz)
        _spmv(h, comb, yc)
        _spmv(habs, z.abs().contiguous(), y3)          # bound for the z part
        torch.cuda.synchronize()
        bound = abs(a) * yb + abs(b) * y3
        assert float(((yc - (a * y2 + b * yz)).abs() / (bound + 1e-300)).max()) <= 1e-12
        # checksum of checksums
        # (formed independently with torch ops, entry by entry, no row structure involved; an index_add_
        # by column would serialise for minutes on the popular columns)
        xg = x[m['colinds'].long()]
        want, got = float((m['values'] * xg).sum()), float(y2.sum())
        scale = float((absv * xg.abs()).sum())
        del xg
        assert abs(want - got) <= 1e-9 * scale
        # sampled rows against the oracle (sequential loop), incl. the longest rows
        rp = m['rowptrs'].cpu().numpy()
        lens = np.diff(rp)
        rows = np.unique(np.concatenate([np.argsort(lens)[-5:], np.random.default_rng(1).integers(0, n, 2000)]))
        x_h = x.cpu().numpy()
        y_h, yb_h = y2.cpu().numpy(), yb.cpu().numpy()
        ci, vs = m['colinds'].cpu().numpy(), m['values'].cpu().numpy()      # one 2.4 GB copy, then host slices
        for r in rows:
            s, e = int(rp[r]), int(rp[r + 1])
            srp = np.array([0, e - s], dtype=np.int32)
            ref = O.mult_vec(1, n, srp, ci[s:e], vs[s:e], x_h)[0]
            assert abs(y_h[r] - ref) <= 1e-6 * yb_h[r] + 1e-300        # north_star tolerance
            assert abs(y_h[r] - ref) <= 1e-12 * yb_h[r] + 1e-300
    finally:
        check(lib.csrk_free(h))
        if habs is not None:
            check(lib.csrk_free(habs))
        check(lib.csrk_trim_cache())


def test_spmm_config3_full_size():
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check
    dev = 'cuda'
    n, nnz, k = 2_000_000, 50_000_000, 64
    m = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=250_000)
    h = _handle(m, n, n)
    habs = _handle(dict(m, values=m['values'].abs()), n, n)
    try:
        B = synth.dense_vector(n * k, device=dev, stream=7).view(n, k)
        Cm = torch.empty(n, k, dtype=torch.float64, device=dev)
        check(lib.csrk_spmm_dense_device(h, B.data_ptr(), k, k, Cm.data_ptr(), k, None))
        y, yb = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
        for c in (0, 31, 63):
            xc = B[:, c].contiguous()
            _spmv(h, xc, y)
            _spmv(habs, xc.abs().contiguous(), yb)
            torch.cuda.synchronize()
            assert float(((Cm[:, c] - y).abs() / (yb + 1e-300)).max()) <= 1e-12
        # ... and against the pinned oracle itself (the reference's mult_ab(A, CSR(B)) recurrence, multiply.py:110-122) on
        # the rows holding the first 2e6 entries plus the 3 longest rows: every panel entry to 1e-12 of sum |a||b|
        from oracle import oracle as O
        rp_h = m['rowptrs'].cpu().numpy()
        r_s = int(np.searchsorted(rp_h, 2_000_000))
        e_s = int(rp_h[r_s])
        B_h = B.cpu().numpy()
        ci_h, vs_h = m['colinds'][:e_s].cpu().numpy(), m['values'][:e_s].cpu().numpy()
        C_o = O.spmm_dense(r_s, rp_h[:r_s + 1], ci_h, vs_h, B_h)
        C_b = O.spmm_dense(r_s, rp_h[:r_s + 1], ci_h, np.abs(vs_h), np.abs(B_h))
        assert np.all(np.abs(Cm[:r_s].cpu().numpy() - C_o) <= 1e-12 * C_b + 1e-300)
        for r in np.argsort(np.diff(rp_h))[-3:]:
            a, b = int(rp_h[r]), int(rp_h[r + 1])
            ci_r, vs_r = m['colinds'][a:b].cpu().numpy(), m['values'][a:b].cpu().numpy()
            srp = np.array([0, b - a], dtype=np.int32)
            c_o = O.spmm_dense(1, srp, ci_r, vs_r, B_h)[0]
            c_b = O.spmm_dense(1, srp, ci_r, np.abs(vs_r), np.abs(B_h))[0]
            assert np.all(np.abs(Cm[int(r)].cpu().numpy() - c_o) <= 1e-12 * c_b + 1e-300)
    finally:
        check(lib.csrk_free(h))
        check(lib.csrk_free(habs))
        check(lib.csrk_trim_cache())


def test_transpose_config5_full_size():
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check, handle_t
    dev = 'cuda'
    nr, nc, nnz = 162_541, 59_047, 25_000_095
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
    m['values'] = (torch.floor((m['values'] + 1.0) * 5.0).clamp_(0, 9) + 1.0) * 0.5     # ratings 0.5 .. 5.0
    h = _handle(m, nr, nc)
    t, tt = handle_t(0), handle_t(0)
    try:
        check(lib.csrk_transpose(h, 1, C.byref(t)))
        check(lib.csrk_transpose(t, 1, C.byref(tt)))
        rpt = np.empty(nc + 1, np.int32)
        cit = np.empty(nnz, np.int32)
        vst = np.empty(nnz)
        check(lib.csrk_export(t, rpt.ctypes.data_as(C.c_void_p), cit.ctypes.data_as(C.c_void_p), vst.ctypes.data_as(C.c_void_p)))
        ci_h = m['colinds'].cpu().numpy()
        assert np.array_equal(np.diff(rpt), np.bincount(ci_h, minlength=nc))        # structure.py:180-188
        # ascending source rows inside every output row = the stable order of the reference
        d = np.diff(cit.astype(np.int64))
        ends = rpt[1:-1] - 1
        ends = ends[(ends >= 0) & (ends < nnz - 1)]
        inner = np.ones(nnz - 1, dtype=bool)
        inner[ends] = False
        assert np.all(d[inner] > 0)
        assert np.isclose(vst.sum(), float(m['values'].sum()), rtol=0, atol=0)      # values are copied bits (halves sum exactly)
        rp2 = np.empty(nr + 1, np.int32)
        ci2 = np.empty(nnz, np.int32)
        vs2 = np.empty(nnz)
        check(lib.csrk_export(tt, rp2.ctypes.data_as(C.c_void_p), ci2.ctypes.data_as(C.c_void_p), vs2.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(rp2, m['rowptrs'].cpu().numpy())
        assert np.array_equal(ci2, ci_h)
        assert np.array_equal(vs2, m['values'].cpu().numpy())
    finally:
        for q in (h, t, tt):
            check(lib.csrk_free(q))
        check(lib.csrk_trim_cache())


def test_mult_abt_config5_large_block():
    """
    BASELINE.json configs[4], the mult_abt half, at a block size a caller would use (int32 product pointers,
    multiply.py:28, force row blocks): A[0:6000] . B[0:20000]^T of the MovieLens-25M-shaped matrix -- 1.2e9 products, 1.2e8
    outputs -- through csrk_spgemm_abt (transpose + column strips, csrc/spgemm.hip), against the oracle's transpose + SMMP:
    row pointers, columns (ascending here, reverse discovery there) and VALUES bit for bit -- the strips add every
    entry's products in the reference's order (multiply.py:117-121).  And the encode/decode property at this size:
    (A B^T)^T == B A^T entry for entry up to the order of addition, checked through row sums.
    """
    from oracle import oracle as O
    from csr_amd import CSR, synth
    from csr_amd.kernels import hip as K
    m = synth.movielens_like(device='cpu')
    M = CSR(m['nrows'], m['ncols'], int(m['colinds'].numel()), m['rowptrs'].numpy(), m['colinds'].numpy(), m['values'].numpy(),
            _cast=False)
    A, B = M.subset_rows(0, 6000), M.subset_rows(0, 20000)
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        ch = K.mult_abt(ah, bh)
        Cm = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    bt = O.transpose(B.nrows, B.ncols, B.rowptrs, B.colinds, B.values)
    nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), bt)
    assert (Cm.nrows, Cm.ncols) == (6000, 20000) == (nr, nc)
    assert np.array_equal(Cm.rowptrs, crp)
    # the reference's raw arrays, bit for bit: columns in its order (reverse of first discovery), the default
    assert K.spgemm_order() == 'reference'
    assert np.array_equal(Cm.colinds, cci)
    assert np.array_equal(Cm.values.view(np.int64), cvs.view(np.int64))
    # column sums of C against B (A^T 1): sum_i C[i, k] = sum_j B[k, j] * (sum_i A[i, j])
    colsum_a = np.bincount(A.colinds, weights=A.values, minlength=A.ncols)
    want = np.add.reduceat(B.values * colsum_a[B.colinds], B.rowptrs[:-1].astype(np.int64))
    want[np.diff(B.rowptrs) == 0] = 0.0
    got = np.bincount(Cm.colinds, weights=Cm.values, minlength=Cm.ncols)
    scale = np.add.reduceat(np.abs(B.values) * np.bincount(A.colinds, weights=np.abs(A.values), minlength=A.ncols)[B.colinds],
                            B.rowptrs[:-1].astype(np.int64))
    assert np.all(np.abs(got - want) <= 1e-10 * scale + 1e-300)


@pytest.mark.parametrize('collective', ['auto', 'allgather', 'p2p-split'])
def test_bench_two_ranks_plumbing(collective):
    """
    bench.py's N > 1 path end to end on ONE GPU, started the way the driver starts N = 1 -- plain
    `python bench.py --gpus 2 ...`, no launcher: bench.py starts its own one-rank-per-GPU job as a child process.  Two
    ranks share cuda:0 over gloo (RCCL refuses two ranks on a device; BENCH_TEST_SHARE_GPU is the bench's own test
    hook).  Covers the self-launch, shard generation, the per-rank plans, the row-partitioned step with its exchange,
    the (bounded) calibration, the max-over-ranks timing and rank 0's JSON line.
    """
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BENCH_TEST_SHARE_GPU'] = '1'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--scale', '0.05']
    cmd += ['--collective', collective]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['value'] > 0 and d['scaling'] == 'strong'
    assert d['config']['parallelism'] == 'row-partition x2' and 'multi_gpu' in d
    assert d['multi_gpu']['y_complete_and_identical_on_every_rank'] is True
    assert d['multi_gpu']['kernel_only_gflops'] > 0
    if collective == 'auto':
        cands = d['multi_gpu']['candidates_ms_per_step']
        assert d['multi_gpu']['exchange'] in cands and 'allgather' in cands and 'p2p-split' in cands
    else:
        assert d['multi_gpu']['exchange'] == collective


@pytest.mark.parametrize('collective', ['auto', 'allreduce', 'p2p-split'])
def test_bench_one_rank_on_rccl(collective):
    """
    The N > 1 code path of bench.py under the REAL backend with ONE rank (`--gpus 1 --force-dist`): RCCL loads and builds
    its communicator (init_process_group('nccl', device_id=...)), all_gather_into_tensor / all_gather / all_reduce /
    batch_isend_irecv and the split-phase form's small all-gather run on device tensors, the calibration picks an
    exchange, the completeness check passes, and the exchanged y equals the plain product bit for bit.  The parallel
    form of csr/csr.py:584-590 (shard products concatenated) as far as one GPU can rehearse it; RCCL refuses two ranks
    on one device, so the 2-rank plumbing test above runs on gloo.
    """
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BENCH_TEST_SHARE_GPU',
                                                            'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '3', '--warmup', '1',
           '--scale', '0.05', '--no-cpu-baseline', '--collective', collective]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    mg = d['multi_gpu']
    assert d['n_gpus'] == 1 and mg['backend'] == 'nccl'
    assert mg['y_complete_and_identical_on_every_rank'] is True and mg['y_equals_plain_product_bitwise'] is True
    assert mg['kernel_only_gflops'] > 0 and mg['end_to_end_gflops'] > 0
    if collective == 'auto':
        cands = mg['candidates_ms_per_step']
        assert mg['exchange'] in cands
        for name in ('allgather', 'allgatherv', 'p2p-split'):      # each ran on RCCL (a number, not an error string)
            assert isinstance(cands[name], float), cands
    else:
        assert mg['exchange'] == collective


def test_bench_refuses_more_ranks_than_gpus():
    "`python bench.py --gpus 8` on a box with fewer GPUs: a clear message and a non-zero exit, before anything is launched"
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BENCH_TEST_SHARE_GPU')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'],
                         cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'GPU(s) visible' in out.stderr


def test_cold_staging_and_stream_forms_are_bit_neutral(monkeypatch):
    """
    The staging pass only changes WHERE the light stream reads an x value from (xg instead of x / the pack), and
    the 16-bit accumulator stream only how a heavy row's index is stored: with staging off the result must be
    the same bit for bit, on a matrix large enough for every part of the plan (tiers, pack, staging) to exist.
    Also the two-part product (csrk_spmv_device_part 1 then 2) at this size.
    """
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check
    dev = 'cuda'
    n, nnz = 3_000_000, 60_000_000
    m = synth.powerlaw_csr(n, n, nnz, device=dev)
    x = synth.dense_vector(n, device=dev, stream=3)
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CSRK_LS_STAGE', mode)
        h = _handle(m, n, n)
        try:
            y = torch.empty(n, dtype=torch.float64, device=dev)
            for _ in range(3):
                _spmv(h, x, y)
            torch.cuda.synchronize()
            st = (C.c_int64 * 25)()
            check(lib.csrk_spmv_plan_stats(h, st, 25))
            assert st[2] > 0 and st[16] > 0 and st[20] == 1      # tiers, pack, light stream
            assert (st[24] > 0) == (mode == '1')                 # staged entries only with staging on
            out[mode] = y.clone()
            if mode == '1':
                yp = torch.full((n,), 3.0, dtype=torch.float64, device=dev)
                check(lib.csrk_spmv_device_part(h, x.data_ptr(), yp.data_ptr(), None, 1))
                check(lib.csrk_spmv_device_part(h, x.data_ptr(), yp.data_ptr(), None, 2))
                torch.cuda.synchronize()
                assert torch.equal(yp, y)
        finally:
            check(lib.csrk_free(h))
    assert torch.equal(out['1'], out['0'])
