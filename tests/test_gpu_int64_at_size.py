"""
Above 2^31 - 1 entries: int64 row pointers AT SIZE (every other ptr64 test holds < 1e6 entries and could not show a
32-bit entry offset overflowing).  The reference's contract includes it -- csr/csr.py:88-93 switches the row pointers to
int64 past INT32_MAX entries; tests/test_mkl.py:94-125 multiplies a 10M x 500 matrix of 2.5e9 entries by a vector and
asks for a finite result; tests/test_initialize.py:56-98 builds such matrices.

One matrix, generated in HBM in eight row shards (csr_amd.synth, torch as plumbing): 4M x 4M, nnz = 2.3e9, power-law rows
(so the SpMV plan's tiers, the packed columns and the staged short rows all work above 2^31), float32 values (the
arrays are 18.4 GB; the plan another ~30 GB).  Checked:
  * csrk_row_nnzs against the generator's degrees (bit-exact);
  * csrk_spmv on the first call (plan-less tile kernel) and on the planned path: finite, equal to each other to 1e-12 of
    sum |a||x|, planned path bitwise reproducible, equal to an independent torch reduction (segment_reduce over the
    entries) to 1e-9, and ~200 sampled rows -- the longest included -- against the oracle's sequential loop at 1e-12;
  * csrk_transpose of the structure (2.3e9 entries through the radix passes): row pointers == the column histogram
    (bit-exact), source rows ascending inside every output row (the stable order of csr/structure.py:207-237), and
    the transpose of the transpose has the original row pointers and columns, bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, NNZ, SHARDS = 4_000_000, 2_300_000_000, 8


class _DevArray:
    "a device pointer as something torch.as_tensor can wrap without copying"

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


@pytest.fixture(scope='module')
def big():
    import torch
    from csr_amd import synth
    free, total = torch.cuda.mem_get_info()
    if total < 150 * 2**30:
        pytest.skip('needs a 288 GB card')
    dev = 'cuda'
    ci = torch.empty(NNZ, dtype=torch.int32, device=dev)
    vs = torch.empty(NNZ, dtype=torch.float32, device=dev)
    rp = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    e0 = 0
    for k in range(SHARDS):
        m = synth.powerlaw_csr(N, N, NNZ, device=dev, rank=k, world=SHARDS, values=False)
        n_loc = int(m['colinds'].numel())
        ci[e0:e0 + n_loc] = m['colinds']
        idx = torch.arange(e0, e0 + n_loc, device=dev, dtype=torch.int64)
        vs[e0:e0 + n_loc] = (synth.hash_uniform(idx, m['seed'], 2) * 2.0 - 1.0).to(torch.float32)
        rp[m['row_begin'] + 1:m['row_end'] + 1] = m['rowptrs'][1:].to(torch.int64) + e0
        e0 += n_loc
        del m, idx
    assert e0 == NNZ and int(rp[-1]) == NNZ and NNZ > 2**31 - 1
    torch.cuda.synchronize()
    yield dict(rowptrs=rp, colinds=ci, values=vs)
    del ci, vs, rp
    torch.cuda.empty_cache()


def _rows_of(m, rows):
    "host copies of the given rows: (rowptrs, colinds, values)"
    rp = m['rowptrs']
    out = []
    for r in rows:
        s, e = int(rp[r]), int(rp[r + 1])
        out.append((m['colinds'][s:e].cpu().numpy(), m['values'][s:e].cpu().numpy()))
    return out


def test_spmv_int64_pointers_at_size(big):
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check, handle_t
    from oracle import oracle as O
    m = big
    dev = 'cuda'
    h = handle_t(0)
    check(lib.csrk_create_device(N, N, NNZ, m['rowptrs'].data_ptr(), 1, m['colinds'].data_ptr(), m['values'].data_ptr(), 1,
                                 C.byref(h)))
    try:
        p64, vt, nnz = C.c_int(), C.c_int(), C.c_int64()
        check(lib.csrk_info(h, None, None, C.byref(nnz), C.byref(p64), C.byref(vt)))
        assert p64.value == 1 and nnz.value == NNZ
        # row_nnzs: int64 output for int64 pointers (csr/csr.py:432-441)
        deg = (m['rowptrs'][1:] - m['rowptrs'][:-1]).cpu().numpy()
        out = np.empty(N, dtype=np.int64)
        check(lib.csrk_row_nnzs(h, out.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(out, deg)
        s, e = C.c_int64(), C.c_int64()
        check(lib.csrk_row_extent(h, N - 1, C.byref(s), C.byref(e)))
        assert e.value == NNZ and s.value == NNZ - int(deg[-1]) and s.value > 2**31

        x = synth.dense_vector(N, device=dev, stream=3)
        y1, y2, y3 = (torch.empty(N, dtype=torch.float64, device=dev) for _ in range(3))
        check(lib.csrk_spmv_device(h, x.data_ptr(), y1.data_ptr(), None))      # first call: plan-less tile kernel
        check(lib.csrk_spmv_device(h, x.data_ptr(), y2.data_ptr(), None))      # second: builds the plan (tiers, streams, staging)
        check(lib.csrk_spmv_device(h, x.data_ptr(), y3.data_ptr(), None))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y1).all()) and bool(torch.isfinite(y2).all())      # tests/test_mkl.py:123-125
        assert torch.equal(y2.view(torch.int64), y3.view(torch.int64))                # bitwise reproducible
        st = (C.c_int64 * 27)()
        check(lib.csrk_spmv_plan_stats(h, st, 27))
        assert st[2] > 0 and st[10] > 0 and st[16] > 0 and st[20] == 1     # rows cut out, tier 0, pack, light stream: all on
        # an independent reduction, entry by entry, shard by shard: y_ref[r] = sum a x, bound[r] = sum |a| |x|
        y_ref = torch.zeros(N, dtype=torch.float64, device=dev)
        bound = torch.zeros(N, dtype=torch.float64, device=dev)
        rp = m['rowptrs']
        cuts = synth.balanced_row_ranges(rp, SHARDS)
        for k in range(SHARDS):
            r0, r1 = cuts[k], cuts[k + 1]
            a, b = int(rp[r0]), int(rp[r1])
            prod = m['values'][a:b].to(torch.float64) * x[m['colinds'][a:b].long()]
            lens = rp[r0 + 1:r1 + 1] - rp[r0:r1]
            y_ref[r0:r1] = torch.segment_reduce(prod, 'sum', lengths=lens, unsafe=True)
            bound[r0:r1] = torch.segment_reduce(prod.abs(), 'sum', lengths=lens, unsafe=True)
            del prod
        for y in (y1, y2):
            assert float(((y - y_ref).abs() / (bound + 1e-300)).max()) <= 1e-9
        assert float(((y1 - y2).abs() / (bound + 1e-300)).max()) <= 1e-12             # the two code paths agree
        # sampled rows against the oracle's sequential loop, the longest and the last rows included
        rows = np.unique(np.concatenate([np.argsort(deg)[-4:], [0, N - 1, N - 2],
                                         np.random.default_rng(3).integers(0, N, 200)]))
        x_h, y_h, b_h = x.cpu().numpy(), y2.cpu().numpy(), bound.cpu().numpy()
        for r, (ci, vs) in zip(rows, _rows_of(m, rows)):
            ref = O.mult_vec(1, N, np.array([0, len(ci)], dtype=np.int32), ci, vs, x_h)[0]
            assert abs(y_h[r] - ref) <= 1e-12 * b_h[r] + 1e-300, (int(r), float(y_h[r]), float(ref))
    finally:
        check(lib.csrk_free(h))
        check(lib.csrk_trim_cache())


def test_transpose_int64_pointers_at_size(big):
    import torch
    from csr_amd._lib import lib, check, handle_t
    m = big
    h, t, tt = handle_t(0), handle_t(0), handle_t(0)
    check(lib.csrk_create_device(N, N, NNZ, m['rowptrs'].data_ptr(), 1, m['colinds'].data_ptr(), None, 0, C.byref(h)))
    try:
        check(lib.csrk_transpose(h, 0, C.byref(t)))
        p64, nnz = C.c_int(), C.c_int64()
        check(lib.csrk_info(t, None, None, C.byref(nnz), C.byref(p64), None))
        assert p64.value == 1 and nnz.value == NNZ                     # the input's pointer width (csr/structure.py:210-216)
        d_rp, d_ci, d_vs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.csrk_device_ptrs(t, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))
        rpt = torch.as_tensor(_DevArray(d_rp.value, N + 1, '<i8'), device='cuda')
        cit = torch.as_tensor(_DevArray(d_ci.value, NNZ, '<i4'), device='cuda')
        # row pointers of the transpose == exclusive scan of the column histogram, bit for bit
        hist = torch.zeros(N, dtype=torch.int64, device='cuda')
        step = 1 << 28
        for a in range(0, NNZ, step):
            hist += torch.bincount(m['colinds'][a:a + step].long(), minlength=N)
        assert int(rpt[0]) == 0 and torch.equal(rpt[1:] - rpt[:-1], hist)
        # source rows ascend inside every output row (stable counting sort): a descent may occur only at a row start
        starts = torch.zeros(NNZ + 1, dtype=torch.bool, device='cuda')
        starts[rpt] = True
        for a in range(0, NNZ - 1, step):
            b = min(a + step, NNZ - 1)
            desc = cit[a + 1:b + 1] <= cit[a:b]
            assert bool((~desc | starts[a + 1:b + 1]).all())
        del starts, hist
        # and back: the transpose of the transpose is the matrix (columns ascend in the generator's rows)
        check(lib.csrk_transpose(t, 0, C.byref(tt)))
        check(lib.csrk_device_ptrs(tt, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))
        rp2 = torch.as_tensor(_DevArray(d_rp.value, N + 1, '<i8'), device='cuda')
        ci2 = torch.as_tensor(_DevArray(d_ci.value, NNZ, '<i4'), device='cuda')
        assert torch.equal(rp2, m['rowptrs']) and torch.equal(ci2, m['colinds'])
    finally:
        for q in (tt, t, h):
            if q.value:
                check(lib.csrk_free(q))
        check(lib.csrk_trim_cache())
