"""
Host-side logic of the drop-in boundary, on CPU: the kernel registry (csr/kernels/__init__.py),
the CSR container (csr/csr.py), and the caller contract of CSR.mult_vec / CSR.multiply
(handle lifetime, max_nnz row sharding, zero filter) driven through the TEST-ONLY oracle kernel
(tests/oracle_kernel.py) and compared with golden vectors captured from the reference.
"""
import pickle
import threading

import numpy as np
import pytest

from conftest import Mat

import oracle_kernel


@pytest.fixture()
def okernel():
    import csr_amd.kernels as KS
    KS.kernels['oracle_test'] = oracle_kernel      # out-of-tree registration, SURVEY.md section 3.5
    save = oracle_kernel.max_nnz
    with KS.use_kernel('oracle_test'):
        yield oracle_kernel
    oracle_kernel.max_nnz = save
    assert oracle_kernel.live_handles == 0          # every handle was released (try/finally)


def _csr(m):
    from csr_amd import CSR
    return CSR(m.nrows, m.ncols, m.nnz, m.rowptrs.copy(), m.colinds.copy(),
               None if m.values is None else m.values.copy())


def test_registry_default_and_use_kernel():
    "tests/test_active_kernel.py of the reference"
    import csr_amd.kernels as KS
    k = KS.get_kernel()
    assert k.__name__ == 'csr_amd.kernels.hip'
    for name in ('max_nnz', 'to_handle', 'from_handle', 'release_handle', 'order_columns',
                 'mult_ab', 'mult_abt', 'mult_vec'):           # csr/kernel.py:9-16 + docs/kernels.rst:98
        assert hasattr(k, name), name
    KS.kernels['oracle_test'] = oracle_kernel
    with KS.use_kernel('oracle_test'):
        assert KS.get_kernel() is oracle_kernel
        with KS.use_kernel('hip'):
            assert KS.get_kernel() is k
        assert KS.get_kernel() is oracle_kernel                 # restores the previous kernel
    assert KS.get_kernel() is k
    with pytest.raises(ImportError):
        KS.get_kernel('no_such_kernel')


def test_active_kernel_is_thread_local():
    "csr/kernels/__init__.py:16: threading.local"
    import csr_amd.kernels as KS
    KS.kernels['oracle_test'] = oracle_kernel
    seen = {}

    def other():
        seen['k'] = KS.get_kernel().__name__
    with KS.use_kernel('oracle_test'):
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen['k'] == 'csr_amd.kernels.hip'


def test_csr_layout_rules():
    "csr/csr.py:79-100: colinds int32, rowptrs int32 unless nnz > INT32_MAX, values untouched"
    from csr_amd import CSR
    m = CSR(2, 3, 2, np.array([0, 1, 2], dtype=np.int64), np.array([2, 0], dtype=np.int64),
            np.array([1, 2], dtype=np.float32))
    assert m.rowptrs.dtype == np.int32 and m.colinds.dtype == np.int32 and m.values.dtype == np.float32
    e = CSR.empty(1, 1)
    assert (e.nrows, e.ncols, e.nnz) == (1, 1, 0) and list(e.rowptrs) == [0, 0]
    e2 = CSR.empty(3, 4, [1, 0, 2], values=False)
    assert e2.nnz == 3 and e2.values is None and list(e2.rowptrs) == [0, 1, 1, 3]
    m2 = pickle.loads(pickle.dumps(m))
    assert np.array_equal(m2.colinds, m.colinds) and m2.values.dtype == np.float32
    with pytest.raises(ValueError):
        m.values = np.zeros(1)


def test_from_coo_matches_reference(golden):
    "the reference's from_coo keeps input order inside a row (csr/structure.py:36-58)"
    from csr_amd import CSR
    g = golden('kat')
    a = Mat(g, 'a_')
    m = CSR.from_coo(np.array([0, 0, 1, 3]), np.array([1, 2, 0, 1]), np.arange(4.0))
    assert np.array_equal(m.rowptrs, a.rowptrs) and np.array_equal(m.colinds, a.colinds)
    assert np.array_equal(m.values, a.values)
    assert [tuple(int(v) for v in m.row_extent(i)) for i in range(4)] == [(0, 2), (2, 3), (3, 3), (3, 4)]
    assert np.array_equal(m.row_nnzs(), g['a_row_nnzs'])
    m = CSR.from_coo(np.array([2, 0, 2, 0]), np.array([5, 1, 0, 0]), None, (3, 6))
    assert list(m.rowptrs) == [0, 2, 2, 4] and list(m.colinds) == [1, 0, 5, 0] and m.values is None


def test_shard_rows_golden(golden):
    "tests/test_transform.py:172-197 of the reference"
    from csr_amd import CSR
    g = golden('shard')
    for c in range(int(g['n'])):
        m = _csr(Mat(g, f'c{c}_'))
        shards = m._shard_rows(500)
        assert [s.nrows for s in shards] == list(g[f'c{c}_shard_rows'])
        assert [s.nnz for s in shards] == list(g[f'c{c}_shard_nnz'])
        assert all(s.nnz <= 500 for s in shards)
        assert np.all(np.concatenate([s.row_nnzs() for s in shards]) == m.row_nnzs())
        assert np.all(np.concatenate([s.colinds for s in shards]) == m.colinds)
        back = CSR._assemble_shards(shards)
        assert np.array_equal(back.rowptrs, g[f'c{c}_assembled_rowptrs'])
        assert np.array_equal(back.colinds, m.colinds) and np.array_equal(back.values, m.values)
    big = CSR(2, 1000, 700, np.array([0, 600, 700]), np.arange(700) % 1000, np.ones(700))
    with pytest.raises(ValueError, match='row too large'):
        big._shard_rows(500)


def test_mult_vec_caller_contract(okernel, golden):
    "csr/csr.py:569-590 incl. the sharded branch, against the reference's own sharded output"
    g = golden('spmv')
    hits = 0
    for c in range(int(g['n'])):
        m = _csr(Mat(g, f'c{c}_'))
        x = g[f'c{c}_x']
        assert np.array_equal(m.mult_vec(x), g[f'c{c}_y'])
        if f'c{c}_y_sharded' in g:
            hits += 1
            okernel.max_nnz = 40
            assert np.array_equal(m.mult_vec(x), g[f'c{c}_y_sharded'])
            okernel.max_nnz = np.iinfo('i8').max
        with pytest.raises(AssertionError):
            m.mult_vec(np.ones(m.ncols + 1))
    assert hits > 5


def test_multiply_caller_contract(okernel, golden):
    "csr/csr.py:524-567: product, zero filter, A B^T, and the sharded branch"
    g = golden('spgemm')
    for c in range(int(g['n'])):
        A, B = Mat(g, f'c{c}_a_'), Mat(g, f'c{c}_b_')
        if A.values.dtype != np.float64:
            continue                     # f4 products round differently in the reference (see oracle tests)
        ab = Mat(g, f'c{c}_ab_')
        P = _csr(A).multiply(_csr(B))
        assert np.array_equal(P.rowptrs, ab.rowptrs) and np.array_equal(P.colinds, ab.colinds)
        assert np.array_equal(P.values, ab.values)
        Bt = Mat(g, f'c{c}_bt_')
        abt = Mat(g, f'c{c}_abt_')
        Pt = _csr(A).multiply(_csr(Bt), transpose=True)
        assert np.array_equal(Pt.colinds, abt.colinds) and np.array_equal(Pt.values, abt.values)
        lim = max(B.nnz, Bt.nnz, int(np.max(np.diff(A.rowptrs), initial=0)), 20)
        if A.nnz > 2 * lim:
            okernel.max_nnz = lim
            Ps = _csr(A).multiply(_csr(B))
            okernel.max_nnz = np.iinfo('i8').max
            assert np.array_equal(Ps.rowptrs, ab.rowptrs) and np.array_equal(Ps.values, ab.values)


def test_host_filter_zeros_matches_reference(golden):
    "csr/_struct.py:61-76 (host flavour used for kernels without a device filter)"
    g = golden('spgemm')
    for c in range(int(g['n'])):
        raw, ab = Mat(g, f'c{c}_raw_'), Mat(g, f'c{c}_ab_')
        m = _csr(raw)
        m._filter_zeros()
        assert m.nnz == ab.nnz and np.array_equal(m.rowptrs, ab.rowptrs)
        assert np.array_equal(m.colinds, ab.colinds) and np.array_equal(m.values, ab.values)


def test_subset_rows_views():
    "csr/structure.py:70-81: colinds/values are views, pointers rebased"
    from csr_amd import CSR
    m = CSR.from_coo(np.array([0, 0, 1, 3]), np.array([1, 2, 0, 1]), np.arange(4.0))
    s = m.subset_rows(1, 4)
    assert (s.nrows, s.nnz) == (3, 2) and list(s.rowptrs) == [0, 1, 1, 2]
    assert np.shares_memory(s.colinds, m.colinds) and np.shares_memory(s.values, m.values)


def test_scipy_round_trip_and_row_accessors():
    "csr/csr.py:171-209 (from_scipy / to_scipy), :366-430 (rowinds, row, row_mask, row_cs, row_vs): host conveniences"
    import scipy.sparse as sps
    from csr_amd import CSR
    rng = np.random.default_rng(5)
    sp = sps.random(40, 30, density=0.1, format='csr', random_state=rng, dtype=np.float64)
    m = CSR.from_scipy(sp)
    assert (m.nrows, m.ncols, m.nnz) == (40, 30, sp.nnz)
    assert m.rowptrs.dtype == np.intc and m.colinds.dtype == np.intc
    assert not np.shares_memory(m.values, sp.data) and not np.shares_memory(m.colinds, sp.indices)
    assert np.shares_memory(CSR.from_scipy(sp, copy=False).values, sp.data)
    assert CSR.from_scipy(sp.tocoo()).nnz == sp.nnz and CSR.from_scipy(sp.tocsc()).to_scipy().nnz == sp.nnz
    back = m.to_scipy()
    assert (back != sp).nnz == 0 and np.shares_memory(back.data, m.values)
    dense = sp.toarray()
    assert np.array_equal(m.row(3), dense[3]) and m.row(3).shape == (30,)
    assert np.array_equal(m.row([5, 0, 5]), dense[[5, 0, 5]])
    assert np.array_equal(m.row_mask(7), dense[7] != 0) and m.row_mask([1, 2]).shape == (2, 30)
    lo, hi = m.row_extent(9)
    assert np.array_equal(m.row_cs(9), sp.indices[lo:hi]) and np.array_equal(m.row_vs(9), sp.data[lo:hi])
    assert np.array_equal(m.rowinds(), sp.tocoo().row) and m.rowinds().dtype == np.intc
    s = m.copy(include_values=False)
    assert np.array_equal(s.row(3), (dense[3] != 0).astype(np.float32)) and s.row(3).dtype == np.float32
    assert np.array_equal(s.row_vs(9), np.ones(hi - lo)) and np.array_equal(s.to_scipy().toarray(), (dense != 0) * 1.0)
    e = CSR.empty(3, 4)
    assert np.array_equal(e.row(1), np.zeros(4)) and e.rowinds().size == 0


def test_partition_rows_c_abi_matches_the_python_partition():
    """
    csrk_partition_rows (host only: no device is touched) = the nnz-balanced contiguous row cut of csr_amd.synth.
    balanced_row_ranges / csr_amd/dist.py -- searchsorted(rowptrs, g * nnz / parts), the primitive of the reference's
    _shard_rows (csr/csr.py:609) -- for int32 and int64 row pointers, empty matrices, rows without entries, more parts than rows.
    """
    import ctypes as C
    import torch
    from csr_amd import synth
    from csr_amd._lib import lib, check
    rng = np.random.default_rng(5)
    for trial in range(60):
        n = int(rng.integers(0, 300))
        lens = rng.integers(0, 9, n) * (rng.random(n) < 0.6)
        for dt in (np.int32, np.int64):
            rp = np.concatenate([[0], np.cumsum(lens)]).astype(dt)
            for parts in (1, 2, 3, 8, 64):
                b = (C.c_int32 * (parts + 1))()
                check(lib.csrk_partition_rows(n, rp.ctypes.data_as(C.c_void_p), int(dt == np.int64), parts, b))
                got = list(b)
                assert got == synth.balanced_row_ranges(torch.from_numpy(rp), parts)
                assert got[0] == 0 and got[-1] == n and all(x <= y for x, y in zip(got, got[1:]))
    with pytest.raises(Exception):
        check(lib.csrk_partition_rows(3, None, 0, 2, (C.c_int32 * 3)()))
