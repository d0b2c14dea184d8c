"""
Randomised sweep in the reference's own test distribution (csr/test_utils.py:30-101: shapes 1..80 /
1..100, density <= 0.5, f4 / f8 / structure-only, zeros dropped), 150 matrices per operation, every
result compared with the oracle through the C ABI -- the GPU analogue of the reference's hypothesis tests
(tests/test_mult_vec.py, test_multiply.py, test_transpose.py, test_transform.py).
"""
import numpy as np
import pytest

from conftest import sort_within_rows

pytestmark = pytest.mark.gpu


def _draw(rng, nrows=None, ncols=None, values=None, max_dim=80):
    from csr_amd import CSR
    nrows = nrows or int(rng.integers(1, max_dim + 1))
    ncols = ncols or int(rng.integers(1, max_dim + 1))
    nnz = int(rng.integers(0, int(np.ceil(nrows * ncols * 0.5)) + 1))
    coords = rng.choice(nrows * ncols, size=nnz, replace=False)
    rows, cols = (coords % nrows).astype(np.int32), (coords // nrows).astype(np.int32)
    dtype = np.dtype(rng.choice(['f4', 'f8']))
    if values is None:
        values = bool(rng.integers(0, 2))
    vals = None
    if values:
        vals = rng.uniform(-1e3, 1e3, size=nnz).astype(dtype)
        nz = vals != 0
        rows, cols, vals = rows[nz], cols[nz], vals[nz]
    return CSR.from_coo(rows, cols, vals, (nrows, ncols))


def test_sweep_mult_vec_and_transpose():
    from oracle import oracle as O
    rng = np.random.default_rng(2026)
    for _ in range(150):
        m = _draw(rng)
        x = rng.uniform(-1e3, 1e3, size=m.ncols)
        y = m.mult_vec(x)
        ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
        vs = None if m.values is None else np.abs(m.values.astype(np.float64))
        bound = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, vs, np.abs(x))
        assert y.shape == (m.nrows,) and np.all(np.abs(y - ref) <= 1e-12 * bound + 1e-300)
        t = m.transpose()
        nr, nc, brp, bci, bvs = O.transpose(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values)
        assert (t.nrows, t.ncols, t.nnz) == (nr, nc, m.nnz)
        assert np.array_equal(t.rowptrs, brp) and np.array_equal(t.colinds, bci)
        assert (t.values is None) == (bvs is None) and (bvs is None or np.array_equal(t.values, bvs))


def test_sweep_multiply():
    from oracle import oracle as O
    rng = np.random.default_rng(777)
    for _ in range(150):
        r, mid, k = (int(rng.integers(1, 101)) for _ in range(3))
        A = _draw(rng, r, mid, values=True)
        B = _draw(rng, mid, k, values=True)
        P = A.multiply(B)
        nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values),
                                          (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values))
        _, _, _, _, babs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, np.abs(A.values)),
                                     (B.nrows, B.ncols, B.rowptrs, B.colinds, np.abs(B.values)))
        # dense comparison, as tests/test_multiply.py does
        dref, dbound, dgot = np.zeros((nr, nc)), np.zeros((nr, nc)), np.zeros((nr, nc))
        rows = np.repeat(np.arange(nr), np.diff(crp))
        dref[rows, cci] = cvs
        dbound[rows, cci] = babs
        prow = np.repeat(np.arange(P.nrows), np.diff(P.rowptrs))
        dgot[prow, P.colinds] = P.values
        assert (P.nrows, P.ncols) == (nr, nc)
        assert P.nnz == 0 or np.all(P.values != 0)                       # tests/test_multiply.py:34-36
        assert np.all(np.abs(dgot - dref) <= 1e-12 * dbound + 1e-300)


def test_sweep_unit_and_center():
    from oracle import oracle as O
    rng = np.random.default_rng(31337)
    for _ in range(150):
        m = _draw(rng, values=True)
        f4 = m.values.dtype == np.float32
        rel = 1e-5 if f4 else 1e-9
        u, ur = m.copy(), m.values.copy()
        with np.errstate(all='ignore'):
            norms = u.normalize_rows('unit')
            rn = O.unit_rows(m.nrows, m.rowptrs, ur)
        assert norms.dtype == m.values.dtype
        assert norms == pytest.approx(rn, rel=rel, abs=0, nan_ok=True)
        assert np.array_equal(np.isnan(u.values), np.isnan(ur))
        assert u.values == pytest.approx(ur, rel=rel, abs=1e-300, nan_ok=True)
        c, cr = m.copy(), m.values.copy()
        means = c.normalize_rows('center')
        rm = O.center_rows(m.nrows, m.rowptrs, cr)
        tol = (float(np.max(np.abs(m.values))) if m.nnz else 1.0) * (1e-6 if f4 else 1e-12)
        assert means == pytest.approx(rm, rel=rel, abs=tol)
        assert c.values == pytest.approx(cr, rel=rel, abs=tol)


def test_sweep_multiply_reference_order():
    "mult_ab with the reference's own column order switched on: raw arrays bit for bit the oracle's (csr/kernels/numba/multiply.py:79-97)"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(4711)
    K.set_spgemm_order('reference')
    try:
        for _ in range(100):
            r, mid, k = (int(rng.integers(1, 101)) for _ in range(3))
            A = _draw(rng, r, mid, values=True)
            B = _draw(rng, mid, k, values=True)
            ah, bh = K.to_handle(A), K.to_handle(B)
            try:
                ch = K.mult_ab(ah, bh)
                C = K.from_handle(ch)
                K.release_handle(ch)
            finally:
                K.release_handle(ah)
                K.release_handle(bh)
            _, _, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values),
                                            (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values))
            assert np.array_equal(C.rowptrs, crp) and np.array_equal(C.colinds, cci)
            # (from_coo rows hold no column twice: the sums are the sequential loop's, bit for bit)
            assert np.array_equal(C.values.view(np.int64), np.asarray(cvs, dtype=np.float64).view(np.int64))
    finally:
        K.set_spgemm_order(None)


def test_sweep_spmm_heavy_rows():
    """
    The dense-panel SpMM with its heavy-row form forced on (CSRK_SPMM_HEAVY=1: B tiles in LDS, accumulators in dynamically
    indexed registers) over random shapes: 1 .. 8 row groups, panel widths around the 64-column launch and the odd / even
    tile loads, f4 / f8 / absent values, matrices narrower than a tile and tiles cut by the matrix' width.
    """
    import os
    from oracle import oracle as O
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(90210)
    old = {k: os.environ.get(k) for k in ('CSRK_SPMM_HEAVY', 'CSRK_SPMM_HEAVY_GROUPS')}
    os.environ['CSRK_SPMM_HEAVY'] = '1'
    try:
        for it in range(40):
            nrows = int(rng.integers(1, 2500))
            ncols = int(rng.integers(1, 3000))
            lens = np.minimum(rng.integers(0, 12, nrows), ncols)
            n_long = int(rng.integers(1, max(2, min(nrows, 700))))
            long_rows = rng.choice(nrows, min(n_long, nrows), replace=False)
            lens[long_rows] = np.minimum(rng.integers(256, 1200, len(long_rows)), ncols)
            rp = np.zeros(nrows + 1, np.int32)
            rp[1:] = np.cumsum(lens)
            ci = np.concatenate([rng.choice(ncols, int(n), replace=False) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
            kind = it % 3
            vals = None if kind == 2 else rng.uniform(-1, 1, len(ci)).astype(np.float32 if kind == 1 else np.float64)
            A = CSR(nrows, ncols, len(ci), rp, ci, vals, _cast=False)
            k = int(rng.choice([1, 2, 3, 8, 31, 64, 65, 100, 129]))
            B = rng.uniform(-1, 1, (ncols, k))
            os.environ['CSRK_SPMM_HEAVY_GROUPS'] = str(int(rng.integers(1, 9)))
            h = K.to_handle(A)
            try:
                Cm = K.mult_dense(h, B)
            finally:
                K.release_handle(h)
                K.invalidate(A)
            v64 = np.ones(len(ci)) if vals is None else vals.astype(np.float64)
            ref = O.spmm_dense(nrows, rp, ci, v64, B)
            bound = O.spmm_dense(nrows, rp, ci, np.abs(v64), np.abs(B))
            assert Cm.shape == ref.shape
            assert np.all(np.abs(Cm - ref) <= 1e-12 * bound + 1e-300), (it, nrows, ncols, k)
    finally:
        for k_, v_ in old.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
