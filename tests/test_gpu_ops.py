"""
GPU parity for transpose, row extents, row normalisation, SpGEMM, order_columns,
filter_zeros and the dense-panel SpMM -- all through the C ABI (csr_amd -> libcsrk.so)
against the CPU oracle and the golden vectors captured from the reference.

Bars: transpose / row_extent / row_nnzs / order_columns / filter_zeros are index work and
must be BIT-EXACT (values are copied bits).  SpGEMM and SpMM are fp64: |c - c_ref| <=
1e-6 * sum |a||b| (north_star's 1e-6 relative, stated against the magnitude of the summed
terms so it is meaningful under cancellation).  unit/center: rel 1e-6 for f8 (the reference
tests' own tolerance, tests/test_transform.py:128-149), 1e-5 for f4.
"""
import numpy as np
import pytest

from conftest import Mat, sort_within_rows, as_library_orders

pytestmark = pytest.mark.gpu


def _csr(m):
    from csr_amd import CSR
    return CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False)


def _rows_hold_no_column_twice(B):
    rows = np.repeat(np.arange(B.nrows, dtype=np.int64), np.diff(B.rowptrs))
    key = rows * B.ncols + B.colinds
    return np.unique(key).size == key.size


def _rand(rng, nrows, ncols, lens, dtype=np.float64, ptr64=False, values=True, sort=False):
    from csr_amd import CSR
    rp = np.zeros(nrows + 1, dtype=np.int64 if ptr64 else np.int32)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    ci = rng.integers(0, ncols, size=nnz).astype(np.int32)
    vs = rng.uniform(-1, 1, size=nnz).astype(dtype) if values else None
    m = CSR(nrows, ncols, nnz, rp, ci, vs, _cast=False)
    if sort:
        ci2, vs2 = sort_within_rows(rp, ci, vs)
        m = CSR(nrows, ncols, nnz, rp, ci2, vs2, _cast=False)
    return m


# ---- transpose ------------------------------------------------------------------------------

def test_transpose_kat(golden):
    "tests/test_transpose.py:11-46 of the reference"
    g = golden('kat')
    a = _csr(Mat(g, 'a_'))
    t = a.transpose()
    assert (t.nrows, t.ncols, t.nnz) == (3, 4, 4)
    assert list(t.rowptrs) == [0, 1, 3, 4]
    at = Mat(g, 'at_')
    assert np.array_equal(t.colinds, at.colinds) and np.array_equal(t.values, at.values)
    ts = a.transpose(False)
    assert ts.values is None and list(ts.rowptrs) == [0, 1, 3, 4]
    assert np.array_equal(ts.colinds, Mat(g, 'ats_').colinds)


def test_transpose_golden(golden):
    "bit-exact against the reference, incl. duplicate entries / unsorted rows / f4 / structure-only"
    g = golden('transpose')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        t = _csr(m).transpose()
        ref = Mat(g, f'c{c}_t_')
        assert (t.nrows, t.ncols, t.nnz) == (ref.nrows, ref.ncols, ref.nnz)
        assert t.rowptrs.dtype == m.rowptrs.dtype            # structure.py:175
        assert np.array_equal(t.rowptrs, ref.rowptrs), c
        assert np.array_equal(t.colinds, ref.colinds), c
        if ref.values is None:
            assert t.values is None
        else:
            assert t.values.dtype == np.float64                # structure.py:177
            assert np.array_equal(t.values, ref.values), c
        ts = _csr(m).transpose(False)
        rs = Mat(g, f'c{c}_ts_')
        assert ts.values is None
        assert np.array_equal(ts.rowptrs, rs.rowptrs) and np.array_equal(ts.colinds, rs.colinds)


@pytest.mark.parametrize('case', ['wide3pass', 'narrow1pass', 'ptr64_f32', 'skewed', 'empty'])
def test_transpose_vs_oracle(case):
    from oracle import oracle as O
    rng = np.random.default_rng(abs(hash(case)) % 2**32)
    if case == 'wide3pass':          # ncols > 65536: three radix passes
        m = _rand(rng, 3000, 200000, rng.integers(0, 60, 3000))
    elif case == 'narrow1pass':      # ncols <= 256: single pass; heavy duplicates
        m = _rand(rng, 5000, 200, rng.integers(0, 30, 5000))
    elif case == 'ptr64_f32':
        m = _rand(rng, 4000, 5000, rng.integers(0, 50, 4000), dtype=np.float32, ptr64=True)
    elif case == 'skewed':           # a few huge rows and a hot column
        lens = np.minimum((rng.pareto(0.8, 20000) * 2).astype(np.int64), 50000)
        m = _rand(rng, 20000, 30000, lens)
        m.colinds[::7] = 17
    else:
        m = _rand(rng, 100, 50, np.zeros(100, dtype=np.int64))
    t = m.transpose()
    nr, nc, brp, bci, bvs = O.transpose(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values)
    assert (t.nrows, t.ncols) == (nr, nc)
    assert t.rowptrs.dtype == m.rowptrs.dtype
    assert np.array_equal(t.rowptrs, brp)
    assert np.array_equal(t.colinds, bci)
    assert t.values.dtype == np.float64 and np.array_equal(t.values, bvs)
    # involution property (size independent): transposing twice sorts rows stably
    tt = t.transpose()
    ci2, vs2 = sort_within_rows(m.rowptrs, m.colinds, m.values)
    assert np.array_equal(tt.rowptrs, m.rowptrs) and np.array_equal(tt.colinds, ci2)
    assert np.array_equal(tt.values, vs2.astype(np.float64))


# ---- row extents -------------------------------------------------------------------------------

def test_row_extent_and_nnzs(golden):
    "tests/test_attributes.py:36-54 of the reference"
    from csr_amd.kernels import hip as K
    g = golden('kat')
    a = _csr(Mat(g, 'a_'))
    h = K.to_handle(a)
    try:
        assert [K.row_extent(h, i) for i in range(4)] == [(0, 2), (2, 3), (3, 3), (3, 4)]
        assert np.array_equal(K.row_nnzs(h), g['a_row_nnzs'])
        with pytest.raises(ValueError):
            K.row_extent(h, 4)
    finally:
        K.release_handle(h)
    g = golden('transpose')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        h = K.to_handle(_csr(m))
        try:
            out = K.row_nnzs(h)
            assert out.dtype == m.rowptrs.dtype and np.array_equal(out, g[f'c{c}_row_nnzs'])
        finally:
            K.release_handle(h)


# ---- handles -------------------------------------------------------------------------------------

def test_handle_roundtrip(golden):
    "tests/test_handles.py:10-21 and tests/test_mkl.py:51-63 of the reference"
    from csr_amd.kernels import hip as K
    g = golden('spmv')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        h = K.to_handle(_csr(m))
        try:
            c2 = K.from_handle(h)
            assert (c2.nrows, c2.ncols, c2.nnz) == (m.nrows, m.ncols, m.nnz)
            assert np.array_equal(c2.rowptrs, m.rowptrs) and np.array_equal(c2.colinds, m.colinds)
            if m.values is None:
                assert c2.values is None
            else:
                assert c2.values.dtype == m.values.dtype and np.array_equal(c2.values, m.values)
        finally:
            K.release_handle(h)
    with pytest.raises(ValueError):
        K.mult_vec(h, np.ones(m.ncols))       # released handle


# ---- unit / center -----------------------------------------------------------------------------------

def test_unit_center_golden(golden):
    g = golden('rows')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        f4 = m.values.dtype == np.float32
        rel = 1e-5 if f4 else 1e-6
        a = _csr(m).copy()
        with np.errstate(all='ignore'):
            norms = a.normalize_rows('unit')
        assert norms.dtype == m.values.dtype and a.values.dtype == m.values.dtype
        assert norms == pytest.approx(g[f'c{c}_unit_norms'], rel=rel, abs=0, nan_ok=True), c
        ref_v = g[f'c{c}_unit_values']
        assert np.array_equal(np.isnan(a.values), np.isnan(ref_v)), c      # zero-norm rows -> NaN
        assert a.values == pytest.approx(ref_v, rel=rel, abs=1e-300, nan_ok=True), c
        b = _csr(m).copy()
        means = b.normalize_rows('center')
        scale = float(np.max(np.abs(m.values))) if m.nnz else 1.0
        tol = scale * (1e-6 if f4 else 1e-12)
        assert means.dtype == m.values.dtype
        assert means == pytest.approx(g[f'c{c}_center_means'], rel=rel, abs=tol), c
        assert b.values == pytest.approx(g[f'c{c}_center_values'], rel=rel, abs=tol), c


def test_unit_rows_properties_large():
    "tests/test_transform.py:128-149 of the reference, at a size with long and empty rows"
    rng = np.random.default_rng(5)
    lens = np.minimum((rng.pareto(0.9, 30000) * 3).astype(np.int64), 40000)
    m = _rand(rng, 30000, 1000, lens)
    m.values[:] *= rng.choice([1e-200, 1.0, 1e150], size=m.nnz)
    orig = m.values.copy()
    norms = m.normalize_rows('unit')
    rows = np.repeat(np.arange(m.nrows), np.diff(m.rowptrs))
    ss = np.zeros(m.nrows)
    np.add.at(ss, rows, m.values ** 2)
    nz = np.diff(m.rowptrs) > 0
    assert np.all(norms[~nz] == 0)
    assert ss[nz] == pytest.approx(1.0, rel=1e-9)
    assert m.values * norms[rows] == pytest.approx(orig, rel=1e-9)


# ---- SpGEMM ------------------------------------------------------------------------------------------

def _dense(nr, nc, rp, ci, vs):
    out = np.zeros((nr, nc))
    rows = np.repeat(np.arange(nr), np.diff(rp))
    np.add.at(out, (rows, ci), vs)
    return out


def test_spgemm_golden(golden):
    "kernel-level mult_ab / mult_abt and CSR.multiply against the reference's outputs"
    from csr_amd.kernels import hip as K
    g = golden('spgemm')
    for c in range(int(g['n'])):
        A, B = Mat(g, f'c{c}_a_'), Mat(g, f'c{c}_b_')
        raw = Mat(g, f'c{c}_raw_')
        bound = 1e-6 * (np.abs(A.dense()) @ np.abs(B.dense())) + 1e-300
        ah, bh = K.to_handle(_csr(A)), K.to_handle(_csr(B))
        try:
            ch = K.mult_ab(ah, bh)
            C = K.from_handle(ch)
            K.release_handle(ch)
        finally:
            K.release_handle(ah)
            K.release_handle(bh)
        # structure: same rowptrs and the same column SET per row as the reference (explicit
        # zeros kept); order inside a row is ascending here
        assert C.rowptrs.dtype == np.int32 and np.array_equal(C.rowptrs, raw.rowptrs), c
        rci, rvs = as_library_orders(raw.rowptrs, raw.colinds, raw.values)
        assert np.array_equal(C.colinds, rci), c
        assert np.all(np.abs(C.values - rvs) <= bound[np.repeat(np.arange(C.nrows), np.diff(C.rowptrs)), C.colinds]), c
        if _rows_hold_no_column_twice(B):
            # the reference's order of addition AND its product precision (f4 * f4 rounded to f4, multiply.py:120): the same bits
            assert np.array_equal(C.values.view(np.int64), rvs.view(np.int64)), c
        # CSR.multiply: filtered product (csr/csr.py:555), tests/test_multiply.py:14-44
        P = _csr(A).multiply(_csr(B))
        ab = Mat(g, f'c{c}_ab_')
        assert (P.nrows, P.ncols) == (ab.nrows, ab.ncols)
        if P.nnz:
            assert np.all(P.values != 0)
        assert np.all(np.abs(_dense(P.nrows, P.ncols, P.rowptrs, P.colinds, P.values) - ab.dense()) <= bound), c
        # A B^T, tests/test_multiply.py:47-79
        Bt = Mat(g, f'c{c}_bt_')
        Pt = _csr(A).multiply(_csr(Bt), transpose=True)
        abt = Mat(g, f'c{c}_abt_')
        assert (Pt.nrows, Pt.ncols) == (abt.nrows, abt.ncols)
        assert np.all(np.abs(_dense(Pt.nrows, Pt.ncols, Pt.rowptrs, Pt.colinds, Pt.values) - abt.dense()) <= bound), c


@pytest.mark.parametrize('case', ['sparse', 'heavy_rows', 'f32_ptr64'])
def test_spgemm_vs_oracle(case):
    "larger products: LDS-hash rows and dense-accumulator rows (product count > 1024) together"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(abs(hash(case)) % 2**32)
    if case == 'sparse':
        A = _rand(rng, 3000, 2000, rng.integers(0, 12, 3000), sort=True)
        B = _rand(rng, 2000, 5000, rng.integers(0, 12, 2000), sort=True)
    elif case == 'heavy_rows':
        la = rng.integers(0, 8, 1500)
        la[::50] = 400
        A = _rand(rng, 1500, 1200, la)
        lb = rng.integers(0, 30, 1200)
        lb[::40] = 900
        B = _rand(rng, 1200, 3000, lb)
    else:
        A = _rand(rng, 500, 400, rng.integers(0, 20, 500), dtype=np.float32, ptr64=True)
        B = _rand(rng, 400, 300, rng.integers(0, 20, 400), dtype=np.float32)
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        ch = K.mult_ab(ah, bh)
        C = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values),
                                      (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values))
    rci, rvs = as_library_orders(crp, cci, cvs)
    assert np.array_equal(C.rowptrs, crp) and np.array_equal(C.colinds, rci)
    _, _, _, _, babs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, np.abs(A.values)),
                                 (B.nrows, B.ncols, B.rowptrs, B.colinds, np.abs(B.values)))
    _, bsorted = as_library_orders(crp, cci, babs)
    assert np.all(np.abs(C.values - rvs) <= 1e-6 * bsorted + 1e-300)
    assert np.all(np.abs(C.values - rvs) <= 1e-12 * bsorted + 1e-300)
    if _rows_hold_no_column_twice(B):
        # the sums are taken in the reference's order (csrc/spgemm.hip, "Determinism"): the same bits
        assert np.array_equal(C.values.view(np.int64), rvs.view(np.int64))


@pytest.fixture
def reference_order():
    "mult_ab / mult_abt emit the reference's column order inside rows for the duration of a test"
    from csr_amd.kernels import hip as K
    K.set_spgemm_order('reference')
    yield K
    K.set_spgemm_order(None)


def test_spgemm_reference_order_golden(golden, reference_order):
    """
    _sym_mm emits each row's columns in REVERSE order of first discovery (csr/kernels/numba/multiply.py:79-82, 94-97).
    With the reference order switched on, mult_ab returns the reference's own raw arrays: row pointers and column
    indices bit for bit, values bit for bit wherever B's rows hold no column twice.
    """
    K = reference_order
    g = golden('spgemm')
    n_reordered = 0
    for c in range(int(g['n'])):
        A, B, raw = Mat(g, f'c{c}_a_'), Mat(g, f'c{c}_b_'), Mat(g, f'c{c}_raw_')
        ah, bh = K.to_handle(_csr(A)), K.to_handle(_csr(B))
        try:
            ch = K.mult_ab(ah, bh)
            C = K.from_handle(ch)
            K.release_handle(ch)
        finally:
            K.release_handle(ah)
            K.release_handle(bh)
        assert np.array_equal(C.rowptrs, raw.rowptrs) and np.array_equal(C.colinds, raw.colinds), c
        sci, _ = sort_within_rows(raw.rowptrs, raw.colinds, raw.values)
        n_reordered += int(not np.array_equal(sci, raw.colinds))
        if _rows_hold_no_column_twice(B):
            assert np.array_equal(C.values.view(np.int64), raw.values.view(np.int64)), c
        else:
            bound = 1e-6 * (np.abs(A.dense()) @ np.abs(B.dense())) + 1e-300
            assert np.all(np.abs(C.values - raw.values) <= bound[np.repeat(np.arange(C.nrows), np.diff(C.rowptrs)), C.colinds]), c
    assert n_reordered > 0            # the fixtures do hold rows whose reference order is not ascending


@pytest.mark.parametrize('case', ['sparse', 'heavy_rows', 'unsorted_b', 'abt_block', 'wide', 'wider', 'ptr64', 'ptr64_wide'])
def test_spgemm_reference_order_vs_oracle(case, reference_order):
    "the same on products that take every accumulator path (hash, strips, expand-sort-compress), against the oracle's raw output"
    from oracle import oracle as O
    from csr_amd import CSR, synth
    K = reference_order
    rng = np.random.default_rng(abs(hash('ro' + case)) % 2**32)
    abt = case == 'abt_block'
    if case == 'sparse':
        A = _rand(rng, 3000, 2000, rng.integers(0, 12, 3000), sort=True)
        B = _rand(rng, 2000, 5000, rng.integers(0, 12, 2000), sort=True)
    elif case == 'heavy_rows':
        la = rng.integers(0, 8, 1500)
        la[::50] = 400
        A = _rand(rng, 1500, 1200, la)
        lb = rng.integers(0, 30, 1200)
        lb[::40] = 900
        B = _rand(rng, 1200, 3000, lb)
    elif case == 'unsorted_b':        # discovery order follows B's STORAGE order, not its columns
        A = _rand(rng, 800, 600, rng.integers(0, 25, 800))
        B = _rand(rng, 600, 900, rng.integers(0, 40, 600))
    elif case in ('ptr64', 'ptr64_wide'):      # int64 row pointers on both operands, rows for every tier of the ordering pass
        nc = 2500 if case == 'ptr64' else 40000   # (by column / by entry)
        la = rng.integers(0, 9, 700)
        la[::45] = 300
        la[::7] = 40
        A = _rand(rng, 700, 500, la, ptr64=True)
        lb = rng.integers(0, 30, 500)
        lb[::25] = 1500
        B = _rand(rng, 500, nc, lb, ptr64=True)
    elif case in ('wide', 'wider'):   # a product too wide for the discovery pass's per-column LDS tables (30 000 columns: the
        # rows of C still fit LDS for the bisection; 70 000: 16-bit positions do not reach, bisection in memory for long rows)
        nc = 30000 if case == 'wide' else 70000
        la = rng.integers(0, 10, 900)
        la[::60] = 300
        A = _rand(rng, 900, 700, la)
        lb = rng.integers(0, 60, 700)
        lb[::35] = 12000 if case == 'wide' else 45000
        B = _rand(rng, 700, nc, lb)
    else:
        m = synth.movielens_like(device='cpu')
        M = CSR(m['nrows'], m['ncols'], int(m['colinds'].numel()), m['rowptrs'].numpy(), m['colinds'].numpy(),
                m['values'].numpy(), _cast=False)
        A, B = M.subset_rows(0, 600), M.subset_rows(0, 8000)
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        ch = K.mult_abt(ah, bh) if abt else K.mult_ab(ah, bh)
        C = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    b = O.transpose(B.nrows, B.ncols, B.rowptrs, B.colinds, B.values) if abt else (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values)
    _, _, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), b)
    assert np.array_equal(C.rowptrs, crp) and np.array_equal(C.colinds, cci)
    assert not np.array_equal(sort_within_rows(crp, cci, cvs)[0], cci)
    if abt or _rows_hold_no_column_twice(B):
        assert np.array_equal(C.values.view(np.int64), cvs.view(np.int64))
    # ascending again once the switch is off
    K.set_spgemm_order('ascending')
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        ch = K.mult_abt(ah, bh) if abt else K.mult_ab(ah, bh)
        C2 = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    assert np.array_equal(C2.colinds, sort_within_rows(crp, cci, cvs)[0])


def test_mult_abt_movielens_shape_blocks():
    """
    BASELINE.json configs[4], the mult_abt half, on its own workload shape: row blocks of a MovieLens-25M-shaped
    matrix (162 541 x 59 047, both degree distributions power-law, values in {0.5 .. 5.0}) multiplied as
    A[block] . B[block]^T through csrk_spgemm_abt (csr/kernels/numba/multiply.py:41-57 = mult_ab(A, transpose(B))),
    blocks because the product's row pointers are int32 (multiply.py:28).  Against the oracle's own
    transpose + SMMP: row pointers bit-equal, column sets equal (ascending here, reverse-discovery there),
    values to 1e-12 of sum |a||b|.
    """
    from oracle import oracle as O
    from csr_amd import CSR, synth
    from csr_amd.kernels import hip as K
    m = synth.movielens_like(device='cpu')
    M = CSR(m['nrows'], m['ncols'], int(m['colinds'].numel()), m['rowptrs'].numpy(), m['colinds'].numpy(),
            m['values'].numpy(), _cast=False)
    assert (M.nrows, M.ncols) == (162541, 59047)
    for a0, a1, b0, b1 in ((0, 2000, 0, 20000), (90000, 90300, 40000, 46000)):
        A, B = M.subset_rows(a0, a1), M.subset_rows(b0, b1)
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
        assert (Cm.nrows, Cm.ncols) == (A.nrows, B.nrows) == (nr, nc)
        assert Cm.rowptrs.dtype == np.int32 and np.array_equal(Cm.rowptrs, crp)
        rci, rvs = as_library_orders(crp, cci, cvs)
        assert np.array_equal(Cm.colinds, rci)
        _, _, _, _, babs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, np.abs(A.values)),
                                     (bt[0], bt[1], bt[2], bt[3], np.abs(bt[4])))
        _, bsorted = as_library_orders(crp, cci, babs)
        assert np.all(np.abs(Cm.values - rvs) <= 1e-12 * bsorted + 1e-300)
        assert np.array_equal(Cm.values.view(np.int64), rvs.view(np.int64))      # same order of addition: same bits
        # the caller's path (csr/csr.py:524-567): same product after _filter_zeros
        P = A.multiply(B, transpose=True)
        keep = rvs != 0.0
        assert P.nnz == int(keep.sum()) and np.array_equal(P.colinds, rci[keep])


@pytest.mark.parametrize('paths', ['default', 'two-pass strips', 'fallbacks'])
def test_spgemm_deterministic(paths, monkeypatch):
    """
    Every SpGEMM path adds an output entry's products in ascending order of A's entries (csrc/spgemm.hip, "Determinism"):
    the values are bitwise reproducible run to run, and -- B's rows holding no column twice -- bit-identical to the
    reference's sequential loop (multiply.py:117-121), which the oracle restates.  'default': wave-per-row, workgroup
    hash, column strips, expand-sort-compress; 'fallbacks' (CSRK_SPGEMM_STRIPS=0, CSRK_SPGEMM_ESC=0, what runs when B's
    rows are unsorted or the products exceed the sort's budget): LDS tiles, the big LDS hash table, HBM work rows.
    """
    if paths == 'two-pass strips':          # (the strips' form for products whose temporary exceeds the budget)
        monkeypatch.setenv('CSRK_SPGEMM_FUSED', '0')            # (strips and wave-per-row kernels in two passes)
    if paths == 'fallbacks':
        monkeypatch.setenv('CSRK_SPGEMM_FUSED', '0')
        monkeypatch.setenv('CSRK_SPGEMM_STRIPS', '0')
        monkeypatch.setenv('CSRK_SPGEMM_ESC', '0')
    from oracle import oracle as O
    from csr_amd import CSR, synth
    from csr_amd.kernels import hip as K

    def uniq(rng, nrows, ncols, lens):
        rp = np.zeros(nrows + 1, dtype=np.int32)
        rp[1:] = np.cumsum(lens)
        ci = np.concatenate([np.sort(rng.choice(ncols, size=int(n), replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        return CSR(nrows, ncols, int(rp[-1]), rp, ci, rng.uniform(-1, 1, size=int(rp[-1])), _cast=False)

    rng = np.random.default_rng(99)
    cases = []
    # short rows (wave kernels), a few hundred products (workgroup hash), thousands of products onto few columns (big hash)
    la = rng.integers(0, 10, 1200)
    la[::40] = 300
    lb = rng.integers(0, 12, 900)
    lb[::30] = 200
    cases.append((uniq(rng, 1200, 900, la), uniq(rng, 900, 6000, lb)))
    # nearly full output rows (LDS tiles) and wide sparse ones (HBM work rows): 40000 output columns
    la = rng.integers(0, 6, 300)
    la[::10] = 700
    lb = rng.integers(0, 40, 2500)
    lb[::7] = 9000
    cases.append((uniq(rng, 300, 2500, la), uniq(rng, 2500, 40000, lb)))
    m = synth.movielens_like(device='cpu')
    M = CSR(m['nrows'], m['ncols'], int(m['colinds'].numel()), m['rowptrs'].numpy(), m['colinds'].numpy(),
            m['values'].numpy(), _cast=False)
    for A, B, abt in [(a, b, False) for a, b in cases] + [(M.subset_rows(500, 700), M.subset_rows(30000, 33000), True)]:
        outs = []
        for _ in range(3):
            ah, bh = K.to_handle(A), K.to_handle(B)
            try:
                ch = K.mult_abt(ah, bh) if abt else K.mult_ab(ah, bh)
                outs.append(K.from_handle(ch))
                K.release_handle(ch)
            finally:
                K.release_handle(ah)
                K.release_handle(bh)
        for c in outs[1:]:
            assert np.array_equal(c.rowptrs, outs[0].rowptrs) and np.array_equal(c.colinds, outs[0].colinds)
            assert np.array_equal(c.values.view(np.int64), outs[0].values.view(np.int64))      # bit for bit, run to run
        bt = O.transpose(B.nrows, B.ncols, B.rowptrs, B.colinds, B.values) if abt else (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values)
        nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), bt)
        rci, rvs = as_library_orders(crp, cci, cvs)
        assert np.array_equal(outs[0].rowptrs, crp) and np.array_equal(outs[0].colinds, rci)
        assert np.array_equal(outs[0].values.view(np.int64), rvs.view(np.int64))                  # = the sequential loop's bits


def test_multiply_sharded(golden):
    "csr/csr.py:558-567: row sharding above max_nnz + _assemble_shards (tests/test_mkl.py:82-91)"
    from csr_amd.kernels import hip as K
    g = golden('spgemm')
    save = K.max_nnz
    hits = 0
    try:
        for c in range(int(g['n'])):
            A, B = Mat(g, f'c{c}_a_'), Mat(g, f'c{c}_b_')
            # B must fit (the reference test assumes B.nnz < max_nnz, tests/test_multiply.py:18)
            lim = max(B.nnz, int(np.max(np.diff(A.rowptrs), initial=0)), 20)
            if A.nnz <= 2 * lim:
                continue
            hits += 1
            K.max_nnz = lim
            P = _csr(A).multiply(_csr(B))
            ab = Mat(g, f'c{c}_ab_')
            bound = 1e-6 * (np.abs(A.dense()) @ np.abs(B.dense())) + 1e-300
            assert (P.nrows, P.ncols) == (ab.nrows, ab.ncols)
            assert np.all(np.abs(_dense(P.nrows, P.ncols, P.rowptrs, P.colinds, P.values) - ab.dense()) <= bound)
    finally:
        K.max_nnz = save
    assert hits >= 2


# ---- order_columns / filter_zeros ---------------------------------------------------------------------

def test_order_columns_golden(golden):
    "tests/test_transform.py:77-87 of the reference: bit-exact with sort_rows on unsorted rows"
    from csr_amd.kernels import hip as K
    g = golden('spgemm')
    for c in range(int(g['n'])):
        raw, srt = Mat(g, f'c{c}_raw_'), Mat(g, f'c{c}_rawsorted_')
        h = K.to_handle(_csr(raw))
        try:
            K.order_columns(h)
            c2 = K.from_handle(h)
        finally:
            K.release_handle(h)
        assert np.array_equal(c2.rowptrs, srt.rowptrs)
        assert np.array_equal(c2.colinds, srt.colinds) and np.array_equal(c2.values, srt.values)
        assert all(np.all(np.diff(c2.colinds[c2.rowptrs[i]:c2.rowptrs[i + 1]]) > 0) for i in range(c2.nrows))
    # f4 values and structure-only survive the round trip exactly
    rng = np.random.default_rng(2)
    for dt, vals in ((np.float32, True), (np.float64, False)):
        m = _rand(rng, 300, 200, rng.integers(0, 40, 300), dtype=dt, values=vals)
        ci, vs = sort_within_rows(m.rowptrs, m.colinds, m.values)
        m.sort_rows()
        assert np.array_equal(m.colinds, ci)
        if vals:
            assert m.values.dtype == dt and np.array_equal(m.values, vs)


def test_filter_zeros(golden):
    "csr/_struct.py:61-76 on the device: bit-exact"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(8)
    m = _rand(rng, 2000, 500, rng.integers(0, 20, 2000))
    m.values[rng.uniform(size=m.nnz) < 0.3] = 0.0
    m.values[5] = np.nan
    m.values[7] = -0.0
    h = K.to_handle(m)
    try:
        fh = K.filter_zeros(h)
        f = K.from_handle(fh)
        K.release_handle(fh)
    finally:
        K.release_handle(h)
    rp, ci, vs = O.filter_zeros(m.nrows, m.rowptrs, m.colinds, m.values)
    assert f.nnz == len(ci)
    assert np.array_equal(f.rowptrs, rp) and np.array_equal(f.colinds, ci)
    assert np.array_equal(f.values, vs, equal_nan=True)


# ---- pick_rows ----------------------------------------------------------------------------------------------

def test_pick_rows_golden(golden):
    "csr/csr.py:347-364 on the device against the reference's outputs: index and byte work, bit-exact"
    from csr_amd.kernels import hip as K
    g = golden('pick')
    for c in range(int(g['n'])):
        m, out = Mat(g, f'c{c}_'), Mat(g, f'c{c}_out_')
        rows, include = g[f'c{c}_rows'], bool(g[f'c{c}_include'])
        h = K.to_handle(_csr(m))
        try:
            ph = K.pick_rows(h, rows, include)
            sub = K.from_handle(ph)
            K.release_handle(ph)
        finally:
            K.release_handle(h)
        assert (sub.nrows, sub.ncols, sub.nnz) == (len(rows), m.ncols, out.nnz)
        assert np.array_equal(sub.rowptrs, out.rowptrs) and np.array_equal(sub.colinds, out.colinds)
        if include and m.values is not None:
            assert sub.values.dtype == m.values.dtype and np.array_equal(sub.values, out.values)
        else:
            assert sub.values is None


def test_pick_rows_large_and_errors():
    "power-law rows incl. a 200k-entry row, int64 pointers, CSR.pick_rows; out-of-range index -> error"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(31)
    lens = np.minimum((rng.pareto(0.9, 20000) * 3).astype(np.int64), 200000)
    lens[77] = 200000
    for ptr64, dt in ((False, np.float64), (True, np.float32)):
        m = _rand(rng, 20000, 50000, lens, dtype=dt, ptr64=ptr64)
        rows = rng.integers(0, m.nrows, size=30000).astype(np.int32)
        rows[:3] = 77                                   # the long row, three times
        sub = m.pick_rows(rows)
        rp, ci, vs = O.pick_rows(m.rowptrs, m.colinds, m.values, rows, True)
        assert sub.nrows == len(rows) and np.array_equal(sub.rowptrs, rp)
        assert np.array_equal(sub.colinds, ci) and sub.values.dtype == dt and np.array_equal(sub.values, vs)
        nv = m.pick_rows(rows[:100], include_values=False)
        assert nv.values is None and np.array_equal(nv.colinds, ci[:rp[100]])
    h = K.to_handle(m)
    try:
        with pytest.raises(ValueError):          # CSRK_ERR_INVALID (the reference: IndexError)
            K.pick_rows(h, np.array([0, m.nrows], dtype=np.int32))
        with pytest.raises(ValueError):
            K.pick_rows(h, np.array([-1], dtype=np.int32))
    finally:
        K.release_handle(h)


def test_in_place_ops_invalidate_the_spmv_plan(monkeypatch):
    """
    unit_rows / center_rows / order_columns change values or columns in place; the SpMV plan holds re-ordered
    copies of them (tiers, light stream), so a later mult_vec on the same handle must see the new matrix.
    """
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    monkeypatch.setenv('CSRK_SPMV_HEAVY_SPLIT', '1')
    monkeypatch.setenv('CSRK_SPMV_HOT', '1')
    rng = np.random.default_rng(12)
    lens = rng.integers(1, 30, size=3000)
    lens[5] = 4000
    lens[9] = 600
    m = _rand(rng, 3000, 20000, lens, sort=True)     # ascending columns: the long rows go to the tiers
    x = rng.uniform(-1, 1, size=m.ncols)
    h = K.to_handle(m)
    try:
        for _ in range(2):
            y0 = K.mult_vec(h, x)                       # plan built (eagerly: split + stream)
        ref0 = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
        assert np.allclose(y0, ref0, rtol=1e-12, atol=1e-12)
        K.unit_rows(h)
        u = K.from_handle(h)
        y1 = K.mult_vec(h, x)
        ref1 = O.mult_vec(u.nrows, u.ncols, u.rowptrs, u.colinds, u.values, x)
        assert np.allclose(y1, ref1, rtol=1e-12, atol=1e-12) and not np.allclose(y1, y0)
        K.order_columns(h)
        s2 = K.from_handle(h)
        y2 = K.mult_vec(h, x)
        ref2 = O.mult_vec(s2.nrows, s2.ncols, s2.rowptrs, s2.colinds, s2.values, x)
        assert np.allclose(y2, ref2, rtol=1e-12, atol=1e-12)
    finally:
        K.release_handle(h)


# ---- dense-panel SpMM ------------------------------------------------------------------------------------

def test_spmm_dense_golden(golden):
    """
    csrk_spmm_dense against the reference's own mult_ab(A, CSR(B)) with B fully populated
    (csr/kernels/numba/multiply.py:13-38, :110-122), densified: k in {1, 7, 64}, f8 and f4 A.  fp64: 1e-6 of the
    summed magnitudes (north_star), asserted at 1e-12.
    """
    from csr_amd.kernels import hip as K
    g = golden('spmm_dense')
    for c in range(int(g['n'])):
        a = Mat(g, f'c{c}_a_')
        B, Cref = g[f'c{c}_B'], g[f'c{c}_C']
        h = K.to_handle(_csr(a))
        try:
            Cm = K.mult_dense(h, B)
        finally:
            K.release_handle(h)
        bound = np.abs(a.dense()) @ np.abs(B)
        assert Cm.shape == Cref.shape
        assert np.all(np.abs(Cm - Cref) <= 1e-12 * bound + 1e-300), c


def _panel_csr(B, cols=None, vals=None, ptr64=False):
    "B [n x k] as the fully populated CSR a reference caller hands to mult_ab (row-major unless cols / vals are given)"
    from csr_amd import CSR
    n, k = B.shape
    rp = np.arange(n + 1, dtype=np.int64 if ptr64 else np.int32) * k
    cols = np.tile(np.arange(k, dtype=np.int32), n) if cols is None else cols
    vals = B.reshape(-1).copy() if vals is None else vals
    return CSR(n, k, n * k, rp, cols, vals, _cast=False)


def test_mult_ab_with_a_dense_b_golden(golden):
    """
    BASELINE configs[2] through the reference's OWN entry point: K.mult_ab(A, CSR(B)) with B fully populated
    (csr/csr.py:524-567 -> csr/kernels/numba/multiply.py:13-38).  The library recognises the row-major panel on the
    device and runs the dense-panel kernels; what comes back are the reference's raw arrays bit for bit (fixtures:
    oracle/gen/gen_golden.py gen_spmm_dense, captured from the imported reference): int32 rowptrs with k entries per row
    of C whose row of A holds an entry, colinds k - 1 .. 0, explicit zeros kept (case 5: a zero row of B), values equal
    bit for bit (every fixture row has at most 64 entries: one segment, the reference's order of additions).
    f8 and f4 values on A, k in {1, 7, 64}; case 12 (rows of B in shuffled column order) must take the general product
    and still return the reference's arrays; case 13 has blocks of empty rows in A.
    """
    from csr_amd.kernels import hip as K
    g = golden('spmm_dense')
    routes = []
    for c in range(int(g['n'])):
        a = Mat(g, f'c{c}_a_')
        B = g[f'c{c}_B']
        bh = K.to_handle(_panel_csr(B, g[f'c{c}_b_colinds'], g[f'c{c}_b_values']))
        ah = K.to_handle(_csr(a))
        try:
            ch = K.mult_ab(ah, bh)
            routes.append(K.spgemm_last_route())
            C = K.from_handle(ch)
            K.release_handle(ch)
        finally:
            K.release_handle(ah)
            K.release_handle(bh)
        assert (C.nrows, C.ncols, C.nnz) == (a.nrows, B.shape[1], int(g[f'c{c}_raw_nnz'])), c
        assert C.rowptrs.dtype == np.int32 and np.array_equal(C.rowptrs, g[f'c{c}_raw_rowptrs']), c
        assert np.array_equal(C.colinds, g[f'c{c}_raw_colinds']), c
        assert C.values.dtype == np.float64 and np.array_equal(C.values, g[f'c{c}_raw_values']), c
    assert routes[12] == 'general' and all(r == 'dense-panel' for i, r in enumerate(routes) if i != 12 and int(g[f'c{i}_a_shape'][2]) > 0)


@pytest.mark.parametrize('k,dtype,ptr64', [(64, np.float64, False), (7, np.float32, False), (130, np.float64, True), (1, np.float64, False)])
def test_mult_ab_with_a_dense_b_vs_oracle(k, dtype, ptr64, monkeypatch):
    """
    The same route on a power-law A with rows far beyond one segment (and, forced, the heavy-row kernels), against the
    pinned oracle's orc_mult_ab on the same two CSR operands: rowptrs and colinds bit for bit, values bit for bit on the
    rows of at most 64 entries and within 1e-12 of sum |a b| on the longer ones (the dense-panel kernels add a long row's
    partial sums in their own fixed order).  CSR.multiply on top of it drops the exact zeros like the reference does
    (csr/csr.py:555).  int64 row pointers on both operands in one case; a float32 B under float64 A is widened.
    """
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    monkeypatch.setenv('CSRK_SPMM_HEAVY', '1')
    rng = np.random.default_rng(1000 + k)
    lens = np.minimum((rng.pareto(0.9, 6000) * 3).astype(np.int64), 7000)
    lens[rng.integers(0, 6000, 900)] = 0
    A = _rand(rng, 6000, 5000, lens, dtype=dtype, ptr64=ptr64)
    B = rng.uniform(-1, 1, (A.ncols, k)).astype(np.float32 if k == 130 else np.float64)
    B[17, :] = 0.0
    Bc = _panel_csr(B, ptr64=ptr64)
    ah, bh = K.to_handle(A), K.to_handle(Bc)
    try:
        ch = K.mult_ab(ah, bh)
        route = K.spgemm_last_route()
        C = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    assert route == 'dense-panel'
    _, _, rp, ci, vs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), (Bc.nrows, Bc.ncols, Bc.rowptrs, Bc.colinds, Bc.values))
    assert C.rowptrs.dtype == np.int32 and np.array_equal(C.rowptrs, rp) and np.array_equal(C.colinds, ci)
    _, _, _, _, bound = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, np.abs(A.values)),
                                  (Bc.nrows, Bc.ncols, Bc.rowptrs, Bc.colinds, np.abs(Bc.values)))
    assert np.all(np.abs(C.values - vs) <= 1e-12 * bound + 1e-300)
    short = np.repeat(np.diff(A.rowptrs) <= 64, np.diff(rp))
    assert short.any() and (~short).any() and np.array_equal(C.values[short], vs[short])
    # the caller's product: exact zeros dropped, as csr/csr.py:555 does
    P = A.multiply(Bc)
    keep = vs != 0.0
    assert P.nnz == int(keep.sum()) and np.array_equal(P.colinds, ci[keep]) and np.array_equal(P.values, C.values[keep])


def test_mult_ab_dense_route_declines(monkeypatch):
    "what is NOT a row-major panel takes the general product: a row short of one column, float32 on both operands, the switch"
    from oracle import oracle as O
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(77)
    A = _rand(rng, 300, 200, rng.integers(0, 30, 300))
    B = rng.uniform(-1, 1, (200, 8))

    def run(a, b):
        ah, bh = K.to_handle(a), K.to_handle(b)
        try:
            ch = K.mult_ab(ah, bh)
            r = K.spgemm_last_route()
            c = K.from_handle(ch)
            K.release_handle(ch)
        finally:
            K.release_handle(ah)
            K.release_handle(bh)
        _, _, rp, ci, vs = O.mult_ab((a.nrows, a.ncols, a.rowptrs, a.colinds, a.values), (b.nrows, b.ncols, b.rowptrs, b.colinds, b.values))
        assert np.array_equal(c.rowptrs, rp) and np.array_equal(c.colinds, ci) and np.array_equal(c.values, vs)
        return r

    full = _panel_csr(B)
    assert run(A, full) == 'dense-panel'
    # the same entries minus one: nnz no longer nrows * ncols
    keep = np.ones(full.nnz, dtype=bool)
    keep[8 * 50 + 3] = False
    rp = np.concatenate([[0], np.cumsum(np.bincount(np.repeat(np.arange(200), 8)[keep], minlength=200))]).astype(np.int32)
    assert run(A, CSR(200, 8, full.nnz - 1, rp, full.colinds[keep], full.values[keep], _cast=False)) == 'general'
    # right count, wrong place: one row holds a column twice and lacks another
    cols = full.colinds.copy()
    cols[8 * 20 + 5] = 4
    assert run(A, CSR(200, 8, full.nnz, full.rowptrs, cols, full.values, _cast=False)) == 'general'
    # float32 x float32: the reference rounds every product to float32 (multiply.py:120)
    A32 = CSR(A.nrows, A.ncols, A.nnz, A.rowptrs, A.colinds, A.values.astype(np.float32), _cast=False)
    B32 = CSR(200, 8, full.nnz, full.rowptrs, full.colinds, full.values.astype(np.float32), _cast=False)
    assert run(A32, B32) == 'general'
    # the look at B is remembered by its handle -- and forgotten when an in-place operation changes it: a B whose rows are
    # complete but shuffled takes the general product; after order_columns on the same handle it IS a row-major panel
    rngp = np.random.default_rng(78)
    cols_s, vals_s = full.colinds.copy(), full.values.copy()
    for j in range(200):
        o = rngp.permutation(8)
        cols_s[8 * j:8 * j + 8] = o
        vals_s[8 * j:8 * j + 8] = B[j, o]
    shuf = CSR(200, 8, full.nnz, full.rowptrs, cols_s, vals_s, _cast=False)
    ah, bh = K.to_handle(A), K.to_handle(shuf)
    try:
        routes = []
        for step in range(4):
            if step == 2:
                K.order_columns(bh)
            ch = K.mult_ab(ah, bh)
            routes.append(K.spgemm_last_route())
            c = K.from_handle(ch)
            K.release_handle(ch)
            bb = shuf if step < 2 else full
            _, _, rp, ci, vs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), (200, 8, bb.rowptrs, bb.colinds, bb.values))
            assert np.array_equal(c.rowptrs, rp) and np.array_equal(c.colinds, ci) and np.array_equal(c.values, vs), step
        assert routes == ['general', 'general', 'dense-panel', 'dense-panel']
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    monkeypatch.setenv('CSRK_SPGEMM_DENSE', '0')
    assert run(A, full) == 'general'


@pytest.mark.parametrize('route', ['dense-panel', 'general'])
def test_mult_ab_overflow_is_an_error_before_any_numeric_pass(route, monkeypatch):
    """
    A product of more than 2^31 - 1 entries cannot have the reference's int32 row pointers (multiply.py:28): CSRK_ERR_OVERFLOW
    with the reason, from the count alone -- no numeric pass runs, nothing of that size is allocated -- on both routes
    (600 000 rows of one entry each times a fully populated 1 x 4096 B: 2.46e9 entries), and the handles stay usable.
    """
    from csr_amd import CSR
    from csr_amd._lib import CsrkError
    from csr_amd.kernels import hip as K
    if route == 'general':
        monkeypatch.setenv('CSRK_SPGEMM_DENSE', '0')
    n, k = 600_000, 4096
    A = CSR(n, 1, n, np.arange(n + 1, dtype=np.int32), np.zeros(n, dtype=np.int32), np.ones(n), _cast=False)
    B = _panel_csr(np.arange(k, dtype=np.float64).reshape(1, k))
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        with pytest.raises(CsrkError, match='int32 row pointers'):
            K.mult_ab(ah, bh)
        assert K.spgemm_last_route() == 'general'         # (nothing was taken: the call failed)
        assert np.array_equal(K.mult_vec(ah, np.array([2.0])), np.full(n, 2.0))
        small = K.to_handle(CSR(2, 1, 2, np.array([0, 1, 2], dtype=np.int32), np.zeros(2, dtype=np.int32), np.array([1.0, 3.0]), _cast=False))
        try:
            ch = K.mult_ab(small, bh)
            C = K.from_handle(ch)
            K.release_handle(ch)
        finally:
            K.release_handle(small)
        assert C.nnz == 2 * k and np.array_equal(C.values[k:], 3.0 * np.arange(k - 1, -1, -1.0))
    finally:
        K.release_handle(ah)
        K.release_handle(bh)


@pytest.mark.parametrize('k', [64, 7, 130])
def test_spmm_dense(k):
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(k)
    lens = np.minimum((rng.pareto(0.9, 8000) * 3).astype(np.int64), 9000)
    A = _rand(rng, 8000, 6000, lens)
    B = rng.uniform(-1, 1, (A.ncols, k))
    h = K.to_handle(A)
    try:
        Cm = K.mult_dense(h, B)
    finally:
        K.release_handle(h)
    ref = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, A.values, B)
    bound = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, np.abs(A.values), np.abs(B))
    assert Cm.shape == ref.shape
    assert np.all(np.abs(Cm - ref) <= 1e-6 * bound + 1e-300)
    assert np.all(np.abs(Cm - ref) <= 1e-12 * bound + 1e-300)


@pytest.mark.parametrize('k,dtype,groups', [(64, np.float64, 0), (6, np.float64, 2), (130, np.float32, 0), (200, None, 4)])
def test_spmm_dense_heavy_rows_in_registers(k, dtype, groups, monkeypatch):
    """
    The heavy-row form of the dense-panel SpMM (north_star's "dense B tile staged in LDS";
    spmm_hrows_kernel): B tiles through LDS, the rows' accumulators in dynamically indexed registers.  Forced on for a
    small matrix (CSRK_SPMM_HEAVY=1), with one, two and four row groups, panel widths that fill a 64-column launch,
    fall short of it and need several, f64 / f32 / absent values, columns sorted or not, a last tile that is cut by
    the matrix' width, and heavy rows that leave whole tiles empty.  Checked against the light-row kernel alone
    (CSRK_SPMM_HEAVY=0) and the oracle; run twice: the bits must repeat.
    """
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(1000 + k)
    n, ncols = 6000, 9001 + k
    lens = np.minimum((rng.pareto(1.0, n) * 4).astype(np.int64), 300)
    heavy = rng.choice(n, 1300, replace=False)
    lens[heavy] = rng.integers(256, 2500, size=len(heavy))
    lens[heavy[0]] = ncols                                   # a full row
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.empty(int(rp[-1]), np.int32)
    for i in range(n):
        m = int(lens[i])
        if m:
            lo = 0 if i % 3 or m > ncols - 2000 else 2000    # (some rows never touch the first tiles)
            c = lo + rng.choice(ncols - lo, m, replace=False)
            ci[rp[i]:rp[i + 1]] = np.sort(c) if i % 2 else c
    vals = None if dtype is None else rng.uniform(-1, 1, int(rp[-1])).astype(dtype)
    A = CSR(n, ncols, int(rp[-1]), rp, ci, vals, _cast=False)
    B = rng.uniform(-1, 1, (ncols, k))
    if groups:
        monkeypatch.setenv('CSRK_SPMM_HEAVY_GROUPS', str(groups))
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('CSRK_SPMM_HEAVY', mode)
        h = K.to_handle(A)
        try:
            outs[mode] = K.mult_dense(h, B)
            again = K.mult_dense(h, B)
        finally:
            K.release_handle(h)
            K.invalidate(A)                                  # (the next mode must build its own plan, not find this one cached)
        assert np.array_equal(outs[mode], again)
    v64 = np.ones(A.nnz) if vals is None else vals.astype(np.float64)
    ref = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, v64, B)
    bound = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, np.abs(v64), np.abs(B))
    for mode in ('0', '1'):
        assert np.all(np.abs(outs[mode] - ref) <= 1e-12 * bound + 1e-300), mode
    light = lens < 256                                       # rows the heavy form never takes: same kernel, same bits
    assert np.array_equal(outs['1'][light].view(np.int64), outs['0'][light].view(np.int64))
    assert not np.array_equal(outs['1'].view(np.int64), outs['0'].view(np.int64)) or k < 2      # (the heavy form did run)


# ---- COO ingest on the device ----------------------------------------------------------------------------

def test_from_coo_golden(golden):
    "csrk_from_coo against the reference's CSR.from_coo outputs (csr/csr.py:138-169), duplicates / unsorted / f4 / None: bit-exact"
    from csr_amd.kernels import hip as K
    g = golden('coo')
    for c in range(int(g['n'])):
        rows, cols = g[f'c{c}_rows'], g[f'c{c}_cols']
        vals = g[f'c{c}_vals'] if f'c{c}_vals' in g else None
        out = Mat(g, f'c{c}_out_')
        h = K.from_coo(rows, cols, vals, (out.nrows, out.ncols))
        try:
            m = K.from_handle(h)
        finally:
            K.release_handle(h)
        assert (m.nrows, m.ncols, m.nnz) == (out.nrows, out.ncols, out.nnz)
        assert m.rowptrs.dtype == out.rowptrs.dtype and np.array_equal(m.rowptrs, out.rowptrs), c
        assert np.array_equal(m.colinds, out.colinds), c
        if vals is None:
            assert m.values is None
        else:
            assert m.values.dtype == out.values.dtype and np.array_equal(m.values, out.values), c


def test_from_coo_device(golden):
    "csr/structure.py:11-67 at sizes the goldens do not reach: vs the oracle's restatement (orc_from_coo, pinned) and the KAT"
    from oracle import oracle as O
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    g = golden('kat')
    a = Mat(g, 'a_')
    h = K.from_coo(np.array([0, 0, 1, 3]), np.array([1, 2, 0, 1]), np.arange(4.0), (4, 3))
    try:
        c = K.from_handle(h)
    finally:
        K.release_handle(h)
    assert np.array_equal(c.rowptrs, a.rowptrs) and np.array_equal(c.colinds, a.colinds)
    assert np.array_equal(c.values, a.values)
    rng = np.random.default_rng(12)
    for nrows, ncols, nnz, dt in ((300, 200, 5000, np.float64), (70000, 1000, 200000, np.float32),
                                  (5, 5, 0, np.float64), (100000, 70000, 300000, None), (3, 2_000_000, 400000, np.float64)):
        rows = rng.integers(0, nrows, size=nnz).astype(np.int32)      # duplicates allowed, unsorted
        cols = rng.integers(0, ncols, size=nnz).astype(np.int32)
        vals = None if dt is None else rng.uniform(-1, 1, size=nnz).astype(dt)
        rp, ci, vs = O.from_coo(nrows, rows, cols, vals)
        host = CSR.from_coo(rows, cols, vals, (nrows, ncols))          # the package's own host ingest agrees too
        assert np.array_equal(host.rowptrs, rp) and np.array_equal(host.colinds, ci)
        h = K.from_coo(rows, cols, vals, (nrows, ncols))
        try:
            c = K.from_handle(h)
        finally:
            K.release_handle(h)
        assert (c.nrows, c.ncols, c.nnz) == (nrows, ncols, nnz)
        assert np.array_equal(c.rowptrs, rp) and np.array_equal(c.colinds, ci)
        if dt is None:
            assert c.values is None
        else:
            assert c.values.dtype == dt and np.array_equal(c.values, vs) and np.array_equal(host.values, vs)
    with pytest.raises(ValueError):
        K.from_coo(np.array([5]), np.array([0]), None, (3, 3))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_unit_center_long_rows_vs_oracle(dtype):
    "rows longer than 512 entries take the chunk kernels (one chunk: one pass; several: partials, join, scale); NaN, infinite and zero rows included"
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    lens = rng.integers(0, 20, size=400)
    lens[[5, 77, 200, 399]] = [9000, 30000, 8193, 12000]
    # rows of ONE 4096-entry chunk are finished by the first class-C kernel (513 .. 4096 entries), 4097 is two chunks
    lens[[9, 150, 151, 300, 301]] = [513, 2500, 4096, 4097, 600]
    m = _rand(rng, 400, 3000, lens, dtype=dtype)
    s300 = int(m.rowptrs[301])
    m.values[s300:s300 + 600] = 0                       # an all-zero one-chunk row
    m.values[int(m.rowptrs[150]) + 3] = np.inf          # an infinite value in a one-chunk row
    s200 = int(m.rowptrs[200])
    m.values[s200:s200 + 8193] = 0                      # an all-zero long row -> norm 0, NaN values
    m.values[int(m.rowptrs[399]) + 17] = np.nan         # NaN in a long row propagates
    rel = 1e-5 if dtype == np.float32 else 1e-9
    u, ur = m.copy(), m.values.copy()
    with np.errstate(all='ignore'):
        norms = u.normalize_rows('unit')
        rn = O.unit_rows(m.nrows, m.rowptrs, ur)
    assert norms == pytest.approx(rn, rel=rel, abs=0, nan_ok=True)
    assert np.array_equal(np.isnan(u.values), np.isnan(ur))
    assert u.values == pytest.approx(ur, rel=rel, abs=1e-300, nan_ok=True)
    m.values[int(m.rowptrs[399]) + 17] = 0.5
    m.values[int(m.rowptrs[150]) + 3] = 0.25
    c, cr = m.copy(), m.values.copy()
    means = c.normalize_rows('center')
    rm = O.center_rows(m.nrows, m.rowptrs, cr)
    tol = 1e-6 if dtype == np.float32 else 1e-12
    assert means == pytest.approx(rm, rel=rel, abs=tol)
    assert c.values == pytest.approx(cr, rel=rel, abs=tol)


@pytest.mark.parametrize('k', [64, 20])
def test_spmm_dense_heavy_rows_blocked(k, monkeypatch):
    "heavy rows in the register-accumulator form (forced on for this small matrix), split light rows beside them"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    monkeypatch.setenv('CSRK_SPMV_HEAVY_SPLIT', '1')
    monkeypatch.setenv('CSRK_SPMM_HEAVY', '1')
    rng = np.random.default_rng(40 + k)
    lens = rng.integers(0, 30, size=3000)
    lens[[0, 10, 1500, 2999]] = [5000, 2048, 9000, 3000]        # tier-0 rows
    lens[[20, 30]] = [1000, 300]                                # split light rows (> 256 entries)
    A = _rand(rng, 3000, 40000, lens, sort=True)
    B = rng.uniform(-1, 1, (A.ncols, k))
    h = K.to_handle(A)
    try:
        Cm = K.mult_dense(h, B)
        C2 = K.mult_dense(h, B)
    finally:
        K.release_handle(h)
    ref = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, A.values, B)
    bound = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, np.abs(A.values), np.abs(B))
    assert np.array_equal(Cm, C2)
    assert np.all(np.abs(Cm - ref) <= 1e-12 * bound + 1e-300)


@pytest.mark.parametrize('k,ldb,ldc', [(64, 64, 64), (64, 70, 67), (37, 41, 64), (130, 131, 140)])
def test_spmm_dense_strided_panels(k, ldb, ldc, monkeypatch):
    "csrk_spmm_dense_device with leading dimensions wider than the panel (B and C as column slices of wider arrays), heavy rows forced"
    import ctypes as C
    import torch
    from oracle import oracle as O
    from csr_amd._lib import lib, check, handle_t
    monkeypatch.setenv('CSRK_SPMM_HEAVY', '1')
    rng = np.random.default_rng(k * 1000 + ldb)
    nrows, ncols = 1500, 5000
    lens = rng.integers(0, 9, size=nrows)
    lens[rng.choice(nrows, 300, replace=False)] = rng.integers(256, 900, size=300)
    A = _rand(rng, nrows, ncols, lens, sort=True)
    Bw = rng.uniform(-1, 1, (ncols, ldb))
    dev = 'cuda'
    rp, ci, vs = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (A.rowptrs, A.colinds, A.values))
    dB = torch.from_numpy(Bw).to(dev)
    dC = torch.full((nrows, ldc), 7.0, dtype=torch.float64, device=dev)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, ncols, A.nnz, rp.data_ptr(), 0, ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
    try:
        check(lib.csrk_spmm_dense_device(h, dB.data_ptr(), k, ldb, dC.data_ptr(), ldc, None))
        torch.cuda.synchronize()
        st = (C.c_int64 * 9)()
        check(lib.csrk_spmm_plan_stats(h, st, 9))
        assert st[0] == 1 and st[2] >= 300                 # the heavy-row form is in use
    finally:
        check(lib.csrk_free(h))
    got = dC.cpu().numpy()
    B = np.ascontiguousarray(Bw[:, :k])
    ref = O.spmm_dense(nrows, A.rowptrs, A.colinds, A.values, B)
    bound = O.spmm_dense(nrows, A.rowptrs, A.colinds, np.abs(A.values), np.abs(B))
    assert np.all(np.abs(got[:, :k] - ref) <= 1e-12 * bound + 1e-300)
    assert np.all(got[:, k:] == 7.0)                       # nothing is written past the panel's width


def test_spmm_survives_spmv_algo_change(monkeypatch):
    "the SpMM plan is independent of the SpMV plan: changing the SpMV algorithm between products changes nothing"
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    monkeypatch.setenv('CSRK_SPMV_HEAVY_SPLIT', '1')
    monkeypatch.setenv('CSRK_SPMM_HEAVY', '1')
    rng = np.random.default_rng(4)
    lens = rng.integers(0, 10, size=1000)
    lens[[1, 500]] = [4000, 2500]
    A = _rand(rng, 1000, 30000, lens, sort=True)
    B = rng.uniform(-1, 1, (A.ncols, 8))
    ref = O.spmm_dense(A.nrows, A.rowptrs, A.colinds, A.values, B)
    h = K.to_handle(A)
    try:
        c1 = K.mult_dense(h, B)
        K.set_spmv_algo(h, 'vector')
        y = K.mult_vec(h, B[:, 0].copy())
        c2 = K.mult_dense(h, B)
        K.set_spmv_algo(h, 'auto')
        c3 = K.mult_dense(h, B)
    finally:
        K.release_handle(h)
    for c in (c1, c2, c3):
        assert np.allclose(c, ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(y, ref[:, 0], rtol=1e-10, atol=1e-12)


def test_spgemm_unsorted_b_rows():
    """
    The column strips need B's rows strictly ascending (sub-range bounds by binary search); sg_sorted_check sends a B
    with unordered rows -- legal for the reference, whose SMMP never looks at the order (multiply.py:60-129) -- to the
    other paths.  Same A, B as the nearly-full case of test_spgemm_deterministic, B's rows shuffled: the product must
    equal the oracle's on the shuffled B bit for bit (the sums' order is A's, untouched by the shuffle).
    """
    from oracle import oracle as O
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(7)
    la = rng.integers(0, 6, 200)
    la[::10] = 500
    lb = rng.integers(0, 40, 1500)
    lb[::7] = 5000
    def mk(nrows, ncols, lens, shuffle):
        rp = np.zeros(nrows + 1, dtype=np.int32)
        rp[1:] = np.cumsum(lens)
        rows = []
        for n in lens:
            c = np.sort(rng.choice(ncols, size=int(n), replace=False))
            if shuffle:
                rng.shuffle(c)
            rows.append(c)
        ci = np.concatenate(rows + [np.zeros(0, np.int64)]).astype(np.int32)
        return CSR(nrows, ncols, int(rp[-1]), rp, ci, rng.uniform(-1, 1, size=int(rp[-1])), _cast=False)
    A, B = mk(200, 1500, la, False), mk(1500, 20000, lb, True)
    ah, bh = K.to_handle(A), K.to_handle(B)
    try:
        ch = K.mult_ab(ah, bh)
        C = K.from_handle(ch)
        K.release_handle(ch)
    finally:
        K.release_handle(ah)
        K.release_handle(bh)
    nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), (B.nrows, B.ncols, B.rowptrs, B.colinds, B.values))
    rci, rvs = as_library_orders(crp, cci, cvs)
    assert np.array_equal(C.rowptrs, crp) and np.array_equal(C.colinds, rci)
    assert np.array_equal(C.values.view(np.int64), rvs.view(np.int64))


def test_row_ops_from_several_threads():
    """
    The kernels are nogil in the reference (csr/transform.py runs under @njit(nogil=True) callers) and ctypes releases the
    GIL: unit_rows / center_rows on DIFFERENT handles from several threads at once.  Inside the library the calls share
    one side stream, two events and the pinned words the list lengths come back through: every thread must get exactly what
    it gets alone.
    """
    import threading
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(77)
    mats = []
    for t in range(6):
        lens = rng.integers(0, 40, 30000 + 1000 * t)
        lens[::97] = 700 + 50 * t              # class B / C rows
        lens[5 + t] = 9000 + 4096 * t          # rows of several chunks
        mats.append(_rand(rng, len(lens), 5000, lens))

    def run(A, op):
        h = K.to_handle(A)
        try:
            stat = (K.unit_rows if op == 'unit' else K.center_rows)(h)
            return stat, K.values_of(h)
        finally:
            K.release_handle(h)
    ops = ['unit', 'center', 'unit', 'center', 'unit', 'center']
    alone = [run(A, op) for A, op in zip(mats, ops)]
    for _ in range(3):
        got = [None] * len(mats)
        errs = []

        def work(i):
            try:
                got[i] = run(mats[i], ops[i])
            except Exception as e:          # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(mats))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for (s0, v0), (s1, v1) in zip(alone, got):
            assert np.array_equal(s0, s1, equal_nan=True) and np.array_equal(v0, v1, equal_nan=True)
