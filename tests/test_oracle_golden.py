"""
Pins the CPU oracle (oracle/csr_oracle.c) to the reference: every function is checked
against golden vectors captured from the reference's own code (oracle/gen/gen_golden.py)
and against the reference tests' fixed known-answer cases.  CPU only.
"""
import numpy as np
import pytest

from conftest import Mat
from oracle import oracle as O


def test_kat_transpose_and_extents(golden):
    "tests/test_transpose.py:11-27 and tests/test_attributes.py:36-45 of the reference"
    g = golden('kat')
    a = Mat(g, 'a_')
    assert list(a.rowptrs) == [0, 2, 3, 3, 4]
    nr, nc, brp, bci, bvs = O.transpose(a.nrows, a.ncols, a.rowptrs, a.colinds, a.values)
    assert (nr, nc) == (3, 4)
    assert list(brp) == [0, 1, 3, 4]
    at = Mat(g, 'at_')
    assert np.array_equal(brp, at.rowptrs) and np.array_equal(bci, at.colinds)
    assert np.array_equal(bvs, at.values)
    _, _, srp, sci, svs = O.transpose(a.nrows, a.ncols, a.rowptrs, a.colinds, a.values, False)
    ats = Mat(g, 'ats_')
    assert svs is None and ats.values is None
    assert np.array_equal(srp, ats.rowptrs) and np.array_equal(sci, ats.colinds)
    ext = [tuple(int(v) for v in O.row_extent(a.rowptrs, i)) for i in range(a.nrows)]
    assert ext == [(0, 2), (2, 3), (3, 3), (3, 4)]
    assert np.array_equal(np.array(ext), g['a_extents'])
    assert np.array_equal(O.row_nnzs(a.rowptrs), g['a_row_nnzs'])
    assert np.array_equal(O.mult_vec(a.nrows, a.ncols, a.rowptrs, a.colinds, a.values, np.ones(3)),
                          g['a_mv_ones'])


def test_mult_vec_golden(golden):
    g = golden('spmv')
    n = int(g['n'])
    seen_none = seen_f4 = 0
    for c in range(n):
        m = Mat(g, f'c{c}_')
        x = g[f'c{c}_x']
        y = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
        ref = g[f'c{c}_y']
        assert y.dtype == np.float64 and y.shape == (m.nrows,)
        seen_none += m.values is None
        seen_f4 += m.values is not None and m.values.dtype == np.float32
        # same order, same precision (f4*f4 products rounded to f4 like NumPy): bit-identical
        assert np.array_equal(y, ref), c
    assert seen_none > 3 and seen_f4 > 3


def test_mult_vec_sharded_golden(golden):
    "csr/csr.py:584-590 with max_nnz = 40"
    g = golden('spmv')
    hits = 0
    for c in range(int(g['n'])):
        if f'c{c}_shard_rows' not in g:
            continue
        hits += 1
        m = Mat(g, f'c{c}_')
        splits = O.shard_splits(m.rowptrs, 40)
        assert [e - b for b, e in splits] == list(g[f'c{c}_shard_rows'])
        parts = []
        for b, e in splits:
            s, t = int(m.rowptrs[b]), int(m.rowptrs[e])
            rp = m.rowptrs[b:e + 1] - m.rowptrs[b]
            vs = None if m.values is None else m.values[s:t]
            parts.append(O.mult_vec(e - b, m.ncols, rp, m.colinds[s:t], vs, g[f'c{c}_x']))
        ys = np.concatenate(parts)
        assert ys == pytest.approx(g[f'c{c}_y_sharded'], rel=1e-6, abs=1e-30)
    assert hits > 5


def test_cfg1_spmv(golden):
    "BASELINE.json configs[0]: 10k x 10k, nnz=1e5, fp64"
    g = golden('cfg1_spmv')
    a = Mat(g, 'a_')
    assert (a.nrows, a.ncols, a.nnz) == (10000, 10000, 100000)
    y = O.mult_vec(a.nrows, a.ncols, a.rowptrs, a.colinds, a.values, g['x'])
    assert np.array_equal(y, g['y'])


def test_transpose_golden(golden):
    g = golden('transpose')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        t = Mat(g, f'c{c}_t_')
        nr, nc, brp, bci, bvs = O.transpose(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values)
        assert (nr, nc) == (t.nrows, t.ncols)
        assert np.array_equal(brp, t.rowptrs)
        assert np.array_equal(bci, t.colinds)
        if m.values is None:
            assert bvs is None and t.values is None
        else:
            assert bvs.dtype == np.float64 and t.values.dtype == np.float64
            assert np.array_equal(bvs, t.values)
        ts = Mat(g, f'c{c}_ts_')
        _, _, srp, sci, svs = O.transpose(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, False)
        assert svs is None
        assert np.array_equal(srp, ts.rowptrs) and np.array_equal(sci, ts.colinds)
        assert np.array_equal(O.row_nnzs(m.rowptrs), g[f'c{c}_row_nnzs'])


def test_unit_center_golden(golden):
    g = golden('rows')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        f4 = m.values.dtype == np.float32
        rel = 2e-6 if f4 else 1e-12
        vs = m.values.copy()
        with np.errstate(all='ignore'):
            norms = O.unit_rows(m.nrows, m.rowptrs, vs)
        assert norms.dtype == m.values.dtype
        assert norms == pytest.approx(g[f'c{c}_unit_norms'], rel=rel, abs=0, nan_ok=True)
        assert vs == pytest.approx(g[f'c{c}_unit_values'], rel=rel, abs=1e-300, nan_ok=True)
        assert np.array_equal(np.isnan(vs), np.isnan(g[f'c{c}_unit_values']))
        vs = m.values.copy()
        means = O.center_rows(m.nrows, m.rowptrs, vs)
        scale = float(np.max(np.abs(m.values))) if m.nnz else 1.0
        assert means == pytest.approx(g[f'c{c}_center_means'], rel=rel, abs=scale * (1e-6 if f4 else 1e-13))
        assert vs == pytest.approx(g[f'c{c}_center_values'], rel=rel, abs=scale * (1e-6 if f4 else 1e-13))


def _dense_of(nr, nc, rp, ci, vs):
    out = np.zeros((nr, nc))
    for i in range(nr):
        for p in range(int(rp[i]), int(rp[i + 1])):
            out[i, ci[p]] += vs[p]
    return out


def test_spgemm_golden(golden):
    g = golden('spgemm')
    n_f8 = 0
    for c in range(int(g['n'])):
        A, B = Mat(g, f'c{c}_a_'), Mat(g, f'c{c}_b_')
        raw = Mat(g, f'c{c}_raw_')
        f8 = A.values.dtype == np.float64
        n_f8 += f8
        nr, nc, crp, cci, cvs = O.mult_ab(A.tup(), B.tup())
        assert (nr, nc) == (raw.nrows, raw.ncols)
        assert crp.dtype == np.int32 and np.array_equal(crp, raw.rowptrs)
        assert np.array_equal(cci, raw.colinds)            # reference column order reproduced
        srp, sci = O.sym_mm(A.tup(), B.tup())
        assert np.array_equal(srp, raw.rowptrs) and np.array_equal(sci, raw.colinds)
        # same accumulation order and the same product precision (f4 * f4 rounded to f4, multiply.py:120): bit-identical
        assert np.array_equal(cvs, raw.values)
        # CSR.multiply = mult_ab + _filter_zeros (csr/csr.py:555)
        ab = Mat(g, f'c{c}_ab_')
        frp, fci, fvs = O.filter_zeros(nr, crp, cci, cvs)
        assert np.all(fvs != 0)
        assert np.array_equal(frp, ab.rowptrs) and np.array_equal(fci, ab.colinds)
        assert np.array_equal(fvs, ab.values)
        # mult_abt = mult_ab(A, transpose(Bt)) (multiply.py:41-57)
        Bt = Mat(g, f'c{c}_bt_')
        abt = Mat(g, f'c{c}_abt_')
        tnr, tnc, trp, tci, tvs = O.transpose(Bt.nrows, Bt.ncols, Bt.rowptrs, Bt.colinds, Bt.values)
        _, _, rp2, ci2, vs2 = O.mult_ab(A.tup(), (tnr, tnc, trp, tci, tvs))
        rp2, ci2, vs2 = O.filter_zeros(nr, rp2, ci2, vs2)
        # transpose() always yields f8 values (structure.py:177), so abt is f8 arithmetic
        assert np.array_equal(rp2, abt.rowptrs) and np.array_equal(ci2, abt.colinds)
        assert np.array_equal(vs2, abt.values)
        # order_columns on the raw product
        srt = Mat(g, f'c{c}_rawsorted_')
        sci2, svs2 = O.sort_rows(raw.nrows, raw.rowptrs, raw.colinds, raw.values)
        assert np.array_equal(sci2, srt.colinds) and np.array_equal(svs2, srt.values)
    assert n_f8 > 3


def test_shard_golden(golden):
    "tests/test_transform.py:172-197 of the reference"
    g = golden('shard')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        splits = O.shard_splits(m.rowptrs, 500)
        assert [e - b for b, e in splits] == list(g[f'c{c}_shard_rows'])
        assert [int(m.rowptrs[e] - m.rowptrs[b]) for b, e in splits] == list(g[f'c{c}_shard_nnz'])
        assert all(int(m.rowptrs[e] - m.rowptrs[b]) <= 500 for b, e in splits)
        assert np.array_equal(g[f'c{c}_assembled_rowptrs'], m.rowptrs)
    assert int(g['big_row_error']) == 1
    with pytest.raises(ValueError):
        O.shard_splits(np.array([0, 600, 700]), 500)


def test_spmm_dense_golden(golden):
    """
    orc_spmm_dense is pinned to the reference itself: K.mult_ab(A, CSR(B)) with B a fully populated CSR
    (csr/kernels/numba/multiply.py:13-38, numeric recurrence :110-122), densified by oracle/gen/gen_golden.py.
    Same additions in the same order (work[k] += a * b over A's row in storage order): bit for bit.
    """
    g = golden('spmm_dense')
    ks, f4 = set(), 0
    for c in range(int(g['n'])):
        a = Mat(g, f'c{c}_a_')
        B, Cref = g[f'c{c}_B'], g[f'c{c}_C']
        assert B.shape[0] == a.ncols and Cref.shape == (a.nrows, B.shape[1])
        Cm = O.spmm_dense(a.nrows, a.rowptrs, a.colinds, a.values, B)
        assert np.array_equal(Cm, Cref), c
        # the raw product keeps explicit zeros: one stored entry per (non-empty row of A, column of B)
        assert int(g[f'c{c}_raw_nnz']) == int(np.count_nonzero(np.diff(a.rowptrs))) * B.shape[1]
        ks.add(B.shape[1])
        f4 += a.values.dtype == np.float32
    assert ks == {1, 7, 64} and f4 >= 2


def test_mult_ab_with_a_dense_b_golden(golden):
    """
    The same products as the reference returns them -- raw rowptrs / colinds / values of K.mult_ab(A, CSR(B)) -- pin
    orc_mult_ab on fully populated B's: k entries per row of C whose row of A holds an entry, none otherwise, columns in
    reverse order of first discovery (for row-major B rows: k - 1 .. 0), explicit zeros kept (multiply.py:79-82, 94-97).
    Case 12 has every row of B in a shuffled column order, case 13 blocks of empty rows in A.
    """
    g = golden('spmm_dense')
    shuffled = gaps = 0
    for c in range(int(g['n'])):
        a = Mat(g, f'c{c}_a_')
        k = g[f'c{c}_B'].shape[1]
        brp = np.arange(a.ncols + 1, dtype=np.int32) * k
        _, _, rp, ci, vs = O.mult_ab((a.nrows, a.ncols, a.rowptrs, a.colinds, a.values),
                                     (a.ncols, k, brp, g[f'c{c}_b_colinds'], g[f'c{c}_b_values']))
        assert np.array_equal(rp, g[f'c{c}_raw_rowptrs']) and np.array_equal(ci, g[f'c{c}_raw_colinds']), c
        assert np.array_equal(vs, g[f'c{c}_raw_values']), c
        rowmajor = np.array_equal(g[f'c{c}_b_colinds'], np.tile(np.arange(k, dtype=np.int32), a.ncols))
        shuffled += not rowmajor
        live = np.diff(a.rowptrs) > 0
        gaps += int((~live).sum() > a.nrows // 4)
        assert np.array_equal(np.diff(rp), live * k)
        if rowmajor:
            assert np.array_equal(ci, np.tile(np.arange(k - 1, -1, -1, dtype=np.int32), int(live.sum())))
    assert shuffled == 1 and gaps >= 1


def test_spmm_dense_matches_dense_product():
    rng = np.random.default_rng(3)
    nr, nc, k = 37, 23, 5
    dense = rng.uniform(-1, 1, (nr, nc)) * (rng.uniform(size=(nr, nc)) < 0.2)
    rp = np.zeros(nr + 1, dtype=np.int32)
    ci, vs = [], []
    for i in range(nr):
        nz = np.nonzero(dense[i])[0]
        ci.extend(nz)
        vs.extend(dense[i, nz])
        rp[i + 1] = len(ci)
    B = rng.uniform(-1, 1, (nc, k))
    Cm = O.spmm_dense(nr, rp, np.array(ci, dtype=np.int32), np.array(vs), B)
    assert Cm == pytest.approx(dense @ B, rel=1e-12, abs=1e-14)


def test_from_coo_golden(golden):
    """
    orc_from_coo against the reference's CSR.from_coo (csr/csr.py:138-169 -> csr/structure.py:11-67) on COO inputs
    with repeated coordinates in arbitrary order: row pointers, column order inside rows (input order kept) and the
    values' bits and dtype.
    """
    g = golden('coo')
    kinds, dups, inferred, empty = set(), 0, 0, 0
    for c in range(int(g['n'])):
        rows, cols = g[f'c{c}_rows'], g[f'c{c}_cols']
        vals = g[f'c{c}_vals'] if f'c{c}_vals' in g else None
        out = Mat(g, f'c{c}_out_')
        assert out.nnz == len(rows)
        if not bool(g[f'c{c}_shape_given']):
            assert out.nrows == int(rows.max()) + 1 and out.ncols == int(cols.max()) + 1      # csr/csr.py:160-161
            inferred += 1
        rp, ci, vs = O.from_coo(out.nrows, rows, cols, vals)
        assert rp.dtype == out.rowptrs.dtype and np.array_equal(rp, out.rowptrs), c
        assert np.array_equal(ci, out.colinds), c
        if vals is None:
            assert vs is None and out.values is None
        else:
            assert vs.dtype == out.values.dtype == vals.dtype and np.array_equal(vs, out.values), c
        kinds.add(None if vals is None else vals.dtype.str)
        dups += len(set(zip(rows.tolist(), cols.tolist()))) < len(rows)
        empty += len(rows) == 0
    assert kinds == {None, '<f4', '<f8'} and dups > 10 and inferred >= 3 and empty >= 1


def test_pick_rows_golden(golden):
    "csr/csr.py:347-364 (tests/test_transform.py:38-62): picked rows in order, repeats allowed, values optional"
    g = golden('pick')
    seen_novals = seen_f4 = seen_repeat = 0
    for c in range(int(g['n'])):
        m, out = Mat(g, f'c{c}_'), Mat(g, f'c{c}_out_')
        rows, include = g[f'c{c}_rows'], bool(g[f'c{c}_include'])
        rp, ci, vs = O.pick_rows(m.rowptrs, m.colinds, m.values, rows, include)
        assert out.nrows == len(rows) and out.ncols == m.ncols
        assert rp.dtype == np.int32 and np.array_equal(rp, out.rowptrs) and np.array_equal(ci, out.colinds)
        if include and m.values is not None:
            assert vs.dtype == m.values.dtype and np.array_equal(vs, out.values)
            seen_f4 += vs.dtype == np.float32
        else:
            assert vs is None and out.values is None
            seen_novals += 1
        seen_repeat += len(set(rows.tolist())) < len(rows)
    assert seen_novals > 3 and seen_f4 > 1 and seen_repeat > 3


def test_row_parallel_variant_is_bit_identical_to_the_sequential_port():
    "bench.py's second CPU figure (OpenMP over rows; NOT the reference's kernel) computes the port's bits"
    from csr_amd import synth
    from oracle import oracle as O
    m = synth.powerlaw_csr(30000, 20000, 400000, device='cpu')
    rp, ci, vs = m['rowptrs'].numpy(), m['colinds'].numpy(), m['values'].numpy()
    x = synth.dense_vector(20000, device='cpu').numpy()
    ref = O.mult_vec(30000, 20000, rp, ci, vs, x)
    for nthr in (1, 3, 8):
        assert np.array_equal(O.mult_vec_rows_parallel(30000, 20000, rp, ci, vs, x, nthr), ref)
    assert np.array_equal(O.mult_vec_rows_parallel(30000, 20000, rp, ci, None, x, 4),
                          O.mult_vec(30000, 20000, rp, ci, None, x))
