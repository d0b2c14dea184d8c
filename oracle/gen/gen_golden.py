#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by running the REFERENCE's own code.

Container-only tool: it imports lenskit/csr from /root/reference in the reference's
NUMBA_DISABLE_JIT mode (csr/csr.py:20-43) with the stand-in `numba` package in
oracle/gen/numba_stub (real Numba is not installable in this image),
feeds it seeded inputs, and stores inputs + the reference's outputs.  The fixtures are
data only; neither this script nor the fixtures contain reference source.  Nothing on the
GPU box needs /root/reference: tests read the committed .npz files.

Run:  python oracle/gen/gen_golden.py        (writes tests/golden/)

Input distributions restate csr/test_utils.py:30-101 (`csrs`, `mm_pairs`): shapes 1..80
(mm: 1..100), density <= 0.5, unique COO coordinates, values in +-1e3 of dtype f4/f8 with
exact zeros dropped, or structure-only.
"""
import os
import sys

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CSR_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, 'numba_stub'))

import numpy as np  # noqa: E402

import csr  # noqa: E402
from csr import CSR  # noqa: E402
from csr.kernels import get_kernel  # noqa: E402

assert csr.__file__.startswith(REF), csr.__file__
K = get_kernel()
assert K.__name__ == 'csr.kernels.numba', K.__name__

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def draw_csr(rng, nrows=None, ncols=None, values=None, dtype=None, max_density=0.5, max_dim=80):
    "restates csr/test_utils.py:30-74"
    if nrows is None:
        nrows = int(rng.integers(1, max_dim + 1))
    if ncols is None:
        ncols = int(rng.integers(1, max_dim + 1))
    nnz_ub = int(np.ceil(nrows * ncols * max_density))
    nnz = int(rng.integers(0, nnz_ub + 1))
    coords = rng.choice(nrows * ncols, size=nnz, replace=False).astype(np.int32)
    rows = np.mod(coords, nrows).astype(np.int32)
    cols = np.floor_divide(coords, nrows).astype(np.int32)
    if dtype is None:
        dtype = rng.choice(['f4', 'f8'])
    dtype = np.dtype(dtype)
    if values is None:
        values = bool(rng.integers(0, 2))
    if values:
        vals = rng.uniform(-1.0e3, 1.0e3, size=nnz).astype(dtype)
        # sprinkle exact zeros (they must be dropped) and tiny magnitudes
        if nnz > 3:
            vals[rng.integers(0, nnz)] = 0.0
            vals[rng.integers(0, nnz)] *= dtype.type(1e-30)
        nz = vals != 0.0
        rows, cols, vals = rows[nz], cols[nz], vals[nz]
    else:
        vals = None
    return CSR.from_coo(rows, cols, vals, (nrows, ncols))


def put(d, prefix, m):
    d[prefix + 'shape'] = np.array([m.nrows, m.ncols, m.nnz], dtype=np.int64)
    d[prefix + 'rowptrs'] = np.asarray(m.rowptrs).copy()
    d[prefix + 'colinds'] = np.asarray(m.colinds).copy()
    if m.values is not None:
        d[prefix + 'values'] = np.asarray(m.values).copy()


def gen_kat():
    "fixed known-answer cases from the reference tests"
    d = {}
    rows = np.array([0, 0, 1, 3], dtype=np.int32)
    cols = np.array([1, 2, 0, 1], dtype=np.int32)
    vals = np.arange(4, dtype=np.float64)
    m = CSR.from_coo(rows, cols, vals)          # tests/test_transpose.py:11-27
    put(d, 'a_', m)
    put(d, 'at_', m.transpose())
    put(d, 'ats_', m.transpose(False))
    d['a_extents'] = np.array([m.row_extent(i) for i in range(m.nrows)], dtype=np.int64)  # test_attributes.py:36-45
    d['a_row_nnzs'] = m.row_nnzs()
    d['a_mv_ones'] = m.mult_vec(np.ones(m.ncols))
    np.savez_compressed(os.path.join(OUT, 'kat.npz'), **d)


def gen_spmv(n=60):
    rng = np.random.default_rng(20261003)
    d = {'n': np.array(n)}
    for c in range(n):
        m = draw_csr(rng)
        x = rng.uniform(-1.0e3, 1.0e3, size=m.ncols)
        if c % 7 == 0:
            x = x.astype(np.float32)
        put(d, f'c{c}_', m)
        d[f'c{c}_x'] = x
        with np.errstate(all='ignore'):
            d[f'c{c}_y'] = m.mult_vec(x)
        # sharded path (csr/csr.py:584-590) with a tiny max_nnz, when every row fits
        lim = 40
        if m.nnz > lim and int(np.max(m.row_nnzs())) <= lim:
            shards = m._shard_rows(lim)
            d[f'c{c}_shard_rows'] = np.array([s.nrows for s in shards], dtype=np.int64)
            d[f'c{c}_y_sharded'] = np.concatenate([K.mult_vec(K.to_handle(s), x) for s in shards])
    np.savez_compressed(os.path.join(OUT, 'spmv.npz'), **d)


def gen_cfg1():
    "BASELINE.json configs[0]: 10k x 10k, nnz = 1e5, fp64 (SURVEY.md section 8d row 1)"
    rng = np.random.default_rng(20261003)
    n, nnz = 10000, 100000
    coords = rng.choice(n * n, size=nnz, replace=False)
    coords.sort()
    rows = (coords // n).astype(np.int32)
    cols = (coords % n).astype(np.int32)
    vals = rng.standard_normal(nnz)
    x = rng.standard_normal(n)
    m = CSR.from_coo(rows, cols, vals, (n, n))
    d = {}
    put(d, 'a_', m)
    d['x'] = x
    d['y'] = m.mult_vec(x)
    np.savez_compressed(os.path.join(OUT, 'cfg1_spmv.npz'), **d)


def gen_transpose(n=40):
    rng = np.random.default_rng(77001)
    d = {'n': np.array(n)}
    for c in range(n):
        m = draw_csr(rng)
        if c >= n - 6:
            # duplicate (i, j) entries and unsorted rows: build the struct directly so the
            # stable-scatter order (structure.py:191-197) is observable
            nr, nc = int(rng.integers(2, 20)), int(rng.integers(2, 12))
            lens = rng.integers(0, 9, size=nr)
            rp = np.zeros(nr + 1, dtype=np.int32)
            rp[1:] = np.cumsum(lens)
            ci = rng.integers(0, nc, size=int(rp[-1])).astype(np.int32)   # dups, unsorted
            vs = rng.uniform(-5, 5, size=int(rp[-1]))
            m = CSR(nr, nc, int(rp[-1]), rp, ci, vs)
        put(d, f'c{c}_', m)
        t = m.transpose()
        put(d, f'c{c}_t_', t)
        ts = m.transpose(False)
        put(d, f'c{c}_ts_', ts)
        assert ts.values is None
        d[f'c{c}_row_nnzs'] = m.row_nnzs()
    np.savez_compressed(os.path.join(OUT, 'transpose.npz'), **d)


def gen_rows(n=40):
    "unit_rows / center_rows (csr/transform.py)"
    rng = np.random.default_rng(424242)
    d = {'n': np.array(n)}
    for c in range(n):
        m = draw_csr(rng, values=True)
        if c == n - 1:      # all-zero row, subnormal-scale row, huge row, empty row (f8)
            rp = np.array([0, 3, 3, 6, 9, 10], dtype=np.int32)
            ci = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 1], dtype=np.int32)
            vs = np.array([0.0, 0.0, 0.0, 1e-200, -3e-200, 2e-200, 1e300, -1e300, 5e299, 5e-324])
            m = CSR(5, 3, 10, rp, ci, vs)
        if c == n - 2:      # same idea in f4
            rp = np.array([0, 2, 2, 5, 7], dtype=np.int32)
            ci = np.array([0, 1, 0, 1, 2, 0, 2], dtype=np.int32)
            vs = np.array([0.0, 0.0, 1e-30, -3e-30, 2e-30, 3e38, -1e38], dtype=np.float32)
            m = CSR(4, 3, 7, rp, ci, vs)
        put(d, f'c{c}_', m)
        u = m.copy()
        with np.errstate(all='ignore'):
            d[f'c{c}_unit_norms'] = u.normalize_rows('unit')
        d[f'c{c}_unit_values'] = np.asarray(u.values).copy()
        z = m.copy()
        with np.errstate(all='ignore'):
            d[f'c{c}_center_means'] = z.normalize_rows('center')
        d[f'c{c}_center_values'] = np.asarray(z.values).copy()
    np.savez_compressed(os.path.join(OUT, 'rows.npz'), **d)


def gen_spgemm(n=24):
    "mult_ab / mult_abt (csr/kernels/numba/multiply.py) and CSR.multiply (csr/csr.py:524-567)"
    rng = np.random.default_rng(9001)
    d = {'n': np.array(n)}
    for c in range(n):
        r, mid, k = (int(rng.integers(1, 101)) for _ in range(3))
        dt = rng.choice(['f4', 'f8'])
        A = draw_csr(rng, r, mid, values=True, dtype=dt)
        B = draw_csr(rng, mid, k, values=True, dtype=dt)
        put(d, f'c{c}_a_', A)
        put(d, f'c{c}_b_', B)
        raw = K.mult_ab(K.to_handle(A), K.to_handle(B))      # explicit zeros kept, reference order
        put(d, f'c{c}_raw_', raw)
        put(d, f'c{c}_ab_', A.multiply(B))                   # after _filter_zeros
        Bt = B.transpose()                                    # k x mid
        put(d, f'c{c}_bt_', Bt)
        put(d, f'c{c}_abt_', A.multiply(Bt, transpose=True))
        # order_columns / sort_rows on the raw product (unsorted rows)
        s = CSR(raw.nrows, raw.ncols, raw.nnz, raw.rowptrs.copy(), raw.colinds.copy(), raw.values.copy())
        s.sort_rows()
        put(d, f'c{c}_rawsorted_', s)
    np.savez_compressed(os.path.join(OUT, 'spgemm.npz'), **d)


def gen_shard(n=20):
    "_shard_rows / _assemble_shards (csr/csr.py:599-650; tests/test_transform.py:172-197)"
    rng = np.random.default_rng(5150)
    d = {'n': np.array(n)}
    for c in range(n):
        m = draw_csr(rng, int(rng.integers(10, 101)), int(rng.integers(10, 101)), values=True, max_dim=100)
        put(d, f'c{c}_', m)
        shards = m._shard_rows(500)
        d[f'c{c}_shard_rows'] = np.array([s.nrows for s in shards], dtype=np.int64)
        d[f'c{c}_shard_nnz'] = np.array([s.nnz for s in shards], dtype=np.int64)
        back = CSR._assemble_shards(shards)
        d[f'c{c}_assembled_rowptrs'] = np.asarray(back.rowptrs).copy()
    # the error case: a row larger than the target
    rp = np.array([0, 600, 700], dtype=np.int32)
    m = CSR(2, 1000, 700, rp, np.arange(700, dtype=np.int32) % 1000, np.ones(700))
    try:
        m._shard_rows(500)
        d['big_row_error'] = np.array(0)
    except ValueError:
        d['big_row_error'] = np.array(1)
    np.savez_compressed(os.path.join(OUT, 'shard.npz'), **d)


def gen_pick(n=40):
    "pick_rows (csr/csr.py:347-364, csr/structure.py:84-149); draws as tests/test_transform.py:38-62"
    rng = np.random.default_rng(20261003)
    d = {'n': np.array(n)}
    for c in range(n):
        m = draw_csr(rng)
        include = bool(rng.integers(0, 2))
        k = int(rng.integers(0, m.nrows * 10 + 1))
        rows = rng.integers(0, m.nrows, size=k).astype(np.int32)
        if c == 0:
            rows = np.zeros(0, dtype=np.int32)          # nothing picked
        put(d, f'c{c}_', m)
        d[f'c{c}_rows'] = rows
        d[f'c{c}_include'] = np.array(include)
        sub = m.pick_rows(rows, include_values=include)
        assert sub.nrows == len(rows)
        put(d, f'c{c}_out_', sub)
    np.savez_compressed(os.path.join(OUT, 'pick.npz'), **d)


def gen_coo(n=36):
    """
    CSR.from_coo (csr/csr.py:138-169 -> csr/structure.py:11-67): the COO INPUTS and the reference's CSR.  Unlike
    draw_csr's unique coordinates these draws repeat (i, j) pairs and come in arbitrary order, so the stable
    counting sort's order (entries of a row keep their input order, structure.py:24-31 / :49-56) is observable.
    """
    rng = np.random.default_rng(60061)
    d = {'n': np.array(n)}
    for c in range(n):
        nrows, ncols = int(rng.integers(1, 81)), int(rng.integers(1, 81))
        nnz = int(rng.integers(0, 4 * max(nrows, ncols) + 1))
        if c == 0:
            nnz = 0                                           # nothing at all
        if c == 1:
            nrows, nnz = 1, 17                                # every entry in one row
        rows = rng.integers(0, nrows, size=nnz).astype(np.int32)          # duplicates, unsorted
        cols = rng.integers(0, ncols, size=nnz).astype(np.int32)
        if c % 5 == 2 and nnz:                                # sorted by row already, columns descending
            o = np.lexsort((-cols, rows))
            rows, cols = rows[o], cols[o]
        kind = c % 3                                          # f8 / f4 / structure only
        vals = None if kind == 2 else rng.uniform(-1.0e3, 1.0e3, size=nnz).astype('f8' if kind == 0 else 'f4')
        shape = None if (c % 7 == 3 and nnz) else (nrows, ncols)         # inferred shape (csr.py:160-161)
        m = CSR.from_coo(rows, cols, vals, shape)
        d[f'c{c}_rows'], d[f'c{c}_cols'] = rows, cols
        if vals is not None:
            d[f'c{c}_vals'] = vals
        d[f'c{c}_shape_given'] = np.array(shape is not None)
        put(d, f'c{c}_out_', m)
    np.savez_compressed(os.path.join(OUT, 'coo.npz'), **d)


def gen_spmm_dense(n=12):
    """
    The dense-panel product of BASELINE.json configs[2] through the reference's own entry point: B [ncols x k] handed to
    K.mult_ab as a fully populated CSR (csr/kernels/numba/multiply.py:13-38; numeric recurrence :110-122), the raw
    product (explicit zeros kept) densified AND as the reference returns it (raw rowptrs / colinds / values: k entries per
    row of C whose row of A holds an entry, columns in reverse order of first discovery, multiply.py:79-82, 94-97).
    k in {1, 7, 64}.  Cases n and n + 1 (appended: the first n keep their inputs): the rows of B in a shuffled column
    order -- a dense B that is not the row-major panel -- and an A with whole blocks of empty rows.
    """
    rng = np.random.default_rng(64064)
    d = {'n': np.array(n + 2)}
    for c in range(n + 2):
        k = (1, 7, 64)[c % 3] if c < n else (7, 64)[c - n]
        A = draw_csr(rng, values=True, dtype='f4' if c % 4 == 3 else 'f8')
        if c == n + 1:                                        # rows 3 .. 3 + a third of them emptied
            keep = np.ones(A.nrows, dtype=bool)
            keep[3:3 + A.nrows // 3] = False
            rows = np.repeat(np.arange(A.nrows), np.diff(A.rowptrs))
            m = keep[rows]
            A = CSR.from_coo(rows[m].astype(np.int32), A.colinds[m].copy(), A.values[m].copy(), (A.nrows, A.ncols))
        B = rng.uniform(-1.0, 1.0, size=(A.ncols, k))
        if c == 5:
            B[rng.integers(0, A.ncols), :] = 0.0              # a zero row of B: explicit zeros in the product
        cols = np.tile(np.arange(k, dtype=np.int32), A.ncols)
        vals = B.reshape(-1).copy()
        if c == n:                                            # every row of B in its own column order
            for j in range(A.ncols):
                o = rng.permutation(k)
                cols[j * k:(j + 1) * k] = o
                vals[j * k:(j + 1) * k] = B[j, o]
        Bc = CSR(A.ncols, k, A.ncols * k, np.arange(A.ncols + 1, dtype=np.int32) * k, cols, vals)
        raw = K.mult_ab(K.to_handle(A), K.to_handle(Bc))
        Cd = np.zeros((A.nrows, k))
        rp, ci, vs = np.asarray(raw.rowptrs), np.asarray(raw.colinds), np.asarray(raw.values)
        for i in range(A.nrows):
            Cd[i, ci[rp[i]:rp[i + 1]]] = vs[rp[i]:rp[i + 1]]
        put(d, f'c{c}_a_', A)
        d[f'c{c}_B'] = B
        d[f'c{c}_b_colinds'] = cols
        d[f'c{c}_b_values'] = vals
        d[f'c{c}_C'] = Cd
        d[f'c{c}_raw_nnz'] = np.array(raw.nnz)
        d[f'c{c}_raw_rowptrs'], d[f'c{c}_raw_colinds'], d[f'c{c}_raw_values'] = rp.copy(), ci.copy(), vs.copy()
    np.savez_compressed(os.path.join(OUT, 'spmm_dense.npz'), **d)


if __name__ == '__main__':
    gen_kat()
    gen_spmv()
    gen_cfg1()
    gen_transpose()
    gen_rows()
    gen_spgemm()
    gen_shard()
    gen_pick()
    gen_coo()
    gen_spmm_dense()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
