"""
ctypes/numpy front end of the CPU parity oracle (oracle/csr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (csr_amd) must never import this module.

Each function takes/returns plain numpy arrays laid out like the reference's CSR struct
(csr/csr.py:79-100): rowptrs[nrows+1], colinds[nnz] int32, values[nnz] or None.
Index arithmetic that the reference does in numpy on the host (row_nnzs, row_extent,
_shard_rows, _assemble_shards) is restated here in numpy, citing the reference lines.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)


def build(force=False):
    "Compile liboracle.so with gcc (idempotent)."
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'csr_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_sym_mm.restype = C.c_int64
        _LIB.orc_mult_ab.restype = C.c_int64
        _LIB.orc_filter_zeros.restype = C.c_int64
        _LIB.orc_pick_rows.restype = C.c_int64
        _LIB.orc_free.argtypes = [C.c_void_p]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _rp64(rowptrs):
    return np.ascontiguousarray(rowptrs, dtype=np.int64)


def _ci(colinds):
    return np.ascontiguousarray(colinds, dtype=np.int32)


def mult_vec_rows_parallel(nrows, ncols, rowptrs, colinds, values, x, nthreads):
    """
    NOT the reference (its kernel is single-threaded): the same per-row loop with the rows shared out over
    `nthreads` OpenMP threads; int32 row pointers and float64 (or no) values only.  bench.py's second CPU figure.
    """
    assert rowptrs.dtype == np.int32 and rowptrs.flags.c_contiguous and x.shape == (ncols,)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(nrows, dtype=np.float64)
    vs = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    L = lib()
    L.orc_mult_vec_i32_rows_omp(C.c_int32(nrows), _p(rowptrs, _i32p), _p(_ci(colinds), _i32p),
                                None if vs is None else _p(vs, _f64p), _p(x, _f64p), _p(y, _f64p), C.c_int32(int(nthreads)))
    return y


def mult_vec(nrows, ncols, rowptrs, colinds, values, x):
    "csr/kernels/numba/__init__.py:55-67"
    x = np.asarray(x)
    assert x.shape == (ncols,)
    nnz = int(rowptrs[nrows])
    y = np.empty(nrows, dtype=np.float64)
    ci = _ci(colinds)
    L = lib()
    if values is not None and values.dtype == np.float32 and x.dtype == np.float32:
        rp = _rp64(rowptrs)
        vs = np.ascontiguousarray(values)
        xf = np.ascontiguousarray(x)
        L.orc_mult_vec_f32f32(C.c_int32(nrows), C.c_int64(nnz), _p(rp, _i64p), _p(ci, _i32p),
                              _p(vs, _f32p), _p(xf, _f32p), _p(y, _f64p))
        return y
    x = np.ascontiguousarray(x, dtype=np.float64)
    if values is not None and values.dtype == np.float32:
        rp = _rp64(rowptrs)
        vs = np.ascontiguousarray(values)
        L.orc_mult_vec_f32vals(C.c_int32(nrows), C.c_int64(nnz), _p(rp, _i64p), _p(ci, _i32p),
                               _p(vs, _f32p), _p(x, _f64p), _p(y, _f64p))
        return y
    vs = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    vp = None if vs is None else _p(vs, _f64p)
    if rowptrs.dtype == np.int32 and rowptrs.flags.c_contiguous:
        L.orc_mult_vec_i32(C.c_int32(nrows), C.c_int64(nnz), _p(rowptrs, _i32p), _p(ci, _i32p),
                           vp, _p(x, _f64p), _p(y, _f64p))
    else:
        rp = _rp64(rowptrs)
        L.orc_mult_vec_i64(C.c_int32(nrows), C.c_int64(nnz), _p(rp, _i64p), _p(ci, _i32p),
                           vp, _p(x, _f64p), _p(y, _f64p))
    return y


def mult_ab(a, b):
    """
    csr/kernels/numba/multiply.py:13-38.  a, b = (nrows, ncols, rowptrs, colinds, values);
    values are required on both (multiply.py:115,120 index .values).  Returns
    (nrows, ncols, c_rp int32, c_ci int32, c_vs f64) with explicit zeros kept and the
    reference's column order (reverse first-discovery).
    """
    anr, anc, arp, aci, avs = a
    bnr, bnc, brp, bci, bvs = b
    assert anc == bnr
    arp, brp = _rp64(arp), _rp64(brp)
    aci, bci = _ci(aci), _ci(bci)
    # float32 times float32 is a float32 product in the reference (multiply.py:120): the C loop rounds it that way
    round32 = int(np.asarray(avs).dtype == np.float32 and np.asarray(bvs).dtype == np.float32)
    avs = np.ascontiguousarray(avs, dtype=np.float64)
    bvs = np.ascontiguousarray(bvs, dtype=np.float64)
    c_rp = np.zeros(anr + 1, dtype=np.int32)
    ci_out = _i32p()
    vs_out = _f64p()
    L = lib()
    n = L.orc_mult_ab(C.c_int32(anr), C.c_int32(anc), C.c_int64(int(arp[anr])),
                      _p(arp, _i64p), _p(aci, _i32p), _p(avs, _f64p),
                      C.c_int32(bnc), C.c_int64(int(brp[bnr])),
                      _p(brp, _i64p), _p(bci, _i32p), _p(bvs, _f64p),
                      _p(c_rp, _i32p), C.byref(ci_out), C.byref(vs_out), C.c_int(round32))
    if n < 0:
        raise MemoryError('oracle mult_ab')
    c_ci = np.ctypeslib.as_array(ci_out, shape=(max(n, 1),))[:n].copy()
    c_vs = np.ctypeslib.as_array(vs_out, shape=(max(n, 1),))[:n].copy()
    L.orc_free(C.cast(ci_out, C.c_void_p))
    L.orc_free(C.cast(vs_out, C.c_void_p))
    return anr, bnc, c_rp, c_ci, c_vs


def sym_mm(a, b):
    "csr/kernels/numba/multiply.py:60-100; returns (c_rp int32, c_ci int32)"
    anr, anc, arp, aci, _ = a
    bnr, bnc, brp, bci, _ = b
    arp, brp = _rp64(arp), _rp64(brp)
    aci, bci = _ci(aci), _ci(bci)
    c_rp = np.zeros(anr + 1, dtype=np.int32)
    ci_out = _i32p()
    L = lib()
    n = L.orc_sym_mm(C.c_int32(anr), C.c_int32(anc), C.c_int64(int(arp[anr])),
                     _p(arp, _i64p), _p(aci, _i32p),
                     C.c_int32(bnc), C.c_int64(int(brp[bnr])), _p(brp, _i64p), _p(bci, _i32p),
                     _p(c_rp, _i32p), C.byref(ci_out))
    if n < 0:
        raise MemoryError('oracle sym_mm')
    c_ci = np.ctypeslib.as_array(ci_out, shape=(max(n, 1),))[:n].copy()
    L.orc_free(C.cast(ci_out, C.c_void_p))
    return c_rp, c_ci


def transpose(nrows, ncols, rowptrs, colinds, values, include_values=True):
    """
    csr/structure.py:172-247.  Returns (ncols, nrows, brp, bci, bvs): brp keeps the input
    pointer dtype (:175), bci int32 (:176), bvs float64 or None (:177, :241-242).
    """
    nnz = int(rowptrs[nrows])
    rp = _rp64(rowptrs)
    ci = _ci(colinds)
    if values is None:
        include_values = False
    brp = np.zeros(ncols + 1, dtype=np.int64)
    bci = np.zeros(nnz, dtype=np.int32)
    bvs = np.zeros(nnz, dtype=np.float64) if include_values else None
    f32 = 0
    vp = None
    if include_values:
        if values.dtype == np.float32:
            vs = np.ascontiguousarray(values)
            f32 = 1
        else:
            vs = np.ascontiguousarray(values, dtype=np.float64)
        vp = vs.ctypes.data_as(C.c_void_p)
    lib().orc_transpose(C.c_int32(nrows), C.c_int32(ncols), C.c_int64(nnz),
                        _p(rp, _i64p), _p(ci, _i32p), vp, C.c_int(f32),
                        _p(brp, _i64p), _p(bci, _i32p),
                        None if bvs is None else _p(bvs, _f64p))
    return ncols, nrows, brp.astype(rowptrs.dtype), bci, bvs


def unit_rows(nrows, rowptrs, values):
    "csr/transform.py:29-66; modifies `values` in place, returns norms (values dtype)."
    rp = _rp64(rowptrs)
    assert values.flags.c_contiguous
    norms = np.zeros(nrows, dtype=values.dtype)
    if values.dtype == np.float32:
        lib().orc_unit_rows_f32(C.c_int32(nrows), _p(rp, _i64p), _p(values, _f32p), _p(norms, _f32p))
    elif values.dtype == np.float64:
        lib().orc_unit_rows_f64(C.c_int32(nrows), _p(rp, _i64p), _p(values, _f64p), _p(norms, _f64p))
    else:
        raise TypeError(values.dtype)
    return norms


def center_rows(nrows, rowptrs, values):
    "csr/transform.py:13-26; modifies `values` in place, returns means (values dtype)."
    rp = _rp64(rowptrs)
    assert values.flags.c_contiguous
    means = np.zeros(nrows, dtype=values.dtype)
    if values.dtype == np.float32:
        lib().orc_center_rows_f32(C.c_int32(nrows), _p(rp, _i64p), _p(values, _f32p), _p(means, _f32p))
    elif values.dtype == np.float64:
        lib().orc_center_rows_f64(C.c_int32(nrows), _p(rp, _i64p), _p(values, _f64p), _p(means, _f64p))
    else:
        raise TypeError(values.dtype)
    return means


def filter_zeros(nrows, rowptrs, colinds, values):
    "csr/_struct.py:61-76; returns new (rowptrs, colinds, values) (copies)."
    rp = _rp64(rowptrs).copy()
    ci = _ci(colinds).copy()
    vs = np.ascontiguousarray(values, dtype=np.float64).copy()
    n = lib().orc_filter_zeros(C.c_int32(nrows), _p(rp, _i64p), _p(ci, _i32p), _p(vs, _f64p))
    return rp.astype(rowptrs.dtype), ci[:n], vs[:n]


def sort_rows(nrows, rowptrs, colinds, values):
    "csr/structure.py:156-169; returns sorted copies (colinds, values)."
    rp = _rp64(rowptrs)
    ci = _ci(colinds).copy()
    vs = None if values is None else np.ascontiguousarray(values, dtype=np.float64).copy()
    lib().orc_sort_rows(C.c_int32(nrows), _p(rp, _i64p), _p(ci, _i32p),
                        None if vs is None else _p(vs, _f64p))
    return ci, vs


def pick_rows(rowptrs, colinds, values, rows, include_values=True):
    """
    csr/csr.py:347-364 -> csr/structure.py:84-149: the picked rows, in the order given (repeats allowed),
    as (rowptrs int32, colinds, values-or-None); values keep their dtype.
    """
    rp = _rp64(rowptrs)
    ci = _ci(colinds)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    vs = np.ascontiguousarray(values) if (include_values and values is not None) else None
    vsize = 0 if vs is None else vs.dtype.itemsize
    vp = None if vs is None else vs.ctypes.data_as(C.c_void_p)
    nnz = lib().orc_pick_rows(_p(rp, _i64p), _p(ci, _i32p), vp, C.c_int32(vsize), _p(rows, _i32p),
                              C.c_int64(len(rows)), None, None, None)
    orp = np.empty(len(rows) + 1, dtype=np.int64)
    oci = np.empty(nnz, dtype=np.int32)
    ovs = None if vs is None else np.empty(nnz, dtype=vs.dtype)
    lib().orc_pick_rows(_p(rp, _i64p), _p(ci, _i32p), vp, C.c_int32(vsize), _p(rows, _i32p), C.c_int64(len(rows)),
                        _p(orp, _i64p), _p(oci, _i32p), None if ovs is None else ovs.ctypes.data_as(C.c_void_p))
    return orp.astype(np.int32), oci, ovs


def from_coo(nrows, rows, cols, values=None):
    """
    csr/structure.py:11-67 as reached from CSR.from_coo (csr/csr.py:138-169): -> (rowptrs, colinds, values-or-None).
    Entries of a row keep their input order; values keep their dtype; rowptrs are int32 unless nnz > INT32_MAX
    (the CSR constructor's narrowing, csr/csr.py:88-93).
    """
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    nnz = len(rows)
    assert len(cols) == nnz and (values is None or len(values) == nnz)
    vs = None if values is None else np.ascontiguousarray(values)
    rp = np.empty(nrows + 1, dtype=np.int64)
    oci = np.empty(nnz, dtype=np.int32)
    ovs = None if vs is None else np.empty(nnz, dtype=vs.dtype)
    rc = lib().orc_from_coo(C.c_int32(nrows), C.c_int64(nnz), _p(rows, _i32p), _p(cols, _i32p),
                            None if vs is None else vs.ctypes.data_as(C.c_void_p),
                            C.c_int32(0 if vs is None else vs.dtype.itemsize), _p(rp, _i64p), _p(oci, _i32p),
                            None if ovs is None else ovs.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise MemoryError('oracle from_coo')
    return (rp.astype(np.int32) if nnz <= np.iinfo(np.int32).max else rp), oci, ovs


def spmm_dense(nrows, rowptrs, colinds, values, B):
    "C = A @ B for dense row-major B (restates multiply.py:110-122 with B fully populated)."
    rp = _rp64(rowptrs)
    ci = _ci(colinds)
    vs = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    k = B.shape[1]
    Cm = np.empty((nrows, k), dtype=np.float64)
    lib().orc_spmm_dense(C.c_int32(nrows), _p(rp, _i64p), _p(ci, _i32p),
                         None if vs is None else _p(vs, _f64p),
                         _p(B, _f64p), C.c_int32(k), C.c_int64(k), _p(Cm, _f64p), C.c_int64(k))
    return Cm


# ---- host-side index arithmetic the reference does in numpy -------------------------

def row_nnzs(rowptrs):
    "csr/csr.py:432-441"
    return np.diff(rowptrs)


def row_extent(rowptrs, row):
    "csr/_rows.py:9-13"
    return rowptrs[row], rowptrs[row + 1]


def shard_splits(rowptrs, tgt_nnz):
    """
    csr/csr.py:599-621 `_shard_rows`, restated on row pointers only: returns the list of
    (begin_row, end_row) ranges.  Raises ValueError when a single row exceeds the target.
    """
    assert tgt_nnz > 0
    rowptrs = np.asarray(rowptrs)
    nrows = len(rowptrs) - 1
    out = []
    base = 0
    while int(rowptrs[nrows] - rowptrs[base]) > tgt_nnz:
        rel = rowptrs[base:] - rowptrs[base]
        split = int(np.searchsorted(rel, tgt_nnz))
        if rel[split] > tgt_nnz:
            if split <= 1:
                raise ValueError("row too large to fit in target matrix size")
            split -= 1
        out.append((base, base + split))
        base += split
    out.append((base, nrows))
    return out
