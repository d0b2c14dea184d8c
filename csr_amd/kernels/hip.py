"""
The `hip` kernel: the reference's kernel-module protocol (csr/kernel.py:9-16,
docs/kernels.rst:61-104) implemented on libcsrk.so -- hand-written HIP kernels for
MI355X (gfx950) behind the C ABI of include/csrk.h.

Protocol members (same names and meaning as csr/kernels/numba/__init__.py:13-67 and
csr/kernels/mkl/*): max_nnz, to_handle, from_handle, release_handle, order_columns,
mult_ab, mult_abt, mult_vec.  Extra members for the operations the reference runs
outside its kernel protocol but on the same hot path: transpose, row_nnzs, unit_rows,
center_rows, filter_zeros, pick_rows, mult_dense.

A handle owns a copy of the matrix in HBM, like the MKL kernel's handle
(csr/kernels/mkl/handle.py:47-70).  There is no CPU fallback: without a GPU every call
raises csr_amd._lib.CsrkError.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import lib, check, ptr, handle_t

# int64 row pointers are supported on the device, so the limit is HBM, not the index type
# (numba kernel: i8.max, csr/kernels/numba/__init__.py:13; MKL: i4.max, mkl/__init__.py:5)
max_nnz = np.iinfo('i8').max

_VAL_CODES = {None: _lib.VAL_NONE, np.dtype('f4'): _lib.VAL_F32, np.dtype('f8'): _lib.VAL_F64}


class hip_h:
    "Opaque handle (cf. mkl_h, csr/kernels/mkl/handle.py:30-43): H is the csrk_handle_t."
    __slots__ = ('H', 'nrows', 'ncols', 'nnz', 'csr_ref')

    def __init__(self, H, nrows, ncols, nnz, csr_ref=None):
        self.H = H
        self.nrows = nrows
        self.ncols = ncols
        self.nnz = nnz
        self.csr_ref = csr_ref

    def __repr__(self):
        return f'<hip_h {self.nrows}x{self.ncols} ({self.nnz} nnz) H={self.H:#x}>'


def _live(h):
    if not h.H:
        raise ValueError('handle has been released')
    return h.H


def to_handle(csr):
    """
    csr/kernels/numba/__init__.py:16-27; csr/kernels/mkl/handle.py:61-70.  Copies the
    matrix to HBM.  Accepts f4/f8/absent values and int32/int64 row pointers; other value
    dtypes are widened to f8 (the reference's results are f8 whatever the storage dtype).
    """
    if csr.nnz > max_nnz:
        raise ValueError('CSR size {} exceeds max nnz {}'.format(csr.nnz, max_nnz))
    rps = np.ascontiguousarray(csr.rowptrs)
    if rps.dtype not in (np.dtype('i4'), np.dtype('i8')):
        rps = rps.astype(np.int64)
    cis = np.ascontiguousarray(csr.colinds, dtype=np.int32)
    vs = csr.values
    if vs is not None:
        vs = np.ascontiguousarray(vs)
        if vs.dtype not in (np.dtype('f4'), np.dtype('f8')):
            vs = vs.astype(np.float64)
    out = handle_t(0)
    check(lib.csrk_create(int(csr.nrows), int(csr.ncols), int(csr.nnz), ptr(rps),
                          int(rps.dtype == np.dtype('i8')), ptr(cis), ptr(vs),
                          _VAL_CODES[None if vs is None else vs.dtype], C.byref(out)))
    return hip_h(out.value, int(csr.nrows), int(csr.ncols), int(csr.nnz), csr)


def _info(H):
    nr, nc, nnz = C.c_int32(), C.c_int32(), C.c_int64()
    p64, vt = C.c_int(), C.c_int()
    check(lib.csrk_info(H, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(p64), C.byref(vt)))
    return nr.value, nc.value, nnz.value, p64.value, vt.value


def _wrap(H):
    nr, nc, nnz, _, _ = _info(H)
    return hip_h(H, nr, nc, nnz, None)


def from_handle(h):
    """
    csr/kernels/numba/__init__.py:30-36; csr/kernels/mkl/handle.py:95-132.  Copies the
    matrix out of HBM into a fresh host CSR; the handle may be released afterwards.
    """
    from ..csr import CSR
    nr, nc, nnz, p64, vt = _info(_live(h))
    rps = np.empty(nr + 1, dtype=np.int64 if p64 else np.int32)
    cis = np.empty(nnz, dtype=np.int32)
    vs = None if vt == _lib.VAL_NONE else np.empty(nnz, dtype=np.float32 if vt == _lib.VAL_F32 else np.float64)
    check(lib.csrk_export(h.H, ptr(rps), ptr(cis), ptr(vs)))
    return CSR(nr, nc, nnz, rps, cis, vs, _cast=False)


def release_handle(h):
    "csr/kernels/numba/__init__.py:39-44; idempotent like mkl/handle.py:144-148"
    if h.H:
        check(lib.csrk_free(h.H))
    h.H = 0
    h.csr_ref = None


def order_columns(h):
    "csr/kernels/numba/__init__.py:47-52: sort each row by column, in place on the handle"
    check(lib.csrk_order_columns(_live(h)))


def mult_vec(h, v):
    """
    csr/kernels/numba/__init__.py:55-67: y = A v as a fresh float64[nrows].
    v may be any real dtype of shape (ncols,); it is widened to float64 (exact for f4).
    """
    x = np.ascontiguousarray(v, dtype=np.float64)
    if x.shape != (h.ncols,):
        raise ValueError(f'vector has shape {x.shape}, expected ({h.ncols},)')
    y = np.empty(h.nrows, dtype=np.float64)
    check(lib.csrk_spmv(_live(h), ptr(x), ptr(y)))
    return y


def mult_ab(a_h, b_h):
    "csr/kernels/numba/multiply.py:13-38: C = A B as a NEW handle the caller must release"
    assert a_h.ncols == b_h.nrows
    out = handle_t(0)
    check(lib.csrk_spgemm_ab(_live(a_h), _live(b_h), C.byref(out)))
    return _wrap(out.value)


def mult_abt(a_h, b_h):
    "csr/kernels/numba/multiply.py:41-57: C = A B^T as a NEW handle"
    assert a_h.ncols == b_h.ncols
    out = handle_t(0)
    check(lib.csrk_spgemm_abt(_live(a_h), _live(b_h), C.byref(out)))
    return _wrap(out.value)


# ---- beyond the protocol: same hot path, not kernel-dispatched in the reference ------------

def transpose(h, include_values=True):
    "csr/structure.py:240-247: transposed matrix as a NEW handle (bit-exact with the reference)"
    out = handle_t(0)
    check(lib.csrk_transpose(_live(h), int(bool(include_values)), C.byref(out)))
    return _wrap(out.value)


def from_coo(rows, cols, vals, shape):
    """
    csr/structure.py:11-67 on the device: COO arrays -> a NEW handle (entries of a row keep their
    input order, like the reference's counting sort).  `vals` may be None (structure only).
    """
    nrows, ncols = (int(v) for v in shape)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    nnz = len(rows)
    if len(cols) != nnz or (vals is not None and len(vals) != nnz):
        raise ValueError('rows, cols and vals must have the same length')
    if nnz and (rows.min() < 0 or rows.max() >= max(nrows, 1) or cols.min() < 0 or cols.max() >= max(ncols, 1)):
        raise ValueError('COO coordinates out of range')
    if vals is not None:
        vals = np.ascontiguousarray(vals)
        if vals.dtype not in (np.dtype('f4'), np.dtype('f8')):
            vals = vals.astype(np.float64)
    out = handle_t(0)
    check(lib.csrk_from_coo(nrows, ncols, nnz, ptr(rows), ptr(cols), ptr(vals),
                            _VAL_CODES[None if vals is None else vals.dtype], C.byref(out)))
    return _wrap(out.value)


def row_nnzs(h):
    "csr/csr.py:432-441"
    _, _, _, p64, _ = _info(_live(h))
    out = np.empty(h.nrows, dtype=np.int64 if p64 else np.int32)
    check(lib.csrk_row_nnzs(h.H, ptr(out)))
    return out


def row_extent(h, row):
    "csr/_rows.py:9-13"
    s, e = C.c_int64(), C.c_int64()
    check(lib.csrk_row_extent(_live(h), int(row), C.byref(s), C.byref(e)))
    return s.value, e.value


def _row_stat(fn, h):
    _, _, _, _, vt = _info(_live(h))
    if vt == _lib.VAL_NONE:
        raise ValueError('matrix has no values')
    out = np.empty(h.nrows, dtype=np.float32 if vt == _lib.VAL_F32 else np.float64)
    check(fn(h.H, ptr(out)))
    return out


def unit_rows(h):
    "csr/transform.py:29-66: unit-normalise rows IN PLACE on the handle; returns the norms"
    return _row_stat(lib.csrk_unit_rows, h)


def center_rows(h):
    "csr/transform.py:13-26: mean-centre rows IN PLACE on the handle; returns the means"
    return _row_stat(lib.csrk_center_rows, h)


def pick_rows(h, rows, include_values=True):
    "csr/csr.py:347-364 on the device: NEW handle with the given rows (in order, repeats allowed)"
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = handle_t(0)
    check(lib.csrk_pick_rows(_live(h), rows.ctypes.data_as(C.c_void_p), rows.size, int(bool(include_values)), C.byref(out)))
    return _wrap(out.value)


def filter_zeros(h):
    "csr/_struct.py:61-76 on the device: NEW handle without exact-zero entries"
    out = handle_t(0)
    check(lib.csrk_filter_zeros(_live(h), C.byref(out)))
    return _wrap(out.value)


def values_of(h):
    "copy only the value array out of HBM (after an in-place row operation)"
    _, _, nnz, _, vt = _info(_live(h))
    if vt == _lib.VAL_NONE:
        return None
    vs = np.empty(nnz, dtype=np.float32 if vt == _lib.VAL_F32 else np.float64)
    check(lib.csrk_export(h.H, None, None, ptr(vs)))
    return vs


def mult_dense(h, B):
    """
    C = A B for a dense row-major B [ncols x k] (BASELINE.json configs[2]); equals
    mult_ab(A, CSR(B)) densified.  Not a reference entry point.
    """
    B = np.ascontiguousarray(B, dtype=np.float64)
    if B.ndim != 2 or B.shape[0] != h.ncols:
        raise ValueError(f'panel has shape {B.shape}, expected ({h.ncols}, k)')
    k = B.shape[1]
    out = np.empty((h.nrows, k), dtype=np.float64)
    check(lib.csrk_spmm_dense(_live(h), ptr(B), k, k, ptr(out), k))
    return out


def set_spmv_algo(h, name):
    "select the SpMV kernel for this handle: 'auto' | 'merge' | 'vector' | 'scalar'"
    code = {'auto': _lib.SPMV_AUTO, 'merge': _lib.SPMV_MERGE, 'vector': _lib.SPMV_VECTOR,
            'scalar': _lib.SPMV_SCALAR}[name]
    check(lib.csrk_set_spmv_algo(_live(h), code))


def device_count():
    n = C.c_int(0)
    check(lib.csrk_device_count(C.byref(n)))
    return n.value


def set_device(i):
    check(lib.csrk_set_device(int(i)))
