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

Handle cache.  The reference's callers make a handle per operation (CSR.mult_vec: to_handle -> mult_vec ->
release_handle, csr/csr.py:580-583); for a matrix in HBM that is a PCIe copy of the whole matrix per product, and
libcsrk's planned kernels (built on a handle's second product) are never reached.  docs/kernels.rst:69-80 lets a
kernel copy in to_handle and obliges the caller to release explicitly, so `to_handle` here keeps the device copy of
a released handle alive and hands it out again when the SAME CSR object comes back with the SAME arrays -- but only
where a stale copy cannot go unseen:
  * by default only for CSR classes that declare `__csrk_cacheable__ = True` (csr_amd.CSR does): their own mutators
    call `invalidate`, their arrays own their memory, and WHILE A DEVICE COPY IS CACHED THE THREE HOST ARRAYS ARE
    WRITE-PROTECTED (ndarray.flags.writeable = False; restored when the copy is dropped) -- `A.values[i] = v` between
    two products raises "assignment destination is read-only" instead of multiplying by the old matrix.  Call
    `invalidate(A)` first (it drops the copy and restores the flags), then edit.  (NumPy raises that error itself:
    the message cannot name `invalidate`; INTEGRATION.md section 2a does.)  The reference's numba handle
    aliases the host arrays (csr/kernels/numba/__init__.py:16-27), so there an edit is simply seen; here it is
    either seen or refused, never missed.  (A view of an array taken BEFORE its first product keeps its own
    writeable flag -- the one alias the guard cannot reach; the sampled fingerprint below is the second line.)
  * CSRK_HANDLE_CACHE=1 extends the cache to every CSR class WITHOUT the write guard (fingerprint only: a single
    element poked into a foreign class's arrays is not seen -- for callers who never edit in place);
    CSRK_HANDLE_CACHE=0 turns it off: every to_handle copies, like the MKL kernel (csr/kernels/mkl/handle.py:61-70);
  * key: id(csr) + shape + nnz + the three arrays' data pointers and dtypes; the entry dies with the CSR object
    (weakref.finalize), so a recycled id or address can never hit; a sampled fingerprint of the arrays (first / last
    32 + 2048 strided elements each) is compared on every hit;
  * a cached copy is handed out only while IDLE: a second live handle on the same CSR gets a copy of its own, so the
    in-place protocol operations (order_columns, unit_rows, center_rows), which also detach their handle from the
    cache, never change a device copy another live handle is reading;
  * idle device copies are evicted least-recently-used beyond CSRK_HANDLE_CACHE_BYTES (default 16 GiB; a copy
    counts with its SpMV / SpMM plans) and whenever a library call runs out of device memory (retried once).
"""
import ctypes as C
import os
import collections
import threading
import weakref

import numpy as np

from .. import _lib
from .._lib import lib, check, ptr, handle_t

# int64 row pointers are supported on the device, so the limit is HBM, not the index type
# (numba kernel: i8.max, csr/kernels/numba/__init__.py:13; MKL: i4.max, mkl/__init__.py:5)
max_nnz = np.iinfo('i8').max

_VAL_CODES = {None: _lib.VAL_NONE, np.dtype('f4'): _lib.VAL_F32, np.dtype('f8'): _lib.VAL_F64}


class hip_h:
    "Opaque handle (cf. mkl_h, csr/kernels/mkl/handle.py:30-43): H is the csrk_handle_t."
    __slots__ = ('H', 'nrows', 'ncols', 'nnz', 'csr_ref', '_entry')

    def __init__(self, H, nrows, ncols, nnz, csr_ref=None):
        self.H = H
        self.nrows = nrows
        self.ncols = ncols
        self.nnz = nnz
        self.csr_ref = csr_ref
        self._entry = None          # the handle cache's record when the device copy is shared (module docstring)

    def __repr__(self):
        return f'<hip_h {self.nrows}x{self.ncols} ({self.nnz} nnz) H={self.H:#x}>'


def _live(h):
    if not h.H:
        raise ValueError('handle has been released')
    return h.H


# ---- handle cache ------------------------------------------------------------------------------------------
class _Entry:
    __slots__ = ('H', 'refs', 'bytes', 'key', 'finger', 'tick', 'cached', 'fin', 'guard')


_cache_lock = threading.RLock()
_cache = {}                     # key -> _Entry (live or idle device copies that may be handed out again)
_guards = {}                    # id(array) -> [array, entries guarding it]: host arrays write-protected while cached
_tick = 0
_SAMPLE = 2048


def _cache_mode():
    "'off' | 'guarded' (classes that declare __csrk_cacheable__, write-protected arrays) | 'all' (fingerprint only)"
    v = os.environ.get('CSRK_HANDLE_CACHE')
    if v is None or v == '':
        return 'guarded'
    return 'off' if v in ('0', 'off', 'false') else 'all'


def _cache_budget():
    return int(os.environ.get('CSRK_HANDLE_CACHE_BYTES', str(16 << 30)))


def _arr_key(a):
    return (0, '') if a is None else (a.ctypes.data, a.dtype.str)


def _sample(a):
    if a is None or a.size == 0:
        return b''
    n = a.size
    if n <= 4 * _SAMPLE:
        return a.tobytes()
    step = n // _SAMPLE
    return a[:32].tobytes() + a[::step].tobytes() + a[-32:].tobytes()


def _fingerprint(rps, cis, vs):
    return hash((_sample(rps), _sample(cis), _sample(vs)))


def _protect(arrays):
    "write-protect the host arrays of a cached copy (caller holds the lock); returns what to hand to _unprotect"
    held = []
    for a in arrays:
        if a is None:
            continue
        g = _guards.get(id(a))
        if g is not None and g[0] is a:
            g[1] += 1                    # already guarded for another cached copy that shares it (CSR.copy(copy_structure=False))
            held.append(a)
            continue
        if not a.flags.writeable:        # read-only before we came: not ours to restore
            continue
        _guards[id(a)] = [a, 1]
        a.flags.writeable = False
        held.append(a)
    return held


def _unprotect(held):
    for a in held:
        g = _guards.get(id(a))
        if g is None:
            continue
        g[1] -= 1
        if g[1] <= 0:
            del _guards[id(a)]
            a.flags.writeable = True


def _drop_entry(e):
    "remove from the index; free the device copy once nobody holds it (caller holds the lock)"
    if e.cached:
        e.cached = False
        if _cache.get(e.key) is e:
            del _cache[e.key]
    if e.guard:
        held, e.guard = e.guard, None
        _unprotect(held)
    if e.refs == 0 and e.H:
        H, e.H = e.H, 0
        check(lib.csrk_free(H))


def _on_csr_collected(key):
    with _cache_lock:
        e = _cache.get(key)
        if e is not None:
            _drop_entry(e)


def _evict_idle(budget):
    "free idle cached copies, least recently used first, until the idle ones fit `budget` bytes"
    with _cache_lock:
        idle = sorted((e for e in _cache.values() if e.refs == 0), key=lambda e: e.tick)
        total = sum(e.bytes for e in idle)
        for e in idle:
            if total <= budget:
                break
            total -= e.bytes
            _drop_entry(e)


def flush_handle_cache():
    "free every idle cached device copy (live handles are untouched)"
    _evict_idle(0)


def invalidate(csr):
    """
    Forget the cached device copy of `csr` and make its arrays writable again: call BEFORE editing them in place
    (csr_amd.CSR's own mutators do); the next to_handle copies afresh.  Cached copies of OTHER matrices that share one
    of csr's arrays (CSR.copy(copy_structure=False)) are dropped with it: the edit would reach them too.
    """
    mine = [a for a in (getattr(csr, 'rowptrs', None), getattr(csr, 'colinds', None), getattr(csr, 'values', None))
            if a is not None]
    with _cache_lock:
        for e in list(_cache.values()):
            if e.key[0] == id(csr) or (e.guard and any(g is a for g in e.guard for a in mine)):
                _drop_entry(e)


def _call(fn, *args):
    """
    A library call that may need device memory: on failure, give back what idle cached copies hold and retry once.
    Only for calls that leave their operands untouched when they fail (products, transposes, copies): the in-place
    operations (order_columns, unit_rows, center_rows) are never retried -- a second pass over a half-updated matrix
    would report success with wrong norms.
    """
    rc = fn(*args)
    if rc == _lib.ERR_HIP:
        with _cache_lock:
            idle = any(e.refs == 0 for e in _cache.values())
        if idle:
            flush_handle_cache()
            check(lib.csrk_trim_cache())
            rc = fn(*args)
    check(rc)


# ---- result arrays -----------------------------------------------------------------------------------------
# The protocol returns FRESH host arrays (mult_vec: csr/kernels/numba/__init__.py:57, from_handle: :30-36).  A fresh
# np.empty of 80 MB is 20 000 unmapped pages: the device-to-host copy into it faults every one of them in (measured on
# the MI355X host: 8.3 ms against 1.43 ms into memory that has been touched before -- the product itself is 0.54 ms).
# Large results therefore come from blocks that have been through that once: a result array is a view of a LEASE on a
# block; when the caller has dropped the array and every view of it, the block goes back to the idle list (at most
# CSRK_RESULT_POOL_BYTES idle, default 2 GiB; 0 = plain np.empty everywhere).  The caller sees an ordinary writable
# ndarray that nobody else holds (`owndata` is False: its memory belongs to the lease).
_POOL_MIN = 1 << 20
_pool_lock = threading.RLock()
_pool = []                      # idle blocks: uint8 arrays whose pages are mapped
_pool_bytes = 0
_leases = {}                    # id(lease) -> weak reference: the leases behind live result arrays (an array whose base
                                # is one owns its memory alone)
# Blocks whose lease has died, not yet back on the idle list.  A lease's finalizer can run inside ANY allocation -- the
# cyclic collector frees a result array held in a reference cycle -- including allocations made while _pool_lock is held:
# it therefore takes no lock and only appends here (deque.append is atomic); _out / flush_result_pool /
# result_pool_bytes drain the queue under the lock.
_returned = collections.deque()


def _pool_cap():
    return int(os.environ.get('CSRK_RESULT_POOL_BYTES', str(2 << 30)))


def _give_back(blk, lease_id):
    _returned.append((blk, lease_id))


def _drain_returned(cap):
    "caller holds _pool_lock"
    global _pool_bytes
    while True:
        try:
            blk, lease_id = _returned.popleft()
        except IndexError:
            return
        r = _leases.get(lease_id)
        if r is not None and r() is None:      # (a NEW lease may sit at the dead one's address: its entry stays)
            del _leases[lease_id]
        if _pool_bytes + blk.nbytes <= cap:
            _pool.append(blk)
            _pool_bytes += blk.nbytes


def _is_lease(obj):
    r = _leases.get(id(obj))
    return r is not None and r() is obj


def flush_result_pool():
    "free the idle result blocks"
    global _pool_bytes
    with _pool_lock:
        _drain_returned(0)
        _pool.clear()
        _pool_bytes = 0


def result_pool_bytes():
    "bytes of idle result blocks (blocks whose arrays have been dropped included)"
    cap = _pool_cap()
    with _pool_lock:
        _drain_returned(cap)
        return _pool_bytes


def _out(shape, dtype):
    "an uninitialised array for a library call to fill: recycled memory for large results (see above)"
    global _pool_bytes
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64))
    nbytes = n * dtype.itemsize
    cap = _pool_cap()
    if nbytes < _POOL_MIN or cap <= 0:
        return np.empty(shape, dtype=dtype)
    blk = None
    with _pool_lock:
        _drain_returned(cap)
        best = -1
        for i, b in enumerate(_pool):      # smallest idle block that fits without wasting more than a quarter
            if nbytes <= b.nbytes <= nbytes + nbytes // 4 and (best < 0 or b.nbytes < _pool[best].nbytes):
                best = i
        if best >= 0:
            blk = _pool.pop(best)
            _pool_bytes -= blk.nbytes
    if blk is None:
        blk = np.empty(nbytes, dtype=np.uint8)
    lease = (C.c_char * blk.nbytes).from_buffer(blk)
    weakref.finalize(lease, _give_back, blk, id(lease)).atexit = False
    _leases[id(lease)] = weakref.ref(lease)      # (a dict store: atomic; a finalizer running inside it only appends to _returned)
    arr = np.frombuffer(lease, dtype=dtype, count=n)      # (its base is the lease itself: what the handle cache checks)
    return arr if np.ndim(shape) == 0 or len(shape) == 1 else arr.reshape(shape)


def _create(csr, rps, cis, vs):
    out = handle_t(0)
    _call(lib.csrk_create, int(csr.nrows), int(csr.ncols), int(csr.nnz), ptr(rps), int(rps.dtype == np.dtype('i8')), ptr(cis),
          ptr(vs), _VAL_CODES[None if vs is None else vs.dtype], C.byref(out))
    return out.value


def to_handle(csr):
    """
    csr/kernels/numba/__init__.py:16-27; csr/kernels/mkl/handle.py:61-70.  Copies the
    matrix to HBM -- or hands out the idle cached copy made for this same CSR object and arrays (module docstring).
    Accepts f4/f8/absent values and int32/int64 row pointers; other value
    dtypes are widened to f8 (the reference's results are f8 whatever the storage dtype).
    """
    global _tick
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
    nr, nc, nnz = int(csr.nrows), int(csr.ncols), int(csr.nnz)
    mode = _cache_mode()
    guarded = mode == 'guarded'
    # small matrices are cheaper to copy than to look up; arrays that had to be converted are temporaries whose
    # addresses mean nothing on the next call; in guarded mode the class must vouch for its mutators and every array
    # must own its memory (a view's base could be written behind the guard)
    def own(a, orig):
        if a is None and orig is None:
            return True
        if not (isinstance(orig, np.ndarray) and a.ctypes.data == orig.ctypes.data):
            return False
        return not guarded or orig.base is None or _is_lease(orig.base)      # (a result array of this module: _out)
    # (a matrix with live subset_rows views is copied per handle: the views write through to its host arrays unguarded)
    cacheable = (mode != 'off' and nnz >= 4096 and (not guarded or getattr(type(csr), '__csrk_cacheable__', False))
                 and not getattr(csr, '_views', None)
                 and own(rps, csr.rowptrs) and own(cis, csr.colinds) and own(vs, csr.values))
    if cacheable:
        key = (id(csr), nr, nc, nnz, _arr_key(rps), _arr_key(cis), _arr_key(vs))
        finger = _fingerprint(rps, cis, vs)
        with _cache_lock:
            _tick += 1
            e = _cache.get(key)
            if e is not None and e.finger != finger:      # same arrays, different contents: edited in place
                _drop_entry(e)
                e = None
            if e is not None and e.refs > 0:
                # another live handle is using the cached copy (and may change it in place: order_columns,
                # unit_rows, center_rows): this one gets a copy of its own, outside the cache
                cacheable = False
                e = None
            if e is not None:
                e.refs += 1
                e.tick = _tick
                h = hip_h(e.H, nr, nc, nnz, csr)
                h._entry = e
                return h
    H = _create(csr, rps, cis, vs)
    h = hip_h(H, nr, nc, nnz, csr)
    if cacheable:
        try:
            fin = weakref.finalize(csr, _on_csr_collected, key)
        except TypeError:                                  # a CSR type without weak references (Numba structref proxy)
            return h
        fin.atexit = False
        e = _Entry()
        e.H, e.refs, e.key, e.finger, e.cached, e.fin, e.guard = H, 1, key, finger, True, fin, None
        e.bytes = rps.nbytes + cis.nbytes + (0 if vs is None else vs.nbytes)
        with _cache_lock:
            old = _cache.get(key)
            if old is not None:                            # another thread cached the same matrix meanwhile
                _drop_entry(old)
            e.tick = _tick
            _cache[key] = e
            if guarded:
                e.guard = _protect([csr.rowptrs, csr.colinds, csr.values])
        h._entry = e
    return h


def _detach(h):
    "an in-place operation is about to change this handle's device copy: it must not be handed out as `csr` again"
    e = h._entry
    if e is not None:
        with _cache_lock:
            if e.cached:
                e.cached = False
                if _cache.get(e.key) is e:
                    del _cache[e.key]
            if e.guard:
                held, e.guard = e.guard, None
                _unprotect(held)


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
    rps = _out(nr + 1, np.int64 if p64 else np.int32)
    cis = _out(nnz, np.int32)
    vs = None if vt == _lib.VAL_NONE else _out(nnz, np.float32 if vt == _lib.VAL_F32 else np.float64)
    check(lib.csrk_export(h.H, ptr(rps), ptr(cis), ptr(vs)))
    return CSR(nr, nc, nnz, rps, cis, vs, _cast=False)


def release_handle(h):
    """
    csr/kernels/numba/__init__.py:39-44; idempotent like mkl/handle.py:144-148.  A cached device copy stays in HBM
    (idle) for the next to_handle of the same CSR; the others are freed here.
    """
    e, h._entry = h._entry, None
    H, h.H = h.H, 0
    h.csr_ref = None
    if not H:
        return
    if e is None:
        check(lib.csrk_free(H))
        return
    with _cache_lock:
        e.refs -= 1
        if e.refs == 0 and not e.cached:
            _drop_entry(e)
        elif e.cached:
            nb = C.c_int64(0)                      # the copy stays: count it with the plans it has grown meanwhile
            if lib.csrk_device_bytes(H, C.byref(nb)) == _lib.OK and nb.value > 0:
                e.bytes = nb.value
    if e.cached:
        _evict_idle(_cache_budget())


def order_columns(h):
    "csr/kernels/numba/__init__.py:47-52: sort each row by column, in place on the handle"
    _detach(h)
    check(lib.csrk_order_columns(_live(h)))         # in place: never retried (see _call)


def mult_vec(h, v):
    """
    csr/kernels/numba/__init__.py:55-67: y = A v as a fresh float64[nrows].
    v may be any real dtype of shape (ncols,).  A float32 v goes to the library as float32: Numba types the reference's
    loop by its operands, so float32 values times a float32 v is a float32 product (one rounding) added to the float64
    accumulator (csrk_spmv_f32x does the same; with float64 or absent values it widens v).  Everything else is widened to
    float64 here (exact).
    """
    v = np.asarray(v)
    if v.shape != (h.ncols,):
        raise ValueError(f'vector has shape {v.shape}, expected ({h.ncols},)')
    y = _out(h.nrows, np.float64)
    if v.dtype == np.float32:
        x = np.ascontiguousarray(v)
        _call(lib.csrk_spmv_f32x, _live(h), ptr(x), ptr(y))
        return y
    x = np.ascontiguousarray(v, dtype=np.float64)
    _call(lib.csrk_spmv, _live(h), ptr(x), ptr(y))
    return y


def mult_ab(a_h, b_h):
    "csr/kernels/numba/multiply.py:13-38: C = A B as a NEW handle the caller must release"
    assert a_h.ncols == b_h.nrows
    out = handle_t(0)
    _call(lib.csrk_spgemm_ab, _live(a_h), _live(b_h), C.byref(out))
    return _wrap(out.value)


def mult_abt(a_h, b_h):
    "csr/kernels/numba/multiply.py:41-57: C = A B^T as a NEW handle"
    assert a_h.ncols == b_h.ncols
    out = handle_t(0)
    _call(lib.csrk_spgemm_abt, _live(a_h), _live(b_h), C.byref(out))
    return _wrap(out.value)


# ---- beyond the protocol: same hot path, not kernel-dispatched in the reference ------------

def transpose(h, include_values=True):
    "csr/structure.py:240-247: transposed matrix as a NEW handle (bit-exact with the reference)"
    out = handle_t(0)
    _call(lib.csrk_transpose, _live(h), int(bool(include_values)), C.byref(out))
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
    _call(lib.csrk_from_coo, nrows, ncols, nnz, ptr(rows), ptr(cols), ptr(vals),
          _VAL_CODES[None if vals is None else vals.dtype], C.byref(out))
    return _wrap(out.value)


def row_nnzs(h):
    "csr/csr.py:432-441"
    _, _, _, p64, _ = _info(_live(h))
    out = _out(h.nrows, np.int64 if p64 else np.int32)
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
    _detach(h)                     # unit_rows / center_rows rewrite the device copy's values
    out = _out(h.nrows, np.float32 if vt == _lib.VAL_F32 else np.float64)
    check(fn(h.H, ptr(out)))                        # in place: never retried (see _call)
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
    _call(lib.csrk_pick_rows, _live(h), rows.ctypes.data_as(C.c_void_p), rows.size, int(bool(include_values)), C.byref(out))
    return _wrap(out.value)


def filter_zeros(h):
    "csr/_struct.py:61-76 on the device: NEW handle without exact-zero entries"
    out = handle_t(0)
    _call(lib.csrk_filter_zeros, _live(h), C.byref(out))
    return _wrap(out.value)


def values_of(h):
    "copy only the value array out of HBM (after an in-place row operation)"
    _, _, nnz, _, vt = _info(_live(h))
    if vt == _lib.VAL_NONE:
        return None
    vs = _out(nnz, np.float32 if vt == _lib.VAL_F32 else np.float64)
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
    out = _out((h.nrows, k), np.float64)
    _call(lib.csrk_spmm_dense, _live(h), ptr(B), k, k, ptr(out), k)
    return out


def set_spgemm_order(order):
    """
    Column order inside the rows mult_ab / mult_abt return: 'reference' (default) -- reverse order of first discovery,
    what csr/kernels/numba/multiply.py:79-82, 94-97 emit --, 'ascending' (the product kernels' own: no ordering pass) or
    None (follow CSRK_SPGEMM_ORDER: "ascending", else the reference's).  Process-wide.  The values are the same bits
    either way.
    """
    code = {None: -1, 'ascending': 0, 'reference': 1}[order]
    check(lib.csrk_spgemm_set_order(code))


def spgemm_order():
    "the column order in force: 'reference' or 'ascending'"
    o = C.c_int(0)
    check(lib.csrk_spgemm_get_order(C.byref(o)))
    return 'reference' if o.value else 'ascending'


def spgemm_last_route():
    """
    what this thread's last mult_ab / mult_abt took: 'dense-panel' -- B was a fully populated CSR in row-major panel form
    (BASELINE configs[2] through the reference's own entry: csr/csr.py:524-567 -> multiply.py:13-38) -- or 'general'
    """
    r = C.c_int(0)
    check(lib.csrk_spgemm_last_route(C.byref(r)))
    return 'dense-panel' if r.value else 'general'


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
