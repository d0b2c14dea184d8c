"""
The handle cache of csr_amd/kernels/hip.py (host logic only: libcsrk's create / free are replaced by counters, so this
runs without a GPU).  The reference's callers make a handle per operation (csr/csr.py:580-583: to_handle -> mult_vec ->
release_handle); the cache keeps the released device copy for the next to_handle of the same CSR object and arrays.
"""
import ctypes as C
import gc

import numpy as np
import pytest


class FakeLib:
    "counts csrk_create / csrk_free; handles are 1, 2, 3, ..."

    def __init__(self):
        self.created, self.freed, self.live, self.fail_next, self.bytes_of = 0, [], set(), 0, {}

    def csrk_create(self, nr, nc, nnz, rp, p64, ci, vs, vt, out):
        if self.fail_next:
            self.fail_next -= 1
            return -2
        self.created += 1
        out._obj.value = self.created
        self.live.add(self.created)
        return 0

    def csrk_free(self, H):
        H = getattr(H, 'value', H)
        assert H in self.live, f'double free of {H}'
        self.live.discard(H)
        self.freed.append(H)
        return 0

    def csrk_trim_cache(self):
        return 0

    def csrk_device_bytes(self, H, out):
        out._obj.value = self.bytes_of.get(getattr(H, 'value', H), 0)      # 0: keep the host-side estimate
        return 0

    def csrk_last_error(self):
        return b'hipMalloc failed: out of memory'


@pytest.fixture
def K(monkeypatch):
    import csr_amd.kernels.hip as hip
    fake = FakeLib()
    monkeypatch.setattr(hip, 'lib', fake)
    import csr_amd._lib as _lib
    monkeypatch.setattr(_lib, 'lib', fake)
    hip.flush_handle_cache()
    hip._cache.clear()
    yield hip
    hip.flush_handle_cache()
    hip._cache.clear()
    assert not hip._guards                            # every write guard was lifted with its entry


def _mat(n=6000, seed=0, vals=True):
    from csr_amd import CSR
    rng = np.random.default_rng(seed)
    rp = np.arange(n + 1, dtype=np.int32)
    return CSR(n, n, n, rp, rng.integers(0, n, n).astype(np.int32), rng.uniform(-1, 1, n) if vals else None, _cast=False)


def test_same_csr_reuses_the_device_copy(K):
    A = _mat()
    h1 = K.to_handle(A)
    assert K.lib.created == 1
    K.release_handle(h1)
    assert h1.H == 0 and K.lib.freed == []            # idle, still in HBM
    K.release_handle(h1)                              # idempotent (csr/kernels/mkl/handle.py:144-148)
    h2 = K.to_handle(A)
    assert K.lib.created == 1 and h2.H == 1           # the idle copy is handed out again
    K.release_handle(h2)
    assert K.lib.freed == []
    B = _mat(seed=1)                                  # another matrix: its own copy
    hb = K.to_handle(B)
    assert K.lib.created == 2 and hb.H == 2
    K.release_handle(hb)
    K.flush_handle_cache()
    assert sorted(K.lib.freed) == [1, 2] and not K.lib.live


def test_two_live_handles_never_share_a_device_copy(K, monkeypatch):
    "an in-place protocol operation on one handle must not change what another live handle reads (every to_handle of the reference yields an independent object)"
    A = _mat()
    h1 = K.to_handle(A)
    h2 = K.to_handle(A)                               # the cached copy is in use: this one is private
    assert K.lib.created == 2 and h1.H != h2.H
    monkeypatch.setattr(K.lib, 'csrk_order_columns', lambda H: 0, raising=False)
    K.order_columns(h2)                               # h1's copy is untouched
    K.release_handle(h2)
    assert K.lib.freed == [h2.H or 2]                 # the private copy is freed on release
    K.release_handle(h1)
    h3 = K.to_handle(A)
    assert h3.H == 1 and K.lib.created == 2           # idle again: reused
    K.release_handle(h3)


def test_entry_dies_with_the_csr_object(K):
    A = _mat()
    K.release_handle(K.to_handle(A))
    assert K.lib.live == {1}
    del A
    gc.collect()
    assert not K.lib.live and not K._cache
    # a live handle keeps its matrix alive (csr_ref); the copy goes when the handle is released and the CSR collected
    A = _mat()
    h = K.to_handle(A)
    del A
    gc.collect()
    assert K.lib.live == {2}
    K.release_handle(h)
    gc.collect()
    assert not K.lib.live


def test_in_place_edits_are_refused_or_seen_never_missed(K):
    "csr/csr.py:580-583: the reference re-reads the arrays on every product; here an edit of a cached matrix raises until invalidate()"
    A = _mat()
    K.release_handle(K.to_handle(A))
    for arr in (A.values, A.colinds, A.rowptrs):
        assert not arr.flags.writeable                # write-protected while the device copy is cached
    with pytest.raises(ValueError):
        A.values[4001] = 7.0                          # ONE element: refused, not silently ignored
    with pytest.raises(ValueError):
        A.values[:] *= 2.0
    K.invalidate(A)                                   # the caller announces the edit ...
    assert A.values.flags.writeable and A.colinds.flags.writeable and K.lib.freed == [1]
    A.values[4001] = 7.0                              # ... and makes it
    h = K.to_handle(A)
    assert K.lib.created == 2                         # fresh copy
    K.release_handle(h)
    A.values = A.values * 2.0                         # csr_amd.CSR's own mutators invalidate by themselves
    assert K.lib.freed == [1, 2] and A.values.flags.writeable
    h = K.to_handle(A)
    assert K.lib.created == 3
    K.release_handle(h)
    K.invalidate(A)
    A.colinds = A.colinds.copy()                      # new arrays on the same object: different key
    h = K.to_handle(A)
    assert K.lib.created == 4
    K.release_handle(h)


def test_foreign_csr_classes_are_cached_only_on_request(K, monkeypatch):
    "a class that does not declare __csrk_cacheable__ (the reference's own CSR) gets a fresh copy per handle unless CSRK_HANDLE_CACHE=1"
    class Foreign:
        pass
    A = _mat()
    F = Foreign()
    F.nrows, F.ncols, F.nnz, F.rowptrs, F.colinds, F.values = A.nrows, A.ncols, A.nnz, A.rowptrs, A.colinds, A.values
    for _ in range(2):
        K.release_handle(K.to_handle(F))
    assert K.lib.created == 2 and K.lib.freed == [1, 2] and F.values.flags.writeable
    monkeypatch.setenv('CSRK_HANDLE_CACHE', '1')      # fingerprint-validated, no write guard: the caller's responsibility
    for _ in range(2):
        K.release_handle(K.to_handle(F))
    assert K.lib.created == 3 and F.values.flags.writeable
    F.values[:] *= 2.0                                # a whole-array edit changes the sampled fingerprint
    K.release_handle(K.to_handle(F))
    assert K.lib.created == 4


def test_views_are_not_cached(K):
    "an array that does not own its memory can be written through its base behind the guard"
    from csr_amd import CSR
    A = _mat()
    S = A.subset_rows(0, 5000)                        # colinds / values are views of A's
    for _ in range(2):
        K.release_handle(K.to_handle(S))
    assert K.lib.created == 2 and A.values.flags.writeable


def test_in_place_protocol_operations_detach(K, monkeypatch):
    A = _mat()
    h = K.to_handle(A)
    monkeypatch.setattr(K.lib, 'csrk_order_columns', lambda H: 0, raising=False)
    K.order_columns(h)                                # the device copy no longer equals A
    assert A.values.flags.writeable                   # ... and no longer guards A's arrays
    h2 = K.to_handle(A)
    assert h2.H != h.H and K.lib.created == 2
    K.release_handle(h)
    assert K.lib.freed == [1]                         # detached copies are freed on release
    K.release_handle(h2)


def test_budget_evicts_least_recently_used(K, monkeypatch):
    mats = [_mat(seed=s) for s in range(4)]
    per = mats[0].rowptrs.nbytes + mats[0].colinds.nbytes + mats[0].values.nbytes
    monkeypatch.setenv('CSRK_HANDLE_CACHE_BYTES', str(2 * per))
    for m in mats:
        K.release_handle(K.to_handle(m))
    assert K.lib.freed == [1, 2] and K.lib.live == {3, 4}
    K.release_handle(K.to_handle(mats[2]))            # touch 3: 4 is now the oldest
    K.release_handle(K.to_handle(mats[0]))
    assert K.lib.freed == [1, 2, 4]
    monkeypatch.setenv('CSRK_HANDLE_CACHE', '0')
    n = K.lib.created
    h = K.to_handle(mats[2])
    assert K.lib.created == n + 1                     # cache off: always a fresh copy, freed on release
    K.release_handle(h)
    assert K.lib.freed[-1] == h.H or K.lib.freed[-1] == n + 1


def test_out_of_memory_flushes_idle_copies_and_retries(K):
    A, B = _mat(seed=0), _mat(seed=1)
    K.release_handle(K.to_handle(A))
    K.lib.fail_next = 1
    h = K.to_handle(B)
    assert K.lib.freed == [1] and h.H == 2
    K.release_handle(h)
    K.lib.fail_next = 2                               # nothing idle left to give back... (B's copy is, once)
    with pytest.raises(Exception):
        K.to_handle(_mat(seed=2))


def test_small_and_converted_matrices_are_not_cached(K):
    from csr_amd import CSR
    S = CSR(3, 3, 2, np.array([0, 1, 2, 2], dtype=np.int32), np.array([0, 1], dtype=np.int32), np.ones(2), _cast=False)
    for _ in range(2):
        K.release_handle(K.to_handle(S))
    assert K.lib.created == 2 and K.lib.freed == [1, 2]
    A = _mat()
    A.colinds = A.colinds.astype(np.int64)            # converted to int32 on every call: a new array each time
    for _ in range(2):
        K.release_handle(K.to_handle(A))
    gc.collect()
    assert K.lib.created == 4
    K.flush_handle_cache()
    assert not K.lib.live


def test_shared_structure_is_guarded_for_every_cached_copy(K):
    "CSR.copy(copy_structure=False) shares rowptrs / colinds: each cached copy holds its own reference on the guard"
    A = _mat()
    B = A.copy(copy_structure=False)
    assert B.rowptrs is A.rowptrs and B.colinds is A.colinds
    K.release_handle(K.to_handle(A))
    K.release_handle(K.to_handle(B))
    assert K.lib.created == 2 and not A.colinds.flags.writeable
    e_a = next(e for e in K._cache.values() if e.key[0] == id(A))
    K._drop_entry(e_a)                                # A's copy goes (eviction): B's still guards the shared arrays
    assert not A.colinds.flags.writeable and not B.rowptrs.flags.writeable
    assert A.values.flags.writeable and not B.values.flags.writeable
    with pytest.raises(ValueError):
        A.colinds[0] = 3                              # would reach B's cached device copy unseen
    K.invalidate(B)
    assert A.colinds.flags.writeable and B.values.flags.writeable


def test_invalidate_reaches_copies_that_share_an_array(K):
    A = _mat()
    B = A.copy(copy_structure=False)
    K.release_handle(K.to_handle(A))
    K.release_handle(K.to_handle(B))
    K.invalidate(A)                                   # A's structure is about to change: so is B's
    assert not K.lib.live and A.colinds.flags.writeable and B.values.flags.writeable
    A.colinds[0] = 3
    K.release_handle(K.to_handle(B))
    assert K.lib.created == 3                         # B was copied afresh, with the edit


def test_subset_rows_under_the_guard_gives_writeable_views(K, monkeypatch):
    """
    A product caches A and write-protects its arrays; a row range taken afterwards must still behave like the
    reference's (csr/structure.py:70-81: views that write through).  While such a sub-matrix is alive A is NOT cached
    again: a write through the view -- announced or not -- must reach the next product.
    """
    A = _mat()
    K.release_handle(K.to_handle(A))
    assert not A.values.flags.writeable
    S = A.subset_rows(10, 5000)
    assert S.values.flags.writeable and S.colinds.flags.writeable and np.shares_memory(S.values, A.values)
    assert not K.lib.live                             # taking the views dropped the cached copy
    K.release_handle(K.to_handle(A))                  # copied for this handle only: S could write behind a cached copy
    assert K.lib.created == 2 and not K.lib.live and A.values.flags.writeable
    before = A.values[10]
    S.values[...] = 7.0                               # a view of A's array, written WITHOUT announcing it
    assert A.values[10] == 7.0 and before != 7.0
    K.release_handle(K.to_handle(A))
    assert K.lib.created == 3 and not K.lib.live      # ... and the next handle is a fresh copy that holds the 7.0s
    del S
    gc.collect()
    K.release_handle(K.to_handle(A))                  # no view left: cached (and guarded) again
    assert K.lib.created == 4 and K.lib.live and not A.values.flags.writeable
    K.release_handle(K.to_handle(A))
    assert K.lib.created == 4
    S2 = A.subset_rows(0, 100)                        # the sub-matrix's own mutators still announce their edits up the chain
    S2._edited()
    assert not K.lib.live


def test_in_place_operations_are_not_retried(K, monkeypatch):
    "a failed unit_rows / order_columns may have rewritten part of the matrix: a second pass would double-apply"
    A, B = _mat(seed=0), _mat(seed=1)
    K.release_handle(K.to_handle(B))                  # an idle copy that a retry would flush
    h = K.to_handle(A)
    calls = []
    monkeypatch.setattr(K.lib, 'csrk_info', lambda H, nr, nc, nnz, p64, vt: (setattr(vt._obj, 'value', 2), 0)[1], raising=False)
    monkeypatch.setattr(K.lib, 'csrk_unit_rows', lambda H, out: (calls.append('u'), -2)[1], raising=False)
    monkeypatch.setattr(K.lib, 'csrk_order_columns', lambda H: (calls.append('o'), -2)[1], raising=False)
    with pytest.raises(Exception):
        K.unit_rows(h)
    with pytest.raises(Exception):
        K.order_columns(h)
    assert calls == ['u', 'o'] and len(K.lib.live) == 2
    K.release_handle(h)


# ---- result arrays (hip._out): large results come from recycled, already mapped memory -------------------------
def test_result_arrays_are_recycled_only_when_nobody_holds_them(K, monkeypatch):
    K.flush_result_pool()
    small = K._out(100, np.float64)
    assert small.flags.owndata                        # below 1 MiB: plain np.empty
    n = 300_000                                       # 2.4 MB
    a = K._out(n, np.float64)
    assert a.shape == (n,) and a.dtype == np.float64 and a.flags.writeable and a.flags.c_contiguous
    a[:] = 1.0
    addr = a.ctypes.data
    view = a[10:20]
    del a
    gc.collect()
    b = K._out(n, np.float64)                         # a view of the first result is alive: its memory is NOT handed out
    assert b.ctypes.data != addr and view[0] == 1.0
    del view
    gc.collect()
    assert K.result_pool_bytes() == n * 8             # now it is idle ...
    c = K._out((n // 2, 2), np.float64)               # ... and serves the next result of that size (any shape / dtype)
    assert c.ctypes.data == addr and c.shape == (n // 2, 2) and K.result_pool_bytes() == 0
    d = K._out(n // 4, np.float32)                    # a quarter of the size: the idle block would be mostly waste
    del b, c
    gc.collect()
    assert d.ctypes.data != addr
    monkeypatch.setenv('CSRK_RESULT_POOL_BYTES', '0')
    e = K._out(n, np.float64)
    assert e.flags.owndata                            # pool off: plain arrays, nothing kept
    del d, e
    gc.collect()
    K.flush_result_pool()
    assert K.result_pool_bytes() == 0 and not K._pool


def test_result_pool_survives_a_collection_inside_its_own_lock(K):
    """
    A result array held in a reference cycle is freed by the cyclic collector, which may run inside ANY allocation --
    including those _out makes while it holds the pool's lock.  The lease's finalizer must not need that lock
    (with a plain Lock the process hung in _give_back -> _give_back).
    """
    import faulthandler
    K.flush_result_pool()

    class Box:
        pass

    old = gc.get_threshold()
    faulthandler.dump_traceback_later(60, exit=True)      # a deadlock would otherwise hang the suite
    try:
        for thr in (1, 2, 3, 5, 8):
            gc.set_threshold(thr, 1, 1)                   # a collection every few allocations
            for _ in range(40):
                b = Box()
                b.me = b                                  # the cycle
                b.y = K._out(1 << 18, np.float64)         # 2 MiB: pooled
                del b
                r = K._out(1 << 18, np.float64)           # dropped at once: its finalizer runs right here
                del r
    finally:
        gc.set_threshold(*old)
        faulthandler.cancel_dump_traceback_later()
    gc.collect()
    assert K.result_pool_bytes() <= K._pool_cap()
    K.flush_result_pool()
    assert K.result_pool_bytes() == 0 and not K._pool and not K._returned


def test_a_matrix_made_of_result_arrays_is_cacheable(K):
    "from_handle's arrays come from _out: a CSR built on them must still reach the handle cache (they own their memory alone)"
    from csr_amd import CSR
    n = 400_000
    rp = K._out(n + 1, np.int32)
    rp[:] = np.arange(n + 1, dtype=np.int32)
    ci = K._out(n, np.int32)
    ci[:] = 0
    vs = K._out(n, np.float64)
    vs[:] = 1.0
    assert not rp.flags.owndata
    A = CSR(n, n, n, rp, ci, vs, _cast=False)
    h = K.to_handle(A)
    K.release_handle(h)
    h2 = K.to_handle(A)
    assert K.lib.created == 1 and h2.H == 1           # cached, like a matrix on plain arrays
    with pytest.raises(ValueError):
        A.values[0] = 2.0                             # ... and guarded like one
    K.release_handle(h2)
    K.invalidate(A)
    A.values[0] = 2.0
    del A, rp, ci, vs
    gc.collect()
    K.flush_result_pool()
