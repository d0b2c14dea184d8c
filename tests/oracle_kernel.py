"""
TEST-ONLY kernel module: the reference's kernel protocol on top of the CPU oracle.  It plays
the role the `scipy` kernel plays in the reference's test-suite (csr/kernels/scipy.py: a
comparator that cannot be used in production) and lets the CPU suite exercise the host-side
caller logic of csr_amd.CSR (handle lifetime, row sharding, zero filtering) without a GPU.
It lives under tests/ and is never importable from the product package.
"""
import numpy as np

from oracle import oracle as O

max_nnz = np.iinfo('i8').max
live_handles = 0


class _H:
    def __init__(self, csr):
        self.csr = csr
        self.released = False


def to_handle(csr):
    global live_handles
    if csr.nnz > max_nnz:
        raise ValueError('CSR size {} exceeds max nnz {}'.format(csr.nnz, max_nnz))
    live_handles += 1
    return _H(csr)


def _product_handle(csr):
    "products are not subject to max_nnz (it limits what the CALLER may hand over)"
    global live_handles
    live_handles += 1
    return _H(csr)


def from_handle(h):
    assert not h.released
    return h.csr


def release_handle(h):
    global live_handles
    if not h.released:
        live_handles -= 1
    h.released = True


def order_columns(h):
    c = h.csr
    ci, vs = O.sort_rows(c.nrows, c.rowptrs, c.colinds, c.values)
    c.colinds[...] = ci
    if vs is not None:
        c.values[...] = vs


def mult_vec(h, v):
    c = h.csr
    assert not h.released
    return O.mult_vec(c.nrows, c.ncols, c.rowptrs, c.colinds, c.values, v)


def _tup(c):
    return c.nrows, c.ncols, c.rowptrs, c.colinds, c.values


def mult_ab(a_h, b_h):
    from csr_amd import CSR
    assert not a_h.released and not b_h.released
    nr, nc, rp, ci, vs = O.mult_ab(_tup(a_h.csr), _tup(b_h.csr))
    return _product_handle(CSR(nr, nc, len(ci), rp, ci, vs))


def mult_abt(a_h, b_h):
    from csr_amd import CSR
    b = b_h.csr
    nr, nc, rp, ci, vs = O.transpose(b.nrows, b.ncols, b.rowptrs, b.colinds, b.values)
    bt = CSR(nr, nc, len(ci), rp, ci, vs)
    nr, nc, rp, ci, vs = O.mult_ab(_tup(a_h.csr), _tup(bt))
    return _product_handle(CSR(nr, nc, len(ci), rp, ci, vs))
