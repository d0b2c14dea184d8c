"""
Host-side CSR container: the struct the reference keeps in csr/csr.py:46-100
(nrows, ncols, nnz, rowptrs, colinds, values) with the methods that sit on the
`csr.kernels` hot path.  The arrays live in NumPy on the host exactly as in the
reference; every arithmetic method hands them to the active kernel (csr_amd.kernels.hip by
default), which runs hand-written HIP kernels on the MI355X.  Only index bookkeeping that
the reference itself does in NumPy on the host (subset_rows views, shard split points,
shard reassembly) is done here.

Layout rules kept verbatim (csr/csr.py:79-100): rowptrs int32 when nnz <= INT32_MAX else
int64; colinds int32; values any float dtype or None; all C-contiguous.
"""
import logging
import sys
import weakref

import numpy as np

from .kernels import get_kernel, releasing

INTC = np.iinfo(np.intc)
_log = logging.getLogger(__name__)


class CSR:
    """
    Compressed sparse row matrix (host arrays), drop-in for the reference's `csr.CSR` on the
    kernel hot path.

    Attributes: nrows, ncols, nnz, rowptrs, colinds, values (None = structure only).
    """

    def __init__(self, nrows, ncols, nnz, rps, cis, vs, _cast=True):
        # csr/csr.py:79-100
        assert nrows >= 0 and nrows <= INTC.max
        assert ncols >= 0 and ncols <= INTC.max
        assert nnz >= 0
        self.nrows = int(nrows)
        self.ncols = int(ncols)
        self.nnz = int(nnz)
        if _cast:
            cis = np.require(cis, np.intc, 'C')
            if nnz <= INTC.max:
                rps = np.require(rps, np.intc, 'C')
            else:
                rps = np.require(rps, np.int64, 'C')
            if vs is not None:
                vs = np.require(vs, requirements='C')
        self.rowptrs = rps
        self.colinds = cis
        self._values = vs

    # While a device copy of this matrix is cached (csr_amd/kernels/hip.py, handle cache) its three arrays are
    # write-protected; every method here that changes them calls _edited() first.
    __csrk_cacheable__ = True

    _parent = None      # subset_rows: the matrix whose colinds / values this one views
    _views = None       # subset_rows: the live sub-matrices that view this one's arrays (weak): while there are any, no
                        # device copy of this matrix is cached -- a write through a view would not be seen

    def _edited(self):
        """
        The arrays are about to change: drop cached device copies (restores the arrays' writeable flags) -- of this
        matrix and of every matrix it is a row range of (subset_rows hands out views: the edit reaches the parent).
        """
        if self._parent is not None:
            self._parent._edited()
        from . import kernels as _k
        mods = list(_k.kernels.values())
        hip = sys.modules.get(_k.__name__ + '.hip')       # (loaded but not yet looked up through the registry)
        if hip is not None and hip not in mods:
            mods.append(hip)
        for kern in mods:
            inv = getattr(kern, 'invalidate', None)
            if inv is not None:
                inv(self)

    # ---- construction ---------------------------------------------------------------------
    @classmethod
    def empty(cls, nrows, ncols, row_nnzs=None, values=True):
        """
        A matrix of the given shape with zeroed arrays: no entries, or (row_nnzs) room for that many per row;
        `values`: True = float64, a dtype, or False = structure only.  (csr/csr.py:102-136 is the counterpart.)
        """
        if nrows < 0 or ncols < 0:
            raise ValueError('negative shape')
        counts = np.zeros(nrows, dtype=np.int64) if row_nnzs is None else np.asarray(row_nnzs, dtype=np.int64)
        if counts.shape != (nrows,):
            raise ValueError('row_nnzs must have one count per row')
        rps = np.concatenate(([0], np.cumsum(counts)))
        nnz = int(rps[-1])
        if not values:
            vs = None
        else:
            vs = np.zeros(nnz, dtype=np.float64 if values is True else values)
        return cls(nrows, ncols, nnz, rps, np.zeros(nnz, dtype=np.intc), vs)

    @classmethod
    def from_coo(cls, rows, cols, vals, shape=None):
        """
        csr/csr.py:138-169 -> csr/structure.py:11-67: stable counting sort of the COO entries
        by row (entries of a row keep their input order).  Host-side ingest, not on the hot
        path (out of scope); done with a stable NumPy argsort.
        """
        rows = np.asarray(rows)
        cols = np.asarray(cols)
        assert np.min(rows, initial=0) >= 0 and np.min(cols, initial=0) >= 0
        if shape is not None:
            nrows, ncols = shape
            assert np.max(rows, initial=0) < max(nrows, 1)
            assert np.max(cols, initial=0) < max(ncols, 1)
        else:
            nrows = int(np.max(rows)) + 1
            ncols = int(np.max(cols)) + 1
        nnz = len(rows)
        assert len(cols) == nnz and (vals is None or len(vals) == nnz)
        order = np.argsort(rows, kind='stable')
        rps = np.zeros(nrows + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=nrows), out=rps[1:])
        return cls(nrows, ncols, nnz, rps, cols[order], None if vals is None else np.asarray(vals)[order])

    @classmethod
    def from_scipy(cls, mat, copy=True):
        """
        csr/csr.py:171-192: any SciPy sparse matrix -> CSR (through .tocsr() when it is not CSR already).
        copy=False shares SciPy's arrays where their dtypes allow.  Host only.
        """
        import scipy.sparse as sps
        if not sps.isspmatrix_csr(mat):
            mat, copy = mat.tocsr(), False           # the conversion already made fresh arrays
        parts = []
        for a, dt in ((mat.indptr, np.intc), (mat.indices, np.intc), (mat.data, None)):
            b = np.require(a, dt, 'C')
            parts.append(b.copy() if copy and np.shares_memory(a, b) else b)
        return cls(mat.shape[0], mat.shape[1], mat.nnz, *parts)

    def to_scipy(self):
        "csr/csr.py:194-209: scipy.sparse.csr_matrix over the same arrays (a structure-only matrix gets 1.0 values)"
        import scipy.sparse as sps
        vs = np.ones(self.nnz) if self._values is None else self._values
        return sps.csr_matrix((vs, self.colinds, self.rowptrs), shape=(self.nrows, self.ncols))

    # ---- fields ---------------------------------------------------------------------------
    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, vs):
        "replace the value array (None = structure only); a longer array is cut to nnz (csr/csr.py:224-242)"
        if vs is not None:
            vs = np.ascontiguousarray(vs)
            if vs.shape[0] < self.nnz:
                raise ValueError('value array too small')
            vs = vs[:self.nnz]
        self._edited()
        self._values = vs

    def copy(self, include_values=True, *, copy_structure=True):
        """
        A matrix of its own with the same entries: fresh arrays (copy_structure=False shares rowptrs / colinds
        instead), without values if include_values is false.  (Counterpart: csr/csr.py:298-321.)
        """
        rps, cis = (self.rowptrs.copy(), self.colinds.copy()) if copy_structure else (self.rowptrs, self.colinds)
        vs = self._values.copy() if include_values and self._values is not None else None
        return type(self)(self.nrows, self.ncols, self.nnz, rps, cis, vs, _cast=False)

    # ---- rows -----------------------------------------------------------------------------
    def row_extent(self, row):
        "csr/csr.py:406-417 -> csr/_rows.py:9-13 (host field read, as in the reference)"
        return self.rowptrs[row], self.rowptrs[row + 1]

    def row_nnzs(self):
        "csr/csr.py:432-441.  Host diff like the reference; the device version is kernel.row_nnzs."
        return np.diff(self.rowptrs)

    def rowinds(self):
        "csr/csr.py:366-371 -> csr/_rows.py:116-122: the row index of every stored entry (COO row array), intc"
        return np.repeat(np.arange(self.nrows, dtype=np.intc), np.diff(self.rowptrs))

    def row_cs(self, row):
        "csr/csr.py:419-423: the column indices stored for `row` (a view)"
        lo, hi = self.row_extent(row)
        return self.colinds[lo:hi]

    def row_vs(self, row):
        "csr/csr.py:425-430: the values stored for `row` (a view); 1.0 per entry for a structure-only matrix"
        lo, hi = self.row_extent(row)
        return np.ones(hi - lo) if self._values is None else self._values[lo:hi]

    def _dense_rows(self, row, dtype, ones):
        row = np.asarray(row, dtype=np.int32)
        out = np.zeros(row.shape + (self.ncols,), dtype=dtype)
        for dst, r in zip(out.reshape(-1, self.ncols), row.reshape(-1)):
            lo, hi = self.row_extent(r)
            dst[self.colinds[lo:hi]] = 1 if ones else self._values[lo:hi]
        return out

    def row(self, row):
        """
        csr/csr.py:373-388 -> csr/_rows.py:70-82: one row (or, for an index array, one row per index) densified:
        stored values, 0 elsewhere; a structure-only matrix gives float32 ones.  Host only.
        """
        if self._values is None:
            return self._dense_rows(row, np.float32, True)
        return self._dense_rows(row, self._values.dtype, False)

    def row_mask(self, row):
        "csr/csr.py:390-404: like row(), but True where the row stores an entry"
        return self._dense_rows(row, np.bool_, True)

    def subset_rows(self, begin, end):
        """
        csr/csr.py:331-346 -> csr/structure.py:70-81: views of colinds/values, rebased pointers.  The views write
        through to this matrix, as in the reference: a cached device copy of this matrix is dropped first (views taken
        under the write guard would stay read-only for good), and the sub-matrix's own mutators drop it again.
        """
        self._edited()
        lo, hi = int(self.rowptrs[begin]), int(self.rowptrs[end])
        sub = CSR(end - begin, self.ncols, hi - lo, self.rowptrs[begin:end + 1] - lo, self.colinds[lo:hi],
                  None if self._values is None else self._values[lo:hi])
        sub._parent = self
        if self._views is None:
            self._views = weakref.WeakSet()
        self._views.add(sub)
        return sub

    def pick_rows(self, rows, *, include_values=True):
        """
        csr/csr.py:347-364 -> csr/structure.py:84-149: the given rows, in order (a row may appear more than
        once), as a new matrix; values are dropped with include_values=False.  Runs on the device when the
        active kernel provides `pick_rows`.
        """
        rows = np.asarray(rows)
        assert rows.ndim == 1
        K, pick = self._ext('pick_rows')
        with releasing(K.to_handle(self), K) as h:
            with releasing(pick(h, rows, include_values), K) as ph:
                return K.from_handle(ph)

    # ---- device operations beyond the kernel protocol -----------------------------------------
    def _ext(self, name):
        K = get_kernel()
        fn = getattr(K, name, None)
        if fn is None:
            raise NotImplementedError(f'kernel {K.__name__} does not provide {name}')
        return K, fn

    def transpose(self, include_values=True):
        """
        csr/csr.py:471-486 -> csr/structure.py:240-247.  Runs on the device; bit-exact with the
        reference (stable counting sort, float64 output values, input pointer width).
        """
        K, tr = self._ext('transpose')
        with releasing(K.to_handle(self), K) as h:
            with releasing(tr(h, include_values), K) as th:
                return K.from_handle(th)

    def transpose_structure(self):
        return self.transpose(False)

    def normalize_rows(self, normalization):
        "csr/csr.py:443-469 -> csr/transform.py: in place; returns the per-row norms / means"
        if normalization not in ('center', 'unit'):
            raise ValueError('unknown normalization: ' + normalization)
        K, fn = self._ext('center_rows' if normalization == 'center' else 'unit_rows')
        with releasing(K.to_handle(self), K) as h:
            stat = fn(h)
            vs = K.values_of(h)
        self._edited()
        self._values[...] = vs
        return stat

    def sort_rows(self):
        "csr/csr.py:323-329 -> csr/structure.py:156-169, in place, via the kernel's order_columns"
        K = get_kernel()
        with releasing(K.to_handle(self), K) as h:
            K.order_columns(h)
            out = K.from_handle(h)
        self._edited()
        self.colinds[...] = out.colinds
        if self._values is not None:
            self._values[...] = out.values

    # ---- the kernel protocol's callers --------------------------------------------------------
    def multiply(self, other, transpose=False):
        """
        csr/csr.py:524-567: A @ B (or A @ B^T).  Handle lifetime, row sharding above
        K.max_nnz and the exact-zero filter on the product follow the reference.
        """
        if transpose:
            assert self.ncols == other.ncols
        else:
            assert self.ncols == other.nrows
        K = get_kernel()
        dev_filter = getattr(K, 'filter_zeros', None)

        def mul(A, b_h):
            with releasing(K.to_handle(A), K) as a_h:
                c_h = K.mult_abt(a_h, b_h) if transpose else K.mult_ab(a_h, b_h)
                with releasing(c_h, K):
                    if dev_filter is not None:
                        with releasing(dev_filter(c_h), K) as f_h:
                            return K.from_handle(f_h)
                    crepr = K.from_handle(c_h)
            crepr._filter_zeros()
            return crepr

        with releasing(K.to_handle(other), K) as b_h:
            # one handle of B serves every row block of A; a single block is returned as it is
            blocks = [mul(blk, b_h) for blk in self._row_blocks(K.max_nnz)]
        return blocks[0] if len(blocks) == 1 else CSR._assemble_shards(blocks)

    def mult_vec(self, v):
        "csr/csr.py:569-590: y = A v; above K.max_nnz the row blocks' products are concatenated in row order"
        v = np.asarray(v)
        assert v.shape == (self.ncols,)
        K = get_kernel()
        ys = []
        for blk in self._row_blocks(K.max_nnz):
            with releasing(K.to_handle(blk), K) as h:
                ys.append(K.mult_vec(h, v))
        return ys[0] if len(ys) == 1 else np.concatenate(ys)

    def _row_blocks(self, limit):
        "the matrix itself when it fits the kernel's max_nnz, else its _shard_rows blocks"
        return [self] if self.nnz <= limit else self._shard_rows(limit)

    def _filter_zeros(self):
        """
        csr/csr.py:592-597 -> csr/_struct.py:61-76, host flavour for kernels without a device
        filter: drop entries whose value is exactly 0 (NaN stays), in place.
        """
        if self.values is None:
            return
        keep = self.values != 0
        self._edited()
        cum = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
        self.rowptrs = cum[self.rowptrs].astype(self.rowptrs.dtype)
        self.colinds = np.ascontiguousarray(self.colinds[keep])
        self._values = np.ascontiguousarray(self.values[keep])
        self.nnz = int(cum[-1])

    def _shard_rows(self, tgt_nnz):
        """
        csr/csr.py:599-621: consecutive row blocks of at most tgt_nnz entries each, cut greedily: a block ends at the
        last row boundary that keeps it within the target; a single row larger than the target cannot be placed.
        Pinned by tests/golden/shard.npz (the reference's own cuts).
        """
        assert tgt_nnz > 0
        ptr = self.rowptrs.astype(np.int64)
        cuts = [0]
        while int(ptr[-1]) - int(ptr[cuts[-1]]) > tgt_nnz:
            first = cuts[-1]
            room = int(ptr[first]) + tgt_nnz
            # first row boundary at or past the target; when it overshoots, the boundary before it ends the block --
            # unless that is the block's own start: then its first row alone exceeds the target
            nxt = first + int(np.searchsorted(ptr[first:], room))
            if ptr[nxt] > room:
                if nxt - first <= 1:
                    raise ValueError("row too large to fit in target matrix size")
                nxt -= 1
            _log.debug('%s: row block [%d, %d) holds %d entries', self, first, nxt, int(ptr[nxt]) - int(ptr[first]))
            cuts.append(nxt)
        cuts.append(self.nrows)
        return [self.subset_rows(a, b) for a, b in zip(cuts[:-1], cuts[1:])]

    @classmethod
    def _assemble_shards(cls, shards):
        "csr/csr.py:623-650: stack row blocks (same ncols up to trailing width) back into one matrix, rows in order"
        counts = np.concatenate([np.diff(s.rowptrs) for s in shards]) if shards else np.zeros(0, np.int64)
        rps = np.zeros(len(counts) + 1, np.int64)
        np.cumsum(counts, out=rps[1:])
        nnz = sum(s.nnz for s in shards)
        assert rps[-1] == nnz, f'{rps[-1]} != {nnz}'
        cis = np.concatenate([s.colinds for s in shards])
        vs = None if shards[0].values is None else np.concatenate([s.values for s in shards])
        return cls(len(counts), max(s.ncols for s in shards), nnz, rps, cis, vs)

    def __str__(self):
        return '<CSR {}x{} ({} nnz)>'.format(self.nrows, self.ncols, self.nnz)

    __repr__ = __str__

    def __reduce__(self):
        "csr/csr.py:690-692"
        return (CSR, (self.nrows, self.ncols, self.nnz, self.rowptrs, self.colinds, self.values, False))
