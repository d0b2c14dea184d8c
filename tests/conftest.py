import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _gpu_unavailable():
    "reason string when the gpu-marked tests cannot run here (no library, no device), else None"
    try:
        import ctypes as C
        from csr_amd._lib import lib
    except Exception as e:                      # library not built: the product has no CPU fallback
        return f'libcsrk.so is not loadable ({type(e).__name__}: {e})'
    n = C.c_int(0)
    if lib.csrk_device_count(C.byref(n)) != 0 or n.value < 1:
        return 'no MI355X visible (csrk_device_count)'
    return None


def pytest_collection_modifyitems(config, items):
    # A plain `pytest` on a CPU box: the gpu-marked parity tests are skipped with the reason, instead of failing and
    # hiding the host / oracle tests' signal.  On the GPU box nothing is skipped (and `-m gpu` selects them).
    gpu_items = [it for it in items if it.get_closest_marker('gpu')]
    if not gpu_items:
        return
    markexpr = (config.getoption('markexpr', '') or '').replace(' ', '')
    if 'gpu' in markexpr and 'notgpu' not in markexpr:
        return                                  # `-m gpu` was asked for: a missing library or device must FAIL loudly
    why = _gpu_unavailable()
    if why:
        skip = pytest.mark.skip(reason=why)
        for it in gpu_items:
            it.add_marker(skip)


class Mat:
    "plain holder for one golden CSR (reference struct layout, csr/csr.py:79-100)"

    def __init__(self, d, prefix):
        self.nrows, self.ncols, self.nnz = (int(v) for v in d[prefix + 'shape'])
        self.rowptrs = d[prefix + 'rowptrs']
        self.colinds = d[prefix + 'colinds']
        self.values = d[prefix + 'values'] if (prefix + 'values') in d else None

    def tup(self):
        return self.nrows, self.ncols, self.rowptrs, self.colinds, self.values

    def dense(self):
        out = np.zeros((self.nrows, self.ncols))
        for i in range(self.nrows):
            for p in range(int(self.rowptrs[i]), int(self.rowptrs[i + 1])):
                out[i, self.colinds[p]] += 1.0 if self.values is None else float(self.values[p])
        return out


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def sort_within_rows(rowptrs, colinds, values):
    "canonical (column-sorted, stable) copy of a CSR's index/value arrays"
    ci = np.asarray(colinds).copy()
    vs = None if values is None else np.asarray(values).copy()
    for i in range(len(rowptrs) - 1):
        s, e = int(rowptrs[i]), int(rowptrs[i + 1])
        o = np.argsort(ci[s:e], kind='stable')
        ci[s:e] = ci[s:e][o]
        if vs is not None:
            vs[s:e] = vs[s:e][o]
    return ci, vs


def as_library_orders(rowptrs, colinds, values):
    """
    The oracle's raw product (columns in the reference's order, reverse of first discovery) in the column order libcsrk is
    set to emit: as it is under the default, column-sorted under CSRK_SPGEMM_ORDER=ascending / set_spgemm_order('ascending').
    """
    from csr_amd.kernels import hip as K
    if K.spgemm_order() == 'reference':
        return np.asarray(colinds), (None if values is None else np.asarray(values))
    return sort_within_rows(rowptrs, colinds, values)
