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
