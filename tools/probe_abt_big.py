# scratch (GPU box): a 20000 x 20000^T block of the MovieLens-shaped matrix (3.8e9 products, 4e8 outputs) through
# csrk_spgemm_abt, against the oracle bit for bit (row pointers, columns, values)
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth, CSR
from csr_amd.kernels import hip as K
from oracle import oracle as O
NA, NB = int(os.environ.get('NA', 20000)), int(os.environ.get('NB', 20000))
m = synth.movielens_like(device='cpu')
M = CSR(m['nrows'], m['ncols'], int(m['colinds'].numel()), m['rowptrs'].numpy(), m['colinds'].numpy(), m['values'].numpy(), _cast=False)
A, B = M.subset_rows(0, NA), M.subset_rows(0, NB)
ah, bh = K.to_handle(A), K.to_handle(B)
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ch = K.mult_abt(ah, bh); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if i == 0: K.release_handle(ch)
print(f'GPU {dt*1e3:.1f} ms', flush=True)
Cm = K.from_handle(ch); K.release_handle(ch)
t0 = time.perf_counter()
bt = O.transpose(B.nrows, B.ncols, B.rowptrs, B.colinds, B.values)
nr, nc, crp, cci, cvs = O.mult_ab((A.nrows, A.ncols, A.rowptrs, A.colinds, A.values), bt)
print(f'oracle {time.perf_counter()-t0:.1f} s, nnz {len(cci)}', flush=True)
assert np.array_equal(Cm.rowptrs, crp)
# the oracle's columns come in reverse discovery order: sort inside rows (stable by construction: distinct columns)
rows = np.repeat(np.arange(nr, dtype=np.int64), np.diff(crp))
o = np.lexsort((cci, rows))
assert np.array_equal(Cm.colinds, cci[o])
assert np.array_equal(Cm.values.view(np.int64), cvs[o].view(np.int64))
print('bit-exact', flush=True)
