import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev = 'cuda'
nr, nc, nnz = 162_541, 59_047, 25_000_095
m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
h = handle_t(0)
check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
ts = []
for i in range(40):
    t = handle_t(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.csrk_transpose(h, 1, C.byref(t))); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    check(lib.csrk_free(t))
ts = ts[5:]
print('transpose: min %.4f ms median %.4f ms' % (min(ts), sorted(ts)[len(ts)//2]), os.environ.get('CSRK_LIBRARY', 'default'))
