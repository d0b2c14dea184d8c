import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, time, numpy as np, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
def run(name, nr, nc, nnz, **kw):
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, **kw)
    vals = m['values'].clone()
    h = handle_t(0)
    check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), vals.data_ptr(), 2, C.byref(h)))
    out = torch.empty(nr, dtype=torch.float64, device=dev)
    for op, fn in (('unit_rows', lib.csrk_unit_rows_device), ('center_rows', lib.csrk_center_rows_device)):
        ts=[]
        for i in range(4):
            vals.copy_(m['values']); torch.cuda.synchronize(); t0=time.perf_counter()
            check(fn(h, out.data_ptr())); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
        alg = 2 * nnz * 8 + nr * 12
        print(f'{name:12s} {op:12s} wall ms {min(ts):8.3f}  (algorithmic {alg/1e9:.2f} GB: values read + written, row pointers, norms -> {alg/min(ts)/1e6:.0f} GB/s = {alg/min(ts)/1e6/8000:.3f} of 8 TB/s; norms stay on the device)', flush=True)
    check(lib.csrk_free(h))
shape = os.environ.get('ROWOPS_SHAPE', 'both')
if shape in ('ml', 'both'):
    run('cfg5 ML25M', 162_541, 59_047, 25_000_095, alpha=0.9, max_degree=7000)
if shape in ('headline', 'both'):
    run('headline', 10_000_000, 10_000_000, 200_000_000)
