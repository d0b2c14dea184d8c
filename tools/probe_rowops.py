import ctypes as C, time, numpy as np, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
def run(name, nr, nc, nnz, **kw):
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, **kw)
    vals = m['values'].clone()
    h = handle_t(0)
    check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), vals.data_ptr(), 2, C.byref(h)))
    out = np.empty(nr)
    for op, fn in (('unit_rows', lib.csrk_unit_rows), ('center_rows', lib.csrk_center_rows)):
        ts=[]
        for i in range(4):
            vals.copy_(m['values']); torch.cuda.synchronize(); t0=time.perf_counter()
            check(fn(h, out.ctypes.data_as(C.c_void_p))); ts.append((time.perf_counter()-t0)*1e3)
        print(f'{name:12s} {op:12s} wall ms {min(ts):8.3f}  (values {nnz*8/1e6:.0f} MB -> {4*nnz*8/min(ts)/1e6:.0f} GB/s for 3 reads + 1 write; D2H of norms {nr*8/1e6:.0f} MB included)', flush=True)
    check(lib.csrk_free(h))
run('cfg5 ML25M', 162_541, 59_047, 25_000_095, alpha=0.9, max_degree=7000)
run('headline', 10_000_000, 10_000_000, 200_000_000)
