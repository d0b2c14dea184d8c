import ctypes as C, time, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
n=10_000_000; nnz=200_000_000
m = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
for rep in range(3):
    h = handle_t(0)
    check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    ts=[]
    for i in range(4):
        torch.cuda.synchronize(); t0=time.perf_counter()
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None)); torch.cuda.synchronize()
        ts.append((time.perf_counter()-t0)*1e3)
    print('rep', rep, ['%.2f' % t for t in ts], flush=True)
    check(lib.csrk_free(h))
