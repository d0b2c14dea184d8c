import ctypes as C, time, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
nr, nc, nnz = 162_541, 59_047, 25_000_095
m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
h = handle_t(0)
check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
for keep in (False, True):
    outs=[]
    for i in range(8):
        torch.cuda.synchronize(); t0=time.perf_counter()
        t = handle_t(0); check(lib.csrk_transpose(h, 1, C.byref(t)))
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
        if keep: outs.append(t)
        else: check(lib.csrk_free(t))
        print(f'keep={keep} call {i}: {dt:.3f} ms', flush=True)
    for o in outs: check(lib.csrk_free(o))
