# scratch (GPU box): cost of the SpMV plan on the headline matrix, first handle of the process and a later one
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
n, nnz = 10_000_000, 200_000_000
dev = torch.device('cuda', 0)
m = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    h = handle_t(0)
    check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    free, total = torch.cuda.mem_get_info()
    print(f'handle {rep}: first call {ts[0]:.2f} ms, plan call {ts[1]:.2f} ms, third {ts[2]:.3f} ms; device memory in use {(total-free)/1e9:.2f} GB', flush=True)
    check(lib.csrk_free(h))
