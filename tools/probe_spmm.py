# scratch (GPU box): time the BASELINE configs[2] dense-panel SpMM and its kernels (torch events; no parity check here)
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev = 'cuda'
n, nnz, k = 2_000_000, 50_000_000, int(os.environ.get('K', '64'))
m = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=250_000)
h = handle_t(0)
check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
B = synth.dense_vector(n * k, device=dev, stream=7).view(n, k)
Cm = torch.empty(n, k, dtype=torch.float64, device=dev)
for _ in range(3):
    check(lib.csrk_spmm_dense_device(h, B.data_ptr(), k, k, Cm.data_ptr(), k, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    check(lib.csrk_spmm_dense_device(h, B.data_ptr(), k, k, Cm.data_ptr(), k, None))
e1.record(); torch.cuda.synchronize()
print(os.environ.get('CSRK_LIBRARY', 'default').split('/')[-1], f'{e0.elapsed_time(e1) / 10:.3f} ms  checksum {float(Cm.sum()):.6e}', flush=True)
