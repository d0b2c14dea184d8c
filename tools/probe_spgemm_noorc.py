import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, time, numpy as np, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
def mk(m, nr, nc):
    h = handle_t(0); nnz=int(m['colinds'].numel())
    check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    return h
for (n, nnz, name) in [(200_000, 2_000_000, 'powerlaw 200k nnz 2e6'), (1_000_000, 5_000_000, 'powerlaw 1M nnz 5e6')]:
    a = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=2000)
    b = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=2000, seed=7)
    ha, hb = mk(a, n, n), mk(b, n, n)
    for i in range(3):
        c = handle_t(0); torch.cuda.synchronize(); t0=time.perf_counter()
        check(lib.csrk_spgemm_ab(ha, hb, C.byref(c))); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
        nn = C.c_int64(); check(lib.csrk_info(c, None, None, C.byref(nn), None, None))
        check(lib.csrk_free(c))
    print(f'{name}: {dt:.1f} ms, product nnz {nn.value}', flush=True)
    check(lib.csrk_free(ha)); check(lib.csrk_free(hb))
