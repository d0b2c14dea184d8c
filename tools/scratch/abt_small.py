import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev = 'cuda'
nr, nc, nnz = 162_541, 59_047, 25_000_095
m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
m['values'] = (torch.floor((m['values'] + 1.0) * 5.0).clamp_(0, 9) + 1.0) * 0.5
def sub(r1):
    rp = m['rowptrs'][:r1 + 1].contiguous(); e = int(rp[-1].item()); hh = handle_t(0)
    check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
    return hh, rp
ha, rpa = sub(500); hb, rpb = sub(5000)
ts = []
for i in range(30):
    c = handle_t(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.csrk_spgemm_abt(ha, hb, C.byref(c))); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    check(lib.csrk_free(c))
print('abt 500x5000: min %.3f ms median %.3f ms' % (min(ts), sorted(ts)[len(ts)//2]))
