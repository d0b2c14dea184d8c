# scratch (GPU box): the 500 x 5000^T block of the MovieLens-shaped matrix: wall time per csrk_spgemm_abt call
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
m = synth.movielens_like(device='cuda'); nc = m['ncols']
def sub(r1):
    rp = m['rowptrs'][:r1 + 1].contiguous(); e = int(rp[-1].item()); hh = handle_t(0)
    check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
    return hh, rp
ha, ka = sub(500); hb, kb = sub(5000)
for i in range(5):
    c = handle_t(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.csrk_spgemm_abt(ha, hb, C.byref(c))); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    check(lib.csrk_free(c))
print(os.environ.get('CSRK_LIBRARY', 'default').split('/')[-1], f'{ms:.2f} ms')
