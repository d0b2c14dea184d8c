import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
m = synth.powerlaw_csr(200_000, 50_000, 4_000_000, device=dev, alpha=0.9, max_degree=3000)
x = synth.dense_vector(50_000, device=dev)
def mk(r1):
    rp = m['rowptrs'][:r1+1].to(torch.int32).contiguous(); e=int(rp[-1]); h=handle_t(0)
    check(lib.csrk_create_device(r1, 50_000, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    return h, rp
def cycle():
    ha, ra = mk(300); hb, rb = mk(3000); h, r = mk(200_000)
    c = handle_t(0); check(lib.csrk_spgemm_abt(ha, hb, C.byref(c))); check(lib.csrk_free(c))
    t = handle_t(0); check(lib.csrk_transpose(h, 1, C.byref(t))); check(lib.csrk_free(t))
    y = torch.empty(200_000, dtype=torch.float64, device=dev)
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    for q in (ha, hb, h): check(lib.csrk_free(q))
for i in range(5): cycle()
check(lib.csrk_trim_cache()); torch.cuda.synchronize(); torch.cuda.empty_cache()
f0 = torch.cuda.mem_get_info()[0]
for i in range(300): cycle()
check(lib.csrk_trim_cache()); torch.cuda.synchronize(); torch.cuda.empty_cache()
f1 = torch.cuda.mem_get_info()[0]
print('free before %.1f MB after %.1f MB: change %.1f MB over 300 cycles' % (f0/1e6, f1/1e6, (f1-f0)/1e6))
