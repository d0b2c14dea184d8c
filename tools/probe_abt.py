import ctypes as C, time, numpy as np, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
from oracle import oracle as O
dev='cuda'
nr, nc, nnz = 162_541, 59_047, 25_000_095
m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
def sub(r1):
    rp = m['rowptrs'][:r1 + 1].contiguous(); e = int(rp[-1].item()); hh = handle_t(0)
    check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
    return hh, rp, e
for ra, rb in ((500, 5000), (2000, 20000)):
    ha, rpa, ea = sub(ra); hb, rpb, eb = sub(rb)
    for i in range(2):
        c = handle_t(0); torch.cuda.synchronize(); t0=time.perf_counter()
        check(lib.csrk_spgemm_abt(ha, hb, C.byref(c))); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)*1e3
        nn=C.c_int64(); check(lib.csrk_info(c, None, None, C.byref(nn), None, None)); check(lib.csrk_free(c))
    print(f'A[{ra}] B[{rb}]^T: GPU {dt:.1f} ms, nnz {nn.value}', flush=True)
    ci=m['colinds'].cpu().numpy(); vs=m['values'].cpu().numpy()
    A=(ra,nc,rpa.cpu().numpy(),ci[:ea],vs[:ea]); 
    t0=time.perf_counter()
    tnr,tnc,trp,tci,tvs = O.transpose(rb,nc,rpb.cpu().numpy(),ci[:eb],vs[:eb])
    r=O.mult_ab(A,(tnr,tnc,trp,tci,tvs)); print(f'   oracle: {(time.perf_counter()-t0)*1e3:.1f} ms nnz {len(r[3])}', flush=True)
    check(lib.csrk_free(ha)); check(lib.csrk_free(hb))
