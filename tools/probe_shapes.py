# scratch: SpMV across matrix families (robustness table for DESIGN.md)
import ctypes as C, torch, json
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
def bench(name, rp, ci, vs, nrows, ncols):
    nnz = int(ci.numel())
    x = synth.dense_vector(ncols, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, ncols, nnz, rp.data_ptr(), 0, ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    alg = nnz*12 + (nrows+1)*4 + nrows*8 + ncols*8
    st=(C.c_int64*16)(); check(lib.csrk_spmv_plan_stats(h, st, 16))
    print(json.dumps({'matrix': name, 'nrows': nrows, 'nnz': nnz, 'ms': round(ms,3), 'GFLOPs': round(2*nnz/ms/1e6,1), 'alg_TBs': round(alg/ms/1e9,2), 'rows_in_panels': int(st[2])}), flush=True)
    check(lib.csrk_free(h))
n=10_000_000
# uniform degree 20, uniform random columns (sorted within rows)
g=torch.Generator(device=dev); g.manual_seed(1)
deg=20; nnz=n*deg
rp=(torch.arange(n+1, device=dev, dtype=torch.int64)*deg).to(torch.int32)
cols=torch.randint(0, n, (nnz,), generator=g, device=dev, dtype=torch.int64)
key=torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(deg)*n + cols
key,_=torch.sort(key); ci=(key % n).to(torch.int32); del key, cols
vs=synth.hash_uniform(torch.arange(nnz, device=dev), 7, 2)*2-1
bench('uniform degree 20, random columns', rp, ci, vs, n, n)
# banded: 20 entries around the diagonal
off=torch.arange(-10, 10, device=dev, dtype=torch.int64)
ci=((torch.arange(n, device=dev, dtype=torch.int64)[:,None] + off[None,:]).clamp_(0, n-1)).reshape(-1).to(torch.int32)
bench('banded, 20 around the diagonal', rp, ci, vs, n, n)
del ci, vs, rp
m = synth.powerlaw_csr(162_541, 59_047, 25_000_095, device=dev, alpha=0.9, max_degree=7000)
bench('MovieLens-25M shape', m['rowptrs'], m['colinds'], m['values'], 162_541, 59_047)
m = synth.powerlaw_csr(2_000_000, 2_000_000, 50_000_000, device=dev, max_degree=250_000)
bench('power-law 2M x 2M nnz 5e7 (config 3 A)', m['rowptrs'], m['colinds'], m['values'], 2_000_000, 2_000_000)
m = synth.powerlaw_csr(10_000_000, 10_000_000, 200_000_000, device=dev)
bench('power-law 10M x 10M nnz 2e8 (headline)', m['rowptrs'], m['colinds'], m['values'], 10_000_000, 10_000_000)
