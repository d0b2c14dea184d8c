# scratch probe (not part of the product): ablate SpMV on the headline matrix by DATA variants
import sys, time, ctypes as C
import numpy as np, torch
from csr_amd import synth, _lib
from csr_amd._lib import lib, check, handle_t
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nrows = int(10_000_000*scale); nnz = int(200_000_000*scale)
dev = 'cuda'
t = time.time(); m = synth.powerlaw_csr(nrows, nrows, nnz, device=dev); torch.cuda.synchronize(); print('gen s', time.time()-t, flush=True)
x = synth.dense_vector(nrows, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
algbytes = nnz*12 + (nrows+1)*4 + nrows*8*2
g = torch.Generator(device=dev); g.manual_seed(20261003 + 1)
colperm = torch.randperm(nrows, generator=g, device=dev)
inv = torch.empty_like(colperm); inv[colperm] = torch.arange(nrows, device=dev)
cols = m['colinds']
variants = {
  'permuted(spec)': cols,
  'hot-contiguous': inv[cols.long()].to(torch.int32),
  'cached(mod1024)': (cols % 1024).to(torch.int32),
  'uniform-random': torch.randint(0, nrows, (nnz,), device=dev, dtype=torch.int32),
}
def timeit(h, algo, n=20):
    check(lib.csrk_set_spmv_algo(h, algo))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for name, cv in variants.items():
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, nrows, nnz, m['rowptrs'].data_ptr(), 0, cv.data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for algo, an in [(1,'merge'),(2,'vector')]:
        ms = timeit(h, algo)
        print(f'{name:18s} {an}: {ms:.3f} ms  {algbytes/ms/1e9:.2f} TB/s alg  {2*nnz/ms/1e6:.1f} GFLOP/s', flush=True)
    check(lib.csrk_free(h))
# reference points: plain copy bandwidth and x-permute cost
a = torch.empty(nnz, dtype=torch.float64, device=dev); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
print(f'torch copy 1.6GB: {ms:.3f} ms {2*nnz*8/ms/1e9:.2f} TB/s')
xp = torch.empty_like(x)
for _ in range(3): torch.index_select(x, 0, colperm, out=xp)
torch.cuda.synchronize(); e0.record()
for _ in range(10): torch.index_select(x, 0, colperm, out=xp)
e1.record(); torch.cuda.synchronize(); print(f'x permute (index_select 1e7): {e0.elapsed_time(e1)/10:.3f} ms')
