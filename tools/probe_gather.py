# scratch probe: SpMV time vs where the gathered x entries live (L1 / L2 / MALL / HBM-ish)
import sys, ctypes as C, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
nrows = 10_000_000; nnz = 200_000_000; dev='cuda'
m = synth.powerlaw_csr(nrows, nrows, nnz, device=dev)
x = synth.dense_vector(nrows, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
algbytes = nnz*12 + (nrows+1)*4 + nrows*8*2
rnd = torch.randint(0, 2**31-1, (nnz,), device=dev, dtype=torch.int64)
def run(cv, name):
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, nrows, nnz, m['rowptrs'].data_ptr(), 0, cv.data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    print(f'{name:28s} {ms:.3f} ms  {algbytes/ms/1e9:.2f} TB/s alg', flush=True)
    check(lib.csrk_free(h))
for bits, label in [(10,'8KB (L1)'),(13,'64KB'),(16,'512KB (L2)'),(18,'2MB (L2)'),(19,'4MB'),(20,'8MB'),(22,'32MB (MALL)'),(23,'64MB (MALL)')]:
    cv = (rnd & ((1<<bits)-1)).to(torch.int32)
    run(cv, f'random in {label}')
cv = (torch.arange(nnz, device=dev) % nrows).to(torch.int32)
run(cv, 'sequential x')
