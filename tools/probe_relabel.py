# scratch: potential of popularity-ordered column labels (hot columns contiguous) with the tiered SpMV
import ctypes as C, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
nrows = 10_000_000; nnz = 200_000_000; dev='cuda'
m = synth.powerlaw_csr(nrows, nrows, nnz, device=dev)
x = synth.dense_vector(nrows, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
def run(name, ci):
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, nrows, nnz, m['rowptrs'].data_ptr(), 0, ci.data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    check(lib.csrk_spmv_profile_begin(h, 20)); e0.record()
    for _ in range(20): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    n=C.c_int(); k=(C.c_float*3)(); check(lib.csrk_spmv_profile_end(h, C.byref(n), k))
    print(f'{name:28s} total {ms:.3f} ms light {k[0]:.3f} t0 {k[1]:.3f} t1 {k[2]:.3f}', flush=True)
    check(lib.csrk_free(h))
run('spec (permuted columns)', m['colinds'])
# relabel: column -> popularity rank (inverse of the generator's permutation), then re-sort inside rows
g = torch.Generator(device=dev); g.manual_seed(20261003 + 1)
colperm = torch.randperm(nrows, generator=g, device=dev)
inv = torch.empty_like(colperm); inv[colperm] = torch.arange(nrows, device=dev)
rows = torch.repeat_interleave(torch.arange(nrows, device=dev, dtype=torch.int64), (m['rowptrs'][1:] - m['rowptrs'][:-1]).long())
key = rows * nrows + inv[m['colinds'].long()]
del rows
key, _ = torch.sort(key)
ci2 = (key % nrows).to(torch.int32); del key
run('popularity-ordered columns', ci2)
# cost of the per-call x permutation the relabelling would need
xp = torch.empty_like(x)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): torch.index_select(x, 0, colperm, out=xp)
e1.record(); torch.cuda.synchronize(); print(f'x permutation (torch index_select): {e0.elapsed_time(e1)/10:.3f} ms')
