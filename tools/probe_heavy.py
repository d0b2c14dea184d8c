import sys, ctypes as C, torch, os
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
nrows = 10_000_000; nnz = 200_000_000; dev='cuda'
m = synth.powerlaw_csr(nrows, nrows, nnz, device=dev)
x = synth.dense_vector(nrows, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
def run(name, **env):
    for k,v in env.items(): os.environ[k]=str(v)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, nrows, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    check(lib.csrk_spmv_profile_begin(h, 10)); e0.record()
    for _ in range(10): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    n=C.c_int(); k=C.c_float(); check(lib.csrk_spmv_profile_end(h, C.byref(n), C.byref(k)))
    print(f'{name:34s} total {ms:.3f} ms  merge {k.value:.3f}  heavy+rest {ms-k.value:.3f}', flush=True)
    check(lib.csrk_free(h))
for W in (32768, 65536, 131072, 262144, 524288, 1048576):
    run(f'W={W}', CSRK_HEAVY_BLOCK=W, CSRK_HEAVY_PIECE=1024, CSRK_HEAVY_MIN=2048)
for M in (1024, 4096):
    run(f'W=262144 min={M}', CSRK_HEAVY_BLOCK=262144, CSRK_HEAVY_PIECE=1024, CSRK_HEAVY_MIN=M)
