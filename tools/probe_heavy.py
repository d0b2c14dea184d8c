import sys, ctypes as C, torch, os
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
nrows = 10_000_000; nnz = 200_000_000; dev='cuda'
m = synth.powerlaw_csr(nrows, nrows, nnz, device=dev)
x = synth.dense_vector(nrows, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
ref = None
def run(name, **env):
    global ref
    for k,v in env.items(): os.environ[k]=str(v)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, nrows, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    check(lib.csrk_spmv_profile_begin(h, 20)); e0.record()
    for _ in range(20): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    n=C.c_int(); k=(C.c_float*3)(); check(lib.csrk_spmv_profile_end(h, C.byref(n), k))
    st=(C.c_int64*16)(); check(lib.csrk_spmv_plan_stats(h, st, 16))
    yy = y.clone()
    if ref is None: ref = yy
    print(f'{name:22s} total {ms:.3f} ms light {k[0]:.3f} t0 {k[1]:.3f} t1 {k[2]:.3f} other {ms-k[0]-k[1]-k[2]:.3f} maxdiff {float((yy-ref).abs().max()):.1e} cut_rows {st[2]} light_nnz {st[3]} t1: rows {st[11]} pairs {st[12]} nnz {st[13]}', flush=True)
    check(lib.csrk_free(h))
run('streams off', CSRK_SPMV_STREAMS=0)
run('streams on', CSRK_SPMV_STREAMS=1)
run('streams off', CSRK_SPMV_STREAMS=0)
run('streams on', CSRK_SPMV_STREAMS=1)
