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
    n=C.c_int(); k=(C.c_float*2)(); check(lib.csrk_spmv_profile_end(h, C.byref(n), k))
    st=(C.c_int64*12)(); check(lib.csrk_spmv_plan_stats(h, st, 12))
    yy = y.clone()
    if ref is None: ref = yy
    print(f'{name:26s} total {ms:.3f} ms  light {k[0]:.3f}  heavy {k[1]:.3f}  other {ms-k[0]-k[1]:.3f}  maxdiff {float((yy-ref).abs().max()):.2e} heavy_rows {st[2]} light_nnz {st[3]}', flush=True)
    check(lib.csrk_free(h))
for mn in (1536, 2048, 3072, 4096, 8192, 16384):
    run(f'panel min={mn}', CSRK_HEAVY_MIN=mn, CSRK_PANEL_TPW=8)
run('panel min=2048 tpw=6', CSRK_HEAVY_MIN=2048, CSRK_PANEL_TPW=6)
run('panel min=2048 tpw=12', CSRK_HEAVY_MIN=2048, CSRK_PANEL_TPW=12)
