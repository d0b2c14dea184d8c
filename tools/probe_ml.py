import ctypes as C, torch, json, os
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'
def bench(name, m, nrows, ncols, **env):
    for k,v in env.items(): os.environ[k]=str(v)
    rp, ci, vs = m['rowptrs'], m['colinds'], m['values']
    nnz = int(ci.numel())
    x = synth.dense_vector(ncols, device=dev); y = torch.empty(nrows, dtype=torch.float64, device=dev)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, ncols, nnz, rp.data_ptr(), 0, ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
    for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    check(lib.csrk_spmv_profile_begin(h, 20)); e0.record()
    for _ in range(20): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    n=C.c_int(); k=(C.c_float*3)(); check(lib.csrk_spmv_profile_end(h, C.byref(n), k))
    alg = nnz*12 + (nrows+1)*4 + nrows*8 + ncols*8
    st=(C.c_int64*16)(); check(lib.csrk_spmv_plan_stats(h, st, 16))
    print(f'{name:34s} {ms:.3f} ms  {alg/ms/1e9:.2f} TB/s  light {k[0]:.3f} t0 {k[1]:.3f} t1 {k[2]:.3f}  panels rows {st[2]} t0 pairs {st[9]} t1 pairs {st[12]}', flush=True)
    check(lib.csrk_free(h))
m = synth.powerlaw_csr(162_541, 59_047, 25_000_095, device=dev, alpha=0.9, max_degree=7000)
bench('ML25M split on', m, 162_541, 59_047, CSRK_SPMV_HEAVY_SPLIT=1)
bench('ML25M split off', m, 162_541, 59_047, CSRK_SPMV_HEAVY_SPLIT=0)
bench('ML25M tier1 off', m, 162_541, 59_047, CSRK_SPMV_HEAVY_SPLIT=1, CSRK_TIERB_MIN=0)
m = synth.powerlaw_csr(2_000_000, 2_000_000, 50_000_000, device=dev, max_degree=250_000)
os.environ['CSRK_TIERB_MIN']='128'
bench('2M split on', m, 2_000_000, 2_000_000, CSRK_SPMV_HEAVY_SPLIT=1)
bench('2M split off', m, 2_000_000, 2_000_000, CSRK_SPMV_HEAVY_SPLIT=0)
