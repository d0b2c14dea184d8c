# scratch: csrk_pick_rows timing (MovieLens-25M shape, 100k picked rows; headline matrix, 2M picked rows)
import ctypes as C, time, numpy as np, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev = 'cuda'
def run(name, nr, nc, nnz, k, **kw):
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, **kw)
    h = handle_t(0)
    check(lib.csrk_create_device(nr, nc, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    rows = np.random.default_rng(1).integers(0, nr, size=k).astype(np.int32)
    best, nn = 1e9, C.c_int64()
    for i in range(5):
        o = handle_t(0); torch.cuda.synchronize(); t0 = time.perf_counter()
        check(lib.csrk_pick_rows(h, rows.ctypes.data_as(C.c_void_p), k, 1, C.byref(o))); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        check(lib.csrk_info(o, None, None, C.byref(nn), None, None)); check(lib.csrk_free(o))
    gb = (nn.value * 12 * 2 + k * 16) / 1e9
    print(f'{name}: pick {k} rows -> nnz {nn.value}: {best:.3f} ms wall (H2D of the row list included), {gb / best * 1e3:.0f} GB/s of read+write', flush=True)
    check(lib.csrk_free(h))
run('ML25M shape', 162_541, 59_047, 25_000_095, 100_000, alpha=0.9, max_degree=7000)
run('headline', 10_000_000, 10_000_000, 200_000_000, 2_000_000)
