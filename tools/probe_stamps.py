# scratch (GPU box): per-phase cycle shares of the light-stream kernel's tile loop (diagnostic build with -DCSRK_LS_STAMPS)
#   CSRK_LIBRARY=$PWD/csr_amd/libcsrk_stamps.so python tools/probe_stamps.py
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
n, nnz = 10_000_000, 200_000_000
dev = torch.device('cuda', 0)
m = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
h = handle_t(0)
check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
for _ in range(12):
    check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.csrk_debug_ls_stamps.argtypes = [C.c_void_p, C.c_int]
check(lib.csrk_debug_ls_stamps(buf.ctypes.data_as(C.c_void_p), buf.size))
b = buf.reshape(4096, 8).astype(np.float64)
tiles = b[:, 7]
names = ['0 top wait + issue loads', '1 row-start flags + exscan', '2 x values arrive, products', '3 lane sums, segscan, carries, s_out', '4 output loop (y stores)', '5 drain stores/prefetch']
tot = b[:, :6].sum(axis=1)
print('waves', (tiles > 0).sum(), 'tiles/wave', tiles.mean(), 'cycles/tile', (tot / np.maximum(tiles, 1)).mean())
for i, nm in enumerate(names):
    print(f'  {nm:40s} {b[:, i].sum() / tot.sum() * 100:5.1f} %   {(b[:, i] / np.maximum(tiles, 1)).mean():8.0f} cycles/tile')
