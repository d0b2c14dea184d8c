import ctypes as C, torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev='cuda'; n=10_000_000; nnz=200_000_000
sh = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
h = handle_t(0)
check(lib.csrk_create_device(n, n, nnz, sh['rowptrs'].data_ptr(), 0, sh['colinds'].data_ptr(), sh['values'].data_ptr(), 2, C.byref(h)))
for _ in range(3): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
st = (C.c_int64 * 25)(); check(lib.csrk_spmv_plan_stats(h, st, 25))
print('tier0 tiles', st[4], 'slots', st[4]*512, 'entries', st[10], 'overhead %.3f%%' % (100.0*(st[4]*512 - st[10])/st[10]), 'blocks', st[5])
