#!/usr/bin/env python3
# scratch (GPU box): does the accumulator kernel's time depend on WHERE the plan's arrays were allocated?  Several handles on
# the same matrix alive at once (each with its own plan allocations), each timed in turn, twice.
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
n, nnz = 10_000_000, 200_000_000
dev = torch.device('cuda', 0)
m = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)
hs = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    h = handle_t(0)
    check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    for _ in range(3):
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    hs.append(h)
torch.cuda.synchronize()
for rnd in range(2):
    for k, h in enumerate(hs):
        for _ in range(20):
            check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        torch.cuda.synchronize()
        check(lib.csrk_spmv_profile_every(h, 5))
        check(lib.csrk_spmv_profile_begin(h, 50))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        e1.record()
        torch.cuda.synchronize()
        nrec, k4 = C.c_int(0), (C.c_float * 4)()
        check(lib.csrk_spmv_profile_end4(h, C.byref(nrec), k4))
        print(f'round {rnd} handle {k}: {e0.elapsed_time(e1) / 200:.4f} ms/step  light {k4[0]:.4f} acc {k4[1]:.4f} t1 {k4[2]:.4f} stage {k4[3]:.4f}', flush=True)
