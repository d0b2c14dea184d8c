#!/usr/bin/env python3
"""
scratch (GPU box): time the headline SpMV under several environment settings in ONE process (the matrix is generated
once; libcsrk reads its CSRK_* knobs with getenv when a handle's plan is built).
    python tools/sweep_inproc.py "A=1 B=2" "A=3" ...        ('' = defaults)
Each configuration: new handle, 2 calls (plan), 30 warm-up + 200 timed steps, kernel breakdown from the library's own
event pairs on every 10th step, and a bit-for-bit comparison of y with the first configuration's.
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth                                    # noqa: E402
from csr_amd._lib import lib, check, handle_t                # noqa: E402

scale = float(os.environ.get('SWEEP_SCALE', '1.0'))
steps = int(os.environ.get('SWEEP_STEPS', '200'))
n = int(10_000_000 * scale)
nnz = int(200_000_000 * scale)
dev = torch.device('cuda', 0)
m = synth.powerlaw_csr(n, n, nnz, device=dev)
x = synth.dense_vector(n, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)
ref = None
base_env = dict(os.environ)
for cfg in sys.argv[1:] or ['']:
    for k in [k for k in os.environ if k.startswith('CSRK_') and k not in base_env]:
        del os.environ[k]
    for kv in cfg.split():
        k, _, v = kv.partition('=')
        os.environ[k] = v
    h = handle_t(0)
    check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(h)))
    t0 = time.perf_counter()
    for _ in range(2):
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    for _ in range(30):
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    check(lib.csrk_spmv_profile_every(h, 10))
    check(lib.csrk_spmv_profile_begin(h, steps // 10 + 2))
    t0 = time.perf_counter()
    for _ in range(steps):
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    nrec, k4 = C.c_int(0), (C.c_float * 4)()
    check(lib.csrk_spmv_profile_end4(h, C.byref(nrec), k4))
    st = (C.c_int64 * 27)()
    check(lib.csrk_spmv_plan_stats(h, st, 27))
    same = None
    if ref is None:
        ref = y.clone()
    else:
        same = bool(torch.equal(ref, y))
        if not same:
            same = f'maxdiff {float((ref - y).abs().max()):.3e}'
    print(f'[{cfg or "defaults"}] {ms:.4f} ms/step  light {k4[0]:.4f} acc {k4[1]:.4f} t1 {k4[2]:.4f} stage {k4[3]:.4f} | '
          f'pack {int(st[16])} cold {int(st[24])} t0 {int(st[10])} t1 {int(st[13])} light {int(st[3])} round {int(st[26])} | plan+first {plan_s*1e3:.0f} ms | same_bits {same}',
          flush=True)
    check(lib.csrk_free(h))
