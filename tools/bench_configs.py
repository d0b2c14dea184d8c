#!/usr/bin/env python3
"""
The other BASELINE.json configs, one JSON line each: configs[2] (dense-panel SpMM), configs[4] (transpose + A B^T on the
MovieLens-25M shape) and unit_rows on the headline matrix are bench_secondary.py's functions -- the same ones bench.py
puts into its `secondary` block -- followed by SpGEMM throughput lines for more block sizes and a power-law A B.
Run under rocprofv3 by tools/collect_profiles_configs.sh; the lines go into profiles/rNN_configs.json.
    python tools/bench_configs.py [spmm|transpose|abt|unit_rows|spmv_f32|protocol|ab|all]
"""
import ctypes as C
import json
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth                                    # noqa: E402
from csr_amd._lib import lib, check, handle_t                # noqa: E402

dev = 'cuda'
what = sys.argv[1] if len(sys.argv) > 1 else 'all'


def timed(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def mk(m, nrows, ncols):
    h = handle_t(0)
    nnz = int(m['colinds'].numel())
    check(lib.csrk_create_device(nrows, ncols, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(),
                                 m['values'].data_ptr(), 2, C.byref(h)))
    return h


import bench_secondary as S                                 # noqa: E402

if what in ('spmv_f32', 'all'):
    print(json.dumps(S.spmv_f32(dev)), flush=True)
if what in ('protocol', 'all'):
    print(json.dumps(S.protocol(dev)), flush=True)
if what in ('unit_rows', 'all'):
    print(json.dumps(S.unit_rows(dev)), flush=True)
    check(lib.csrk_trim_cache())
if what in ('spmm', 'all'):
    print(json.dumps(S.spmm(dev)), flush=True)
    check(lib.csrk_trim_cache())
if what in ('transpose', 'abt', 'all'):
    nr, nc, nnz = S.ML_SHAPE
    m = S.ml_matrix(dev)
if what in ('transpose', 'all'):
    print(json.dumps(S.transpose(dev, m)), flush=True)
if what in ('abt', 'all'):
    print(json.dumps(S.abt(dev, m)), flush=True)

    def sub(r1):
        rp = m['rowptrs'][:r1 + 1].contiguous()
        e = int(rp[-1].item())
        hh = handle_t(0)
        check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
        return hh, rp, e

    # Throughput lines: intermediate products (what the symbolic pass counts: sum over A's entries of
    # the length of the B^T row they select = sum over columns of cntA * cntB), products/s, and the bytes of A, B^T and C
    def abt_block(ra, rb, reps=3):
        ha, rpa, ea = sub(ra)
        hb, rpb, eb = sub(rb)
        cnt_a = torch.bincount(m['colinds'][:ea].long(), minlength=nc).to(torch.float64)
        cnt_b = torch.bincount(m['colinds'][:eb].long(), minlength=nc).to(torch.float64)
        products = int(float((cnt_a * cnt_b).sum()))
        ts = []
        nn = C.c_int64()
        for _ in range(reps):
            cc = handle_t(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(lib.csrk_spgemm_abt(ha, hb, C.byref(cc)))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            check(lib.csrk_info(cc, None, None, C.byref(nn), None, None))
            vs_sum = None
            if _ == reps - 1:      # checksum property on the last one: sum of C == sum_col (1^T A)_col (1^T B)_col
                d_rp, d_ci, d_vs = C.c_void_p(), C.c_void_p(), C.c_void_p()
                check(lib.csrk_device_ptrs(cc, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))

                class _Dev:
                    pass
                dv = _Dev()
                dv.__cuda_array_interface__ = {'shape': (nn.value,), 'typestr': '<f8', 'data': (int(d_vs.value), False), 'version': 2}
                vs_sum = float(torch.as_tensor(dv, device=dev).sum())
            check(lib.csrk_free(cc))
        ca = torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, m['colinds'][:ea].long(), m['values'][:ea])
        cb = torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, m['colinds'][:eb].long(), m['values'][:eb])
        want = float((ca * cb).sum())
        ms_b = min(ts[1:]) if reps > 1 else ts[0]
        abc = (ea + eb) * 12 + (ra + rb + 2) * 4 + nn.value * 12 + (ra + 1) * 4
        print(json.dumps({'config': f'mult_abt ({ra} x {nc}) x ({rb} x {nc})^T, ML25M-shape rows: throughput', 'ms': round(ms_b, 2),
                          'intermediate_products': products, 'products_per_s': round(products / ms_b * 1e3, -6),
                          'flops_2_per_product_gflops': round(2 * products / ms_b / 1e6, 1),
                          'product_nnz': nn.value, 'output_bytes': nn.value * 12 + (ra + 1) * 4,
                          'a_bt_c_bytes': abc, 'a_bt_c_GBs': round(abc / ms_b / 1e6, 1),
                          'checksum_rel_err': abs(vs_sum - want) / abs(want)}), flush=True)
        check(lib.csrk_free(ha)); check(lib.csrk_free(hb))
    abt_block(500, 5000)
    abt_block(2000, 20000)
    abt_block(20000, 20000, reps=2)

if what in ('ab', 'all'):
    # power-law A . B, 1M x 1M, nnz 5e6 each (the expand-sort-compress path's workload)
    n1, nnz1 = 1_000_000, 5_000_000
    a = synth.powerlaw_csr(n1, n1, nnz1, device=dev, max_degree=2000)
    b = synth.powerlaw_csr(n1, n1, nnz1, device=dev, max_degree=2000, seed=7)
    ha, hb = mk(a, n1, n1), mk(b, n1, n1)
    len_b = (b['rowptrs'][1:] - b['rowptrs'][:-1]).to(torch.float64)
    products = int(float(len_b[a['colinds'].long()].sum()))
    ts = []
    nn = C.c_int64()
    for _ in range(3):
        cc = handle_t(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib.csrk_spgemm_ab(ha, hb, C.byref(cc)))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        check(lib.csrk_info(cc, None, None, C.byref(nn), None, None))
        check(lib.csrk_free(cc))
    ms_b = min(ts[1:])
    abc = 2 * (nnz1 * 12 + (n1 + 1) * 4) + nn.value * 12 + (n1 + 1) * 4
    print(json.dumps({'config': 'mult_ab power-law 1M x 1M, nnz 5e6 each: throughput', 'ms': round(ms_b, 2),
                      'intermediate_products': products, 'products_per_s': round(products / ms_b * 1e3, -6),
                      'flops_2_per_product_gflops': round(2 * products / ms_b / 1e6, 1), 'product_nnz': nn.value,
                      'output_bytes': nn.value * 12 + (n1 + 1) * 4, 'a_b_c_bytes': abc, 'a_b_c_GBs': round(abc / ms_b / 1e6, 1)}),
          flush=True)
    check(lib.csrk_free(ha)); check(lib.csrk_free(hb))
