#!/usr/bin/env python3
"""
Measurements for BASELINE.json configs[2] (dense-panel SpMM, A 2M x 2M nnz 5e7, k = 64) and
configs[4] (transpose + A B^T on a MovieLens-25M-shaped 162541 x 59047 matrix, nnz 2.5e7), with
size-independent parity properties at full size and the CPU oracle (sequential restatement of the
reference loops, 1 core) timed beside each on a bounded sample.  Not the driver's bench (bench.py is); the
numbers go into DESIGN.md and profiles/r01_configs.json.
    python tools/bench_configs.py [spmm|transpose|abt|ab|all]
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


if what in ('spmm', 'all'):
    n, nnz, k = 2_000_000, 50_000_000, 64
    m = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=250_000)
    h = mk(m, n, n)
    B = synth.dense_vector(n * k, device=dev, stream=7).view(n, k)
    Cm = torch.empty(n, k, dtype=torch.float64, device=dev)
    ms = timed(lambda: check(lib.csrk_spmm_dense_device(h, B.data_ptr(), k, k, Cm.data_ptr(), k, None)))
    alg = nnz * 12 + (n + 1) * 4 + 2 * n * k * 8
    # linearity property: A (B1 + 2 B2) == A B1 + 2 A B2 (to rounding), and column 0 equals SpMV with B[:,0]
    y = torch.empty(n, dtype=torch.float64, device=dev)
    x0 = B[:, 0].contiguous()
    check(lib.csrk_spmv_device(h, x0.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    absA = (m['values'].abs())
    habs = handle_t(0)
    check(lib.csrk_create_device(n, n, nnz, m['rowptrs'].data_ptr(), 0, m['colinds'].data_ptr(), absA.data_ptr(), 2, C.byref(habs)))
    bound = torch.empty(n, dtype=torch.float64, device=dev)
    xa = x0.abs().contiguous()
    check(lib.csrk_spmv_device(habs, xa.data_ptr(), bound.data_ptr(), None))
    torch.cuda.synchronize()
    err = float(((Cm[:, 0] - y).abs() / (bound + 1e-300)).max())
    # CPU baseline: the oracle's dense-panel product on the first rows of A holding ~2e6 entries
    from oracle import oracle as O
    rp_h = m['rowptrs'].cpu().numpy()
    r_s = int(np.searchsorted(rp_h, 2_000_000))
    e_s = int(rp_h[r_s])
    ci_h, vs_h = m['colinds'][:e_s].cpu().numpy(), m['values'][:e_s].cpu().numpy()
    B_h = B.cpu().numpy()
    t0 = time.perf_counter()
    C_h = O.spmm_dense(r_s, rp_h[:r_s + 1], ci_h, vs_h, B_h)
    t_cpu = time.perf_counter() - t0
    samp_err = float(np.max(np.abs(C_h - Cm[:r_s].cpu().numpy())) / max(1e-300, float(np.max(np.abs(C_h)))))
    print(json.dumps({'config': 'spmm_dense 2Mx2M nnz5e7 k64 f64', 'ms': round(ms, 3), 'gflops': round(2 * nnz * k / ms / 1e6, 1),
                      'algorithmic_GB': round(alg / 1e9, 3), 'achieved_GBs_alg': round(alg / ms / 1e6, 1),
                      'col0_vs_spmv_max_err_over_bound': err,
                      'cpu_baseline': {'gflops': round(2 * e_s * k / t_cpu / 1e9, 3), 'cores': 1, 'kind': 'port',
                                       'sample': f'first {r_s} rows ({e_s} entries), {t_cpu:.2f} s',
                                       'gpu_vs_oracle_max_rel_err_on_sample': samp_err}}), flush=True)
    check(lib.csrk_free(h)); check(lib.csrk_free(habs))
    del m, B, Cm

if what in ('transpose', 'abt', 'all'):
    nr, nc, nnz = 162_541, 59_047, 25_000_095
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
    # MovieLens-like ratings 0.5 .. 5.0
    m['values'] = (torch.floor((m['values'] + 1.0) * 5.0).clamp_(0, 9) + 1.0) * 0.5
    h = mk(m, nr, nc)

if what in ('transpose', 'all'):
    outs = []

    def tr(keep=False):
        t = handle_t(0)
        check(lib.csrk_transpose(h, 1, C.byref(t)))
        if keep:
            outs.append(t)
        else:
            check(lib.csrk_free(t))      # what CSR.transpose does (from_handle, then release)
    tr()
    tr()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    tr(keep=True)
    # properties: transpose(transpose(A)) == A bit for bit (rows are column-sorted); rowptrs of A^T = column histogram
    t = outs[-1]
    tt = handle_t(0)
    check(lib.csrk_transpose(t, 1, C.byref(tt)))
    rp2 = np.empty(nr + 1, np.int32); ci2 = np.empty(nnz, np.int32); vs2 = np.empty(nnz)
    check(lib.csrk_export(tt, rp2.ctypes.data_as(C.c_void_p), ci2.ctypes.data_as(C.c_void_p), vs2.ctypes.data_as(C.c_void_p)))
    ok = (np.array_equal(rp2, m['rowptrs'].cpu().numpy()) and np.array_equal(ci2, m['colinds'].cpu().numpy())
          and np.array_equal(vs2, m['values'].cpu().numpy()))
    rpt = np.empty(nc + 1, np.int32)
    check(lib.csrk_export(t, rpt.ctypes.data_as(C.c_void_p), None, None))
    hist_ok = np.array_equal(np.diff(rpt), np.bincount(m['colinds'].cpu().numpy(), minlength=nc))
    alg = 4 * nnz + (4 + 8) * nnz + (4 + 8) * nnz + (nr + nc + 2) * 4
    # CPU baseline + full-size bit-exact check: the oracle's transpose of the whole matrix
    from oracle import oracle as O
    rp_h, ci_h, vs_h = m['rowptrs'].cpu().numpy(), m['colinds'].cpu().numpy(), m['values'].cpu().numpy()
    t0 = time.perf_counter()
    _, _, orp, oci, ovs = O.transpose(nr, nc, rp_h, ci_h, vs_h)
    t_cpu = time.perf_counter() - t0
    ci_t = np.empty(nnz, np.int32); vs_t = np.empty(nnz)
    check(lib.csrk_export(t, rpt.ctypes.data_as(C.c_void_p), ci_t.ctypes.data_as(C.c_void_p), vs_t.ctypes.data_as(C.c_void_p)))
    exact = bool(np.array_equal(rpt, orp) and np.array_equal(ci_t, oci) and np.array_equal(vs_t, ovs))
    print(json.dumps({'config': 'transpose ML25M-shape 162541x59047 nnz 25000095 (wall time per csrk_transpose call, result arrays from the pool)',
                      'ms': round(ms, 3), 'algorithmic_GB': round(alg / 1e9, 3), 'achieved_GBs_alg': round(alg / ms / 1e6, 1),
                      'double_transpose_bit_exact': bool(ok), 'rowptrs_match_histogram': bool(hist_ok),
                      'bit_exact_vs_oracle_full_size': exact,
                      'cpu_baseline': {'ms': round(t_cpu * 1e3, 1), 'GBs_alg': round(alg / t_cpu / 1e9, 2), 'cores': 1, 'kind': 'port',
                                       'sample': 'the whole matrix, one pass'}}), flush=True)
    for o in outs:
        check(lib.csrk_free(o))
    check(lib.csrk_free(tt))

if what in ('abt', 'all'):
    # A_blk B^T with A_blk = first 2000 rows (users), B = first 20000 rows: product rows are dense-ish
    def sub(r1):
        rp = m['rowptrs'][:r1 + 1].contiguous()
        e = int(rp[-1].item())
        hh = handle_t(0)
        check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
        return hh, rp, e
    ha, rpa, ea = sub(2000)
    hb, rpb, eb = sub(20000)
    c = handle_t(0)
    t0 = time.perf_counter()
    check(lib.csrk_spgemm_abt(ha, hb, C.byref(c)))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    nrc, ncc, nnzc = C.c_int32(), C.c_int32(), C.c_int64()
    check(lib.csrk_info(c, C.byref(nrc), C.byref(ncc), C.byref(nnzc), None, None))
    # checksum property: sum of all entries of A B^T == (1^T A) . (1^T B) summed over columns
    rpc = np.empty(nrc.value + 1, np.int32); cic = np.empty(nnzc.value, np.int32); vsc = np.empty(nnzc.value)
    check(lib.csrk_export(c, rpc.ctypes.data_as(C.c_void_p), cic.ctypes.data_as(C.c_void_p), vsc.ctypes.data_as(C.c_void_p)))
    ca = torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, m['colinds'][:ea].long(), m['values'][:ea])
    cb = torch.zeros(nc, dtype=torch.float64, device=dev).index_add_(0, m['colinds'][:eb].long(), m['values'][:eb])
    want = float((ca * cb).sum())
    got = float(vsc.sum())
    # warm call, and the oracle (transpose + mult_ab, the reference's mult_abt) on the same block
    c2 = handle_t(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib.csrk_spgemm_abt(ha, hb, C.byref(c2)))
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) * 1e3
    check(lib.csrk_free(c2))
    from oracle import oracle as O
    ci_h, vs_h = m['colinds'].cpu().numpy(), m['values'].cpu().numpy()
    t0 = time.perf_counter()
    tnr, tnc, trp, tci, tvs = O.transpose(20000, nc, rpb.cpu().numpy(), ci_h[:eb], vs_h[:eb])
    r = O.mult_ab((2000, nc, rpa.cpu().numpy(), ci_h[:ea], vs_h[:ea]), (tnr, tnc, trp, tci, tvs))
    t_cpu = time.perf_counter() - t0
    print(json.dumps({'config': 'mult_abt (2000 x 59047) x (20000 x 59047)^T, ML25M-shape rows', 'ms_first_call': round(ms, 1),
                      'ms': round(ms2, 1), 'product_nnz': nnzc.value, 'checksum_rel_err': abs(got - want) / abs(want),
                      'cols_sorted': bool(all(np.all(np.diff(cic[rpc[i]:rpc[i + 1]]) > 0) for i in range(0, 2000, 97))),
                      'cpu_baseline': {'ms': round(t_cpu * 1e3, 1), 'cores': 1, 'kind': 'port', 'product_nnz': int(len(r[3])),
                                       'sample': 'the same block, one pass'}}), flush=True)
    check(lib.csrk_free(c)); check(lib.csrk_free(ha)); check(lib.csrk_free(hb))

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
