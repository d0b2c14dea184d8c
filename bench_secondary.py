"""
The other BASELINE.json configurations, measured the way bench.py measures the headline SpMV: through the libcsrk
C ABI with inputs resident in HBM, each with its algorithmic bytes, the fraction of the 8 TB/s
roofline, a parity flag against the CPU oracle, and the oracle timed on a stated sample as `cpu_baseline` (kind
"port", 1 core).  bench.py puts the dicts these functions return into the `secondary` block of its JSON line
(`--no-secondary` skips it); tools/bench_configs.py prints the same dicts one per line for profiles/rNN_configs.json.

  spmm       configs[2]: dense-panel SpMM, A 2M x 2M nnz 5e7 (power-law) x B 2M x 64 f64 (csrk_spmm_dense_device;
             reference: mult_ab with a fully populated B, csr/kernels/numba/multiply.py:13-38, 110-122)
  transpose  configs[4]: csrk_transpose of the MovieLens-25M-shaped 162541 x 59047 matrix, nnz 25000095
             (csr/structure.py:172-204), bit-exact against the oracle at full size
  abt        configs[4]: mult_abt of ratings blocks A[2000] x B[20000]^T (csr/kernels/numba/multiply.py:41-57),
             values compared bit for bit with the oracle (columns compared as sets per row: DESIGN.md section 3)
  unit_rows  csrk_unit_rows_device on the headline matrix (csr/transform.py:29-66)
  spmv_f32   float32 values x float32 vector on the headline matrix: products rounded to float32 inside the planned
             kernels (csrk_spmv_f32x_device; csr/kernels/numba/__init__.py:55-67 as Numba types it)
  protocol   what a caller of the kernel protocol pays for mult_vec on the headline matrix with HOST vectors
             (csr/csr.py:569-590), next to the box's PCIe rates
"""
import ctypes as C
import time

import numpy as np
import torch

from csr_amd import synth
from csr_amd._lib import lib, check, handle_t

HBM_PEAK_GBS = 8000.0
ML_SHAPE = (162_541, 59_047, 25_000_095)


def _mk(rp, ci, vs, nrows, ncols):
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, ncols, int(ci.numel()), rp.data_ptr(), int(rp.dtype == torch.int64),
                                 ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
    return h


def _wall_ms(fn, reps, warm=2):
    "mean and minimum wall time of fn() in ms, the device idle before and after each call"
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.mean(ts)), float(np.min(ts))


def _events_ms(fn, reps, warm=2):
    "mean ms per call of `reps` back-to-back calls between two events on the null stream (where libcsrk launches by default)"
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _roof(alg_bytes, ms):
    gbs = alg_bytes / ms / 1e6
    return {'algorithmic_bytes': int(alg_bytes), 'achieved_gbs': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4)}


def ml_matrix(dev):
    nr, nc, nnz = ML_SHAPE
    m = synth.powerlaw_csr(nr, nc, nnz, device=dev, alpha=0.9, max_degree=7000)
    m['values'] = (torch.floor((m['values'] + 1.0) * 5.0).clamp_(0, 9) + 1.0) * 0.5      # ratings 0.5 .. 5.0
    return m


def spmm(dev, reps=10, cpu_entries=None):
    from oracle import oracle as O
    n, nnz, k = 2_000_000, 50_000_000, 64
    m = synth.powerlaw_csr(n, n, nnz, device=dev, max_degree=250_000)
    h = _mk(m['rowptrs'], m['colinds'], m['values'], n, n)
    B = synth.dense_vector(n * k, device=dev, stream=7).view(n, k)
    Cm = torch.empty(n, k, dtype=torch.float64, device=dev)
    ms = _events_ms(lambda: check(lib.csrk_spmm_dense_device(h, B.data_ptr(), k, k, Cm.data_ptr(), k, None)), reps)
    alg = nnz * 12 + (n + 1) * 4 + 2 * n * k * 8
    # property at full size: panel column 0 equals the SpMV with that column, to 1e-12 of sum |a x|
    x0 = B[:, 0].contiguous()
    y = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib.csrk_spmv_device(h, x0.data_ptr(), y.data_ptr(), None))
    absv = m['values'].abs()
    habs = _mk(m['rowptrs'], m['colinds'], absv, n, n)
    bound = torch.empty(n, dtype=torch.float64, device=dev)
    xa = x0.abs().contiguous()
    check(lib.csrk_spmv_device(habs, xa.data_ptr(), bound.data_ptr(), None))
    torch.cuda.synchronize()
    col0 = float(((Cm[:, 0] - y).abs() / (bound + 1e-300)).max())
    # the oracle on the WHOLE of configs[2] (cpu_entries=None; ~4 s on one core) or on the first rows holding ~cpu_entries
    # entries: baseline + parity
    rp_h = m['rowptrs'].cpu().numpy()
    r_s = n if cpu_entries is None else int(np.searchsorted(rp_h, cpu_entries))
    e_s = int(rp_h[r_s])
    ci_h, vs_h = m['colinds'][:e_s].cpu().numpy(), m['values'][:e_s].cpu().numpy()
    B_h = B.cpu().numpy()
    t0 = time.perf_counter()
    C_h = O.spmm_dense(r_s, rp_h[:r_s + 1], ci_h, vs_h, B_h)
    t_cpu = time.perf_counter() - t0
    # |C - C_oracle| <= 1e-6 * sum_j |a_ij| max_c |B_jc| is implied by the max-norm check below on U(-1,1) data
    samp = float(np.max(np.abs(C_h - Cm[:r_s].cpu().numpy())) / max(1e-300, float(np.max(np.abs(C_h)))))
    # What bounds this product is not its algorithmic bytes but the rows of B that have to ENTER a CU: one 8 k-byte row per
    # entry of a light row (L1 line fills: the guide's ~17 TB/s chip-wide for gathers served by L2), one LDS-staged tile
    # sweep of B per heavy row group (coalesced), and what crosses the fabric (measured by the PMC passes of the newest
    # profile, at the ~6.4 TB/s the fabric delivers).  The larger of the two times is the design's own bound.
    st = (C.c_int64 * 9)()
    check(lib.csrk_spmm_plan_stats(h, st, 9))
    heavy_nnz = int(st[5]) if int(st[0]) else 0
    light_fill = (nnz - heavy_nnz) * k * 8
    heavy_stage = int(st[3]) * n * k * 8 if int(st[0]) else 0
    fabric = None
    try:
        import glob
        import json
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        js = sorted(glob.glob(os.path.join(here, 'profiles', 'r*_configs_pmc_traffic.json')))
        tr = json.load(open(js[-1]))['hbm_bytes_per_launch'] if js else {}
        fabric = sum(v for kn, v in tr.items() if kn.startswith('spmm_')) or None
        fabric_src = os.path.basename(js[-1]) if js else None
    except Exception:                         # noqa: BLE001 -- the bound is reported without the measured half
        fabric_src = None
    l1_ms = (light_fill + heavy_stage) / 17e12 * 1e3
    fab_ms = fabric / 6.4e12 * 1e3 if fabric else None
    gather = {'light_row_l1_fill_bytes': light_fill, 'heavy_tile_staging_bytes': heavy_stage, 'l1_side_ms_at_17_TBs': round(l1_ms, 4),
              'fabric_bytes_measured': fabric, 'fabric_source': fabric_src, 'fabric_ms_at_6.4_TBs': None if fab_ms is None else round(fab_ms, 4),
              'bound_ms': round(max(l1_ms, fab_ms or 0.0), 4), 'frac_of_gather_bound': round(max(l1_ms, fab_ms or 0.0) / ms, 4),
              'heavy_rows': {'on': bool(st[0]), 'min_entries': int(st[1]), 'rows': int(st[2]), 'row_groups': int(st[3]),
                             'column_ranges': int(st[4]), 'entries': heavy_nnz}}
    via = None
    try:
        via = _spmm_via_mult_ab(dev, m, h, B, Cm, n, nnz, k, ms)
    except Exception as e:                    # noqa: BLE001 -- reported beside the panel figure, never hidden
        via = {'error': f'{type(e).__name__}: {e}'[:300]}
    check(lib.csrk_free(h))
    check(lib.csrk_free(habs))
    out = {'config': 'spmm_dense A 2000000x2000000 nnz 50000000 (power-law) x B 2000000x64 f64', 'entry': 'csrk_spmm_dense_device',
           'ms': round(ms, 4), 'gflops': round(2.0 * nnz * k / ms / 1e6, 1), 'bound': 'hbm', **_roof(alg, ms),
           'parity': {'col0_vs_spmv_max_err_over_sum_abs_terms': col0, 'sample_vs_oracle_max_rel_err': samp,
                      'tolerance': 1e-6, 'ok': bool(col0 <= 1e-6 and samp <= 1e-6)},
           'gather_bound': gather,
           'via_mult_ab': via,
           'cpu_baseline': {'value': round(2.0 * e_s * k / t_cpu / 1e9, 3), 'unit': 'GFLOP/s', 'cores': 1, 'kind': 'port',
                            'sample': (f'the whole of configs[2] ({e_s} entries)' if r_s == n else f'the first {r_s} rows of A ({e_s} entries)') +
                                      f' x the same B, one pass of orc_spmm_dense ({t_cpu:.2f} s)'}}
    return out


def _spmm_via_mult_ab(dev, m, ha, B, Cm, n, nnz, k, panel_ms, reps=5):
    """
    BASELINE configs[2] the way a reference caller reaches it: B as a fully populated CSR through mult_ab
    (csr/csr.py:524-567 -> csr/kernels/numba/multiply.py:13-38).  csrk_spgemm_ab recognises the row-major panel on the
    device, runs the dense-panel kernels and returns C as the reference does (k entries per row of C whose row of A holds
    an entry, columns k - 1 .. 0).  Wall per call: the check of B (0.5 GB of column indices), C's three arrays allocated,
    its index arrays written, the product, the result handle freed.
    """
    rp_b = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k).to(torch.int32)
    ci_b = torch.arange(k, device=dev, dtype=torch.int32).repeat(n)
    hb = _mk(rp_b, ci_b, B.reshape(-1), n, k)

    def run(keep=False):
        c = handle_t(0)
        check(lib.csrk_spgemm_ab(ha, hb, C.byref(c)))
        if keep:
            return c
        check(lib.csrk_free(c))
    ms, ms_min = _wall_ms(run, reps, warm=2)
    route = C.c_int(0)
    check(lib.csrk_spgemm_last_route(C.byref(route)))
    c = run(keep=True)
    nrc, ncc, nnzc = C.c_int32(), C.c_int32(), C.c_int64()
    check(lib.csrk_info(c, C.byref(nrc), C.byref(ncc), C.byref(nnzc), None, None))
    d_rp, d_ci, d_vs = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.csrk_device_ptrs(c, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))

    def dview(p, cnt, typestr):
        class _D:
            pass
        d = _D()
        d.__cuda_array_interface__ = {'shape': (cnt,), 'typestr': typestr, 'data': (int(p.value), False), 'version': 2}
        return torch.as_tensor(d, device=dev)
    live = (m['rowptrs'][1:] > m['rowptrs'][:-1])
    n_live = int(live.sum().item())
    g_rp = dview(d_rp, n + 1, '<i4').to(torch.int64)
    want_rp = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(live.to(torch.int64), 0)]) * k
    ok_rp = bool(nnzc.value == n_live * k and torch.equal(g_rp, want_rp))
    g_ci = dview(d_ci, nnzc.value, '<i4').view(n_live, k)
    ok_ci = bool(torch.equal(g_ci, torch.arange(k - 1, -1, -1, device=dev, dtype=torch.int32).expand(n_live, k)))
    # values: the panel product's rows with an entry in A, columns reversed -- the same kernels, so the same bits
    g_vs = dview(d_vs, nnzc.value, '<f8').view(n_live, k)
    ok_vs = bool(torch.equal(g_vs.flip(1).contiguous().view(torch.int64), Cm[live].contiguous().view(torch.int64)))
    check(lib.csrk_free(c))
    check(lib.csrk_free(hb))
    c_bytes = nnzc.value * 12 + (n + 1) * 4
    return {'config': 'mult_ab(A, CSR(B)): the same A and B, B handed over as a fully populated CSR (rowptrs 8 MB + colinds 0.5 GB + values 1 GB)',
            'entry': 'csrk_spgemm_ab', 'route': 'dense-panel' if route.value else 'general', 'ms': round(ms, 4), 'ms_min': round(ms_min, 4),
            'timing': 'wall per call (check of B, C allocated and indexed, product, result handle freed)',
            'over_the_panel_call': round(ms / panel_ms, 3), 'product_rows': n_live, 'product_nnz': int(nnzc.value), 'product_bytes': int(c_bytes),
            'bound_asked': '<= 1.5 x the panel call + C\'s 1.5 GB at the write rate',
            'parity': {'rowptrs_as_the_reference_returns_them': ok_rp, 'colinds_k_minus_1_down_to_0': ok_ci,
                       'values_bitwise_the_panel_product_reversed': ok_vs, 'ok': bool(ok_rp and ok_ci and ok_vs and route.value == 1)}}


def transpose(dev, m=None, reps=10):
    from oracle import oracle as O
    nr, nc, nnz = ML_SHAPE
    m = m or ml_matrix(dev)
    h = _mk(m['rowptrs'], m['colinds'], m['values'], nr, nc)

    def tr():
        t = handle_t(0)
        check(lib.csrk_transpose(h, 1, C.byref(t)))
        check(lib.csrk_free(t))              # what CSR.transpose does after from_handle
    ms, ms_min = _wall_ms(tr, reps)
    t = handle_t(0)
    check(lib.csrk_transpose(h, 1, C.byref(t)))
    rpt = np.empty(nc + 1, np.int32)
    ci_t = np.empty(nnz, np.int32)
    vs_t = np.empty(nnz)
    check(lib.csrk_export(t, rpt.ctypes.data_as(C.c_void_p), ci_t.ctypes.data_as(C.c_void_p), vs_t.ctypes.data_as(C.c_void_p)))
    rp_h, ci_h, vs_h = m['rowptrs'].cpu().numpy(), m['colinds'].cpu().numpy(), m['values'].cpu().numpy()
    t0 = time.perf_counter()
    _, _, orp, oci, ovs = O.transpose(nr, nc, rp_h, ci_h, vs_h)
    t_cpu = time.perf_counter() - t0
    exact = bool(np.array_equal(rpt, orp) and np.array_equal(ci_t, oci) and np.array_equal(vs_t.view(np.int64), ovs.view(np.int64)))
    check(lib.csrk_free(t))
    check(lib.csrk_free(h))
    alg = 4 * nnz + 2 * (4 + 8) * nnz + (nr + nc + 2) * 4
    return {'config': f'transpose MovieLens-25M shape {nr}x{nc} nnz {nnz} f64', 'entry': 'csrk_transpose',
            'ms': round(ms, 4), 'ms_min': round(ms_min, 4), 'timing': 'wall per call (create + free of the result handle included)',
            'bound': 'hbm', **_roof(alg, ms),
            'parity': {'bit_exact_vs_oracle_full_size': exact, 'ok': exact},
            'cpu_baseline': {'value': round(alg / t_cpu / 1e9, 3), 'unit': 'GB/s', 'ms': round(t_cpu * 1e3, 1), 'cores': 1, 'kind': 'port',
                             'sample': 'the whole matrix, one pass of orc_transpose'}}


def abt(dev, m=None, ra=2000, rb=20000, reps=3):
    from oracle import oracle as O
    nr, nc, nnz = ML_SHAPE
    m = m or ml_matrix(dev)

    def sub(r1):
        rp = m['rowptrs'][:r1 + 1].contiguous()
        return rp, int(rp[-1].item())
    rpa, ea = sub(ra)
    rpb, eb = sub(rb)
    ha = _mk(rpa, m['colinds'][:ea], m['values'][:ea], ra, nc)
    hb = _mk(rpb, m['colinds'][:eb], m['values'][:eb], rb, nc)

    def run(keep=False):
        c = handle_t(0)
        check(lib.csrk_spgemm_abt(ha, hb, C.byref(c)))
        if keep:
            return c
        check(lib.csrk_free(c))
    order = C.c_int(0)
    check(lib.csrk_spgemm_get_order(C.byref(order)))      # the library's default: the reference's column order (1) unless the environment asks for ascending
    ms, ms_min = _wall_ms(run, reps, warm=1)
    c = run(keep=True)
    nrc, ncc, nnzc = C.c_int32(), C.c_int32(), C.c_int64()
    check(lib.csrk_info(c, C.byref(nrc), C.byref(ncc), C.byref(nnzc), None, None))
    d_rp, d_ci, d_vs = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.csrk_device_ptrs(c, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))

    def dview(p, n, typestr, dt):
        class _D:
            pass
        d = _D()
        d.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (int(p.value), False), 'version': 2}
        return torch.as_tensor(d, device=dev)
    g_rp = dview(d_rp, ra + 1, '<i4', torch.int32)
    cnt_a = torch.bincount(m['colinds'][:ea].long(), minlength=nc).to(torch.float64)
    cnt_b = torch.bincount(m['colinds'][:eb].long(), minlength=nc).to(torch.float64)
    products = int(float((cnt_a * cnt_b).sum()))
    # the oracle on the same block: transpose + mult_ab = the reference's mult_abt, its rows' columns in reverse order of
    # first discovery (multiply.py:79-82, 94-97)
    ci_h, vs_h = m['colinds'][:eb].cpu().numpy(), m['values'][:eb].cpu().numpy()
    t0 = time.perf_counter()
    tnr, tnc, trp, tci, tvs = O.transpose(rb, nc, rpb.cpu().numpy(), ci_h, vs_h)
    r = O.mult_ab((ra, nc, rpa.cpu().numpy(), ci_h[:ea], vs_h[:ea]), (tnr, tnc, trp, tci, tvs))
    t_cpu = time.perf_counter() - t0
    o_rp, o_ci, o_vs = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (r[2], r[3], r[4]))

    def same_raw(cc):        # row pointers, column indices and values bit for bit the oracle's raw arrays, no sorting on either side
        check(lib.csrk_device_ptrs(cc, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))
        g_ci = dview(d_ci, nnzc.value, '<i4', torch.int32)
        g_vs = dview(d_vs, nnzc.value, '<f8', torch.float64)
        return bool(o_ci.numel() == nnzc.value and torch.equal(o_rp.to(torch.int64), g_rp.to(torch.int64))
                    and torch.equal(g_ci, o_ci.to(torch.int32)) and torch.equal(g_vs.view(torch.int64), o_vs.view(torch.int64)))

    def same_sorted(cc):     # ascending columns: (row, column) keys of the oracle's arrays sorted on the card, values bit for bit
        check(lib.csrk_device_ptrs(cc, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))
        g_ci = dview(d_ci, nnzc.value, '<i4', torch.int32)
        g_vs = dview(d_vs, nnzc.value, '<f8', torch.float64)
        if not (o_ci.numel() == nnzc.value and torch.equal(o_rp.to(torch.int64), g_rp.to(torch.int64))):
            return False
        rows = torch.repeat_interleave(torch.arange(ra, device=dev), (o_rp[1:] - o_rp[:-1]).long())
        key, perm = torch.sort(rows * int(rb) + o_ci.long())
        return bool(torch.equal(key, rows * int(rb) + g_ci.long())
                    and torch.equal(o_vs[perm].view(torch.int64), g_vs.view(torch.int64)))
    ok = same_raw(c) if order.value == 1 else same_sorted(c)
    abc = (ea + eb) * 12 + (ra + rb + 2) * 4 + nnzc.value * 12 + (ra + 1) * 4
    check(lib.csrk_free(c))
    # the other column order (csrk_spgemm_set_order): ascending saves the ordering pass
    check(lib.csrk_spgemm_set_order(1 - order.value))
    try:
        ms_other, _ = _wall_ms(run, 3, warm=1)
        c = run(keep=True)
        g_rp = dview(d_rp, ra + 1, '<i4', torch.int32)
        ok_other = same_sorted(c) if order.value == 1 else same_raw(c)
        check(lib.csrk_free(c))
    finally:
        check(lib.csrk_spgemm_set_order(-1))
    names = {0: 'ascending', 1: 'reference (reverse of first discovery: the raw arrays of csr/kernels/numba/multiply.py)'}
    ms_asc, ms_ref = (ms_other, ms) if order.value == 1 else (ms, ms_other)
    ok_raw, ok_sorted = (ok, ok_other) if order.value == 1 else (ok_other, ok)
    check(lib.csrk_free(ha))
    check(lib.csrk_free(hb))
    return {'config': f'mult_abt ({ra} x {nc}) x ({rb} x {nc})^T, rows of the MovieLens-25M-shaped matrix', 'entry': 'csrk_spgemm_abt',
            'ms': round(ms, 3), 'ms_min': round(ms_min, 3), 'timing': 'wall per call', 'intermediate_products': products,
            'products_per_s': round(products / ms * 1e3, -6), 'gflops_2_per_product': round(2.0 * products / ms / 1e6, 1),
            'product_nnz': int(nnzc.value), 'bound': 'data-dependent (no roofline stated: DESIGN.md section 6)',
            'a_bt_c_bytes': int(abc), 'a_bt_c_gbs': round(abc / ms / 1e6, 1),
            'column_order': names[order.value], 'ms_reference_order': round(ms_ref, 3), 'ms_ascending_order': round(ms_asc, 3),
            'parity': {'reference_order_rowptrs_colinds_and_values_bit_exact_vs_oracle_raw': ok_raw,
                       'ascending_order_colinds_and_values_bit_exact_vs_oracle_sorted': ok_sorted, 'ok': bool(ok_raw and ok_sorted)},
            'cpu_baseline': {'value': round(2.0 * products / t_cpu / 1e9, 3), 'unit': 'GFLOP/s', 'ms': round(t_cpu * 1e3, 1), 'cores': 1,
                             'kind': 'port', 'sample': 'the same block, one pass of orc_transpose + orc_mult_ab'}}


def unit_rows(dev, rp=None, ci=None, vs=None, nrows=None, ncols=None, reps=5, cpu_entries=40_000_000):
    "on the headline matrix (bench.py hands over its resident arrays; values are cloned: the operation is in place)"
    from oracle import oracle as O
    if rp is None:
        nrows = ncols = 10_000_000
        m = synth.powerlaw_csr(nrows, ncols, 200_000_000, device=dev)
        rp, ci, vs = m['rowptrs'], m['colinds'], m['values']
    nnz = int(ci.numel())
    vals = vs.clone()
    h = _mk(rp, ci, vals, nrows, ncols)
    norms = torch.empty(nrows, dtype=torch.float64, device=dev)
    ts = []
    for i in range(reps + 1):
        vals.copy_(vs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib.csrk_unit_rows_device(h, norms.data_ptr()))
        torch.cuda.synchronize()
        if i:
            ts.append((time.perf_counter() - t0) * 1e3)
    ms, ms_min = float(np.mean(ts)), float(np.min(ts))
    alg = 2 * nnz * 8 + (nrows + 1) * rp.element_size() + nrows * 8
    rp_h = rp.cpu().numpy()
    r_s = int(np.searchsorted(rp_h, cpu_entries))
    e_s = int(rp_h[r_s])
    v_h = vs[:e_s].cpu().numpy().copy()
    t0 = time.perf_counter()
    n_h = O.unit_rows(r_s, rp_h[:r_s + 1], v_h)
    t_cpu = time.perf_counter() - t0
    g_n, g_v = norms[:r_s].cpu().numpy(), vals[:e_s].cpu().numpy()
    with np.errstate(invalid='ignore', divide='ignore'):
        en = float(np.nanmax(np.abs(g_n - n_h) / np.maximum(np.abs(n_h), 1e-300)))
        ev = float(np.nanmax(np.abs(g_v - v_h)))          # unit rows: |v| <= 1, absolute = relative to the row norm
    nan_same = bool(np.array_equal(np.isnan(g_v), np.isnan(v_h)))
    check(lib.csrk_free(h))
    alg_s = 2 * e_s * 8 + (r_s + 1) * 4 + r_s * 8
    return {'config': f'unit_rows {nrows}x{ncols} nnz {nnz} f64 (the headline matrix), norms left in HBM', 'entry': 'csrk_unit_rows_device',
            'ms': round(ms, 4), 'ms_min': round(ms_min, 4), 'timing': 'wall per call', 'bound': 'hbm', **_roof(alg, ms),
            'parity': {'sample_norms_max_rel_err': en, 'sample_values_max_abs_err': ev, 'nan_pattern_identical': nan_same,
                       'tolerance': 1e-6, 'ok': bool(en <= 1e-6 and ev <= 1e-6 and nan_same)},
            'cpu_baseline': {'value': round(alg_s / t_cpu / 1e9, 3), 'unit': 'GB/s', 'cores': 1, 'kind': 'port',
                             'sample': f'the first {r_s} rows ({e_s} entries), one pass of orc_unit_rows ({t_cpu:.2f} s)'}}


def protocol(dev, rp=None, ci=None, vs=None, nrows=None, ncols=None, product_ms=None, reps=10):
    """
    What a caller of the kernel PROTOCOL pays for mult_vec on the headline matrix: host vector in, fresh host vector
    out (csr/csr.py:569-590; csr/kernels/mkl/multiply.py:31-41), through csr_amd.CSR.mult_vec -> hip.to_handle (handle
    cache warm: the matrix and its plan are in HBM) -> csrk_spmv -> release_handle.  Wall time per call, next to the
    PCIe rates of this box (80 MB each way, measured here with pinned and with pageable memory) and the product alone.
    """
    from csr_amd import CSR
    from csr_amd.kernels import hip
    if rp is None:
        nrows = ncols = 10_000_000
        m = synth.powerlaw_csr(nrows, ncols, 200_000_000, device=dev)
        rp, ci, vs = m['rowptrs'], m['colinds'], m['values']
    nnz = int(ci.numel())
    A = CSR(nrows, ncols, nnz, np.array(rp.cpu().numpy()), np.array(ci.cpu().numpy()), np.array(vs.cpu().numpy()), _cast=False)
    x_d = synth.dense_vector(ncols, device=dev)
    x = np.array(x_d.cpu().numpy())

    def med(fn, n=reps, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), float(np.min(ts))
    # the box's PCIe rates for a vector of this size
    d = torch.empty(ncols, dtype=torch.float64, device=dev)
    hp = torch.empty(ncols, dtype=torch.float64).pin_memory()
    hq = torch.from_numpy(x.copy())
    h2d_pin, _ = med(lambda: d.copy_(hp, non_blocking=True), 5)
    d2h_pin, _ = med(lambda: hp.copy_(d, non_blocking=True), 5)
    h2d_page, _ = med(lambda: d.copy_(hq), 5)
    d2h_page, _ = med(lambda: hq.copy_(d), 5)
    d2h_fresh, _ = med(lambda: torch.from_numpy(np.empty(ncols)).copy_(d), 3, 1)
    del hp, hq
    t0 = time.perf_counter()
    A.mult_vec(x)                              # copies the matrix to HBM (2.44 GB over PCIe)
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    A.mult_vec(x)                              # builds the plan
    second_ms = (time.perf_counter() - t0) * 1e3
    ms, ms_min = med(lambda: A.mult_vec(x))
    y = A.mult_vec(x)
    h = hip.to_handle(A)
    y_d = torch.empty(nrows, dtype=torch.float64, device=dev)
    if product_ms is None:
        product_ms, _ = med(lambda: check(lib.csrk_spmv_device(h.H, x_d.data_ptr(), y_d.data_ptr(), None)))
    check(lib.csrk_spmv_device(h.H, x_d.data_ptr(), y_d.data_ptr(), None))
    same = bool(np.array_equal(y, y_d.cpu().numpy()))
    x32 = x.astype(np.float32)
    ms32, _ = med(lambda: hip.mult_vec(h, x32), 5)
    hip.release_handle(h)
    hip.invalidate(A)
    hip.flush_result_pool()
    bound = product_ms + h2d_pin + d2h_pin
    return {'config': f'csr_amd.CSR.mult_vec(x) on the headline matrix {nrows}x{ncols} nnz {nnz}: host x in, fresh host y out, handle cache warm',
            'entry': 'CSR.mult_vec -> hip.to_handle / mult_vec / release_handle -> csrk_spmv', 'ms': round(ms, 3), 'ms_min': round(ms_min, 3),
            'timing': f'wall per call, median of {reps}',
            'split_ms': {'h2d_x': round(h2d_page, 3), 'product': round(product_ms, 4), 'd2h_y': round(d2h_page, 3),
                         'everything_else': round(ms - h2d_page - product_ms - d2h_page, 3)},
            'pcie_this_box': {'bytes_each_way': ncols * 8, 'pinned_h2d_ms': round(h2d_pin, 3), 'pinned_d2h_ms': round(d2h_pin, 3),
                              'pageable_h2d_ms': round(h2d_page, 3), 'pageable_d2h_ms': round(d2h_page, 3),
                              'pageable_d2h_into_a_fresh_array_ms': round(d2h_fresh, 3),
                              'pinned_gbs': round(ncols * 8 / (h2d_pin * 1e-3) / 1e9, 1)},
            'bound_ms': round(bound, 3), 'bound': 'product + 2 x 80 MB at the pinned PCIe rate', 'within_10pct_of_bound': bool(ms <= 1.1 * bound),
            'f32_x_ms': round(ms32, 3), 'first_call_ms_copies_the_matrix': round(first_ms, 1), 'second_call_ms_builds_the_plan': round(second_ms, 1),
            'parity': {'y_bitwise_equal_to_the_device_product': same, 'ok': same}}


def spmv_f32(dev, rp=None, ci=None, vs=None, nrows=None, ncols=None, reps=20, cpu_rows=1_000_000):
    """
    float32 values times a float32 vector on the headline matrix (the reference's generators draw float32 half the time,
    csr/test_utils.py:32,57-61): Numba types the loop's product as float32 (csr/kernels/numba/__init__.py:55-67), so every
    product is rounded to float32 before it joins the float64 sum -- inside the planned kernels (csrk_spmv_f32x_device).
    """
    from oracle import oracle as O
    from csr_amd import _lib
    if rp is None:
        nrows = ncols = 10_000_000
        m = synth.powerlaw_csr(nrows, ncols, 200_000_000, device=dev)
        rp, ci, vs = m['rowptrs'], m['colinds'], m['values']
    nnz = int(ci.numel())
    v32 = vs.to(torch.float32)
    h = handle_t(0)
    check(lib.csrk_create_device(nrows, ncols, nnz, rp.data_ptr(), int(rp.dtype == torch.int64), ci.data_ptr(), v32.data_ptr(),
                                 _lib.VAL_F32, C.byref(h)))
    x32 = synth.dense_vector(ncols, device=dev).to(torch.float32)
    y = torch.empty(nrows, dtype=torch.float64, device=dev)
    go = lambda: check(lib.csrk_spmv_f32x_device(h, x32.data_ptr(), y.data_ptr(), None))
    ms = _events_ms(go, reps, warm=3)
    rp_h = rp[:cpu_rows + 1].cpu().numpy()
    e_s = int(rp_h[-1])
    ci_h, v_h, x_h = ci[:e_s].cpu().numpy(), v32[:e_s].cpu().numpy(), x32.cpu().numpy()
    t0 = time.perf_counter()
    ref = O.mult_vec(cpu_rows, ncols, rp_h, ci_h, v_h, x_h)
    t_cpu = time.perf_counter() - t0
    bound = O.mult_vec(cpu_rows, ncols, rp_h, ci_h, np.abs(v_h).astype(np.float64), np.abs(x_h).astype(np.float64))
    ref64 = O.mult_vec(cpu_rows, ncols, rp_h, ci_h, v_h.astype(np.float64), x_h.astype(np.float64))
    got = y[:cpu_rows].cpu().numpy()
    err = float(np.max(np.abs(got - ref) / (bound + 1e-300)))
    check(lib.csrk_free(h))
    alg = nnz * 8 + (nrows + 1) * rp.element_size() + ncols * 4 + nrows * 8
    return {'config': f'mult_vec {nrows}x{ncols} nnz {nnz}, float32 values x float32 vector, y float64', 'entry': 'csrk_spmv_f32x_device',
            'ms': round(ms, 4), 'timing': f'device events, {reps} products (the kernels widen the float32 vector as they load it)',
            'gflops': round(2.0 * nnz / (ms * 1e-3) / 1e9, 1), 'bound': 'hbm', **_roof(alg, ms),
            'parity': {'sample_max_err_over_sum_abs_terms': err, 'tolerance': 1e-6,
                       'float32_products_differ_from_float64_by': float(np.max(np.abs(ref64 - ref) / (bound + 1e-300))),
                       'ok': bool(err <= 1e-6 and err < float(np.max(np.abs(ref64 - ref) / (bound + 1e-300))))},
            'cpu_baseline': {'value': round(2.0 * e_s / t_cpu / 1e9, 3), 'unit': 'GFLOP/s', 'cores': 1, 'kind': 'port',
                             'sample': f'the first {cpu_rows} rows ({e_s} entries), one pass of orc_mult_vec_f32f32 ({t_cpu:.2f} s)'}}


def run_all(dev, headline=None, log=None, product_ms=None):
    """
    -> {'protocol': ..., 'spmm': ..., 'transpose': ..., 'abt': ..., 'unit_rows': ..., 'seconds': ...}; a part that raises is reported as
    {'error': ...} (the headline line must still be printed).  `headline` = (rp, ci, vs, nrows, ncols) resident arrays.
    """
    out = {}
    t_all = time.perf_counter()

    def part(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:                # noqa: BLE001 -- reported, never hidden
            out[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
        out[name]['seconds'] = round(time.perf_counter() - t0, 2)
        if log:
            log(f'[bench secondary] {name}: {out[name].get("ms", out[name].get("error"))} ms, {out[name]["seconds"]} s')
    if headline is not None:
        part('protocol', lambda: protocol(dev, *headline, product_ms=product_ms))
        part('spmv_f32', lambda: spmv_f32(dev, *headline))
        part('unit_rows', lambda: unit_rows(dev, *headline))
    else:
        part('spmv_f32', lambda: spmv_f32(dev))
        part('protocol', lambda: protocol(dev))
        part('unit_rows', lambda: unit_rows(dev))
    part('spmm', lambda: spmm(dev))
    ml = None
    try:
        ml = ml_matrix(dev)
    except Exception as e:                    # noqa: BLE001
        out['transpose'] = out['abt'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    if ml is not None:
        part('transpose', lambda: transpose(dev, ml))
        part('abt', lambda: abt(dev, ml))
    del ml
    try:
        check(lib.csrk_trim_cache())
    except Exception:                         # noqa: BLE001
        pass
    out['seconds'] = round(time.perf_counter() - t_all, 2)
    return out


def main():
    """
    `python bench_secondary.py [--product-ms MS] [--only NAME ...]`: the secondary block as a process of its own -- how
    bench.py runs it (a child process: a hang or a GPU fault in here cannot cost the headline line, which the parent has
    timed and verified and still holds).  Prints ONE JSON object (the `secondary` value) as the last line of stdout.
    """
    import argparse
    import json
    import sys
    ap = argparse.ArgumentParser()
    ap.add_argument('--product-ms', type=float, default=None, help="the headline product's step time (for `protocol`'s split)")
    ap.add_argument('--only', nargs='*', default=None, help='run only these parts (spmm transpose abt unit_rows protocol spmv_f32)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    n, nnz = 10_000_000, 200_000_000
    m = synth.powerlaw_csr(n, n, nnz, alpha=1.1, device=dev)
    head = (m['rowptrs'], m['colinds'], m['values'], n, n)
    log = lambda t: print(t, file=sys.stderr, flush=True)      # noqa: E731
    if a.only:
        out = {}
        fns = {'spmm': lambda: spmm(dev), 'transpose': lambda: transpose(dev), 'abt': lambda: abt(dev),
               'unit_rows': lambda: unit_rows(dev, *head), 'protocol': lambda: protocol(dev, *head, product_ms=a.product_ms),
               'spmv_f32': lambda: spmv_f32(dev, *head)}
        for name in a.only:
            t0 = time.perf_counter()
            try:
                out[name] = fns[name]()
            except Exception as e:            # noqa: BLE001
                out[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
            out[name]['seconds'] = round(time.perf_counter() - t0, 2)
    else:
        out = run_all(dev, headline=head, log=log, product_ms=a.product_ms)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
