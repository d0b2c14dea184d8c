# GPU box: randomised stress of csrk_spgemm_ab / _abt against SciPy on the host (what the reference's own tests compare
# with): power-law operands of random shape and skew, rows of A up to tens of thousands of entries.
# usage: python tools/stress_spgemm.py [n_cases] [seed]
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def draw(nr, nc, nnz, rng):
    heavy = bool(os.environ.get('STRESS_HEAVY'))
    m = synth.powerlaw_csr(nr, nc, nnz, alpha=float(rng.uniform(1.0 if heavy else 0.5, 1.4)), max_degree=int(10 ** rng.uniform(3.8 if heavy else 1.5, 4.7)),
                           seed=int(rng.integers(1, 2 ** 31)), device='cpu')
    rp = m['rowptrs'].numpy().astype(np.int32)
    return sp.csr_matrix((m['values'].numpy(), m['colinds'].numpy().astype(np.int32), rp), shape=(nr, nc))


def to_handle(a):
    h = handle_t(0)
    check(lib.csrk_create(a.shape[0], a.shape[1], a.nnz, a.indptr.ctypes.data_as(C.c_void_p), 0,
                          a.indices.ctypes.data_as(C.c_void_p), a.data.ctypes.data_as(C.c_void_p), 2, C.byref(h)))
    return h


def export(h):
    r, c, n, p, v = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int(), C.c_int()
    check(lib.csrk_info(h, C.byref(r), C.byref(c), C.byref(n), C.byref(p), C.byref(v)))
    rp = np.empty(r.value + 1, dtype=np.int64 if p.value else np.int32)
    ci = np.empty(n.value, dtype=np.int32)
    vs = np.empty(n.value, dtype=np.float64)
    check(lib.csrk_export(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), vs.ctypes.data_as(C.c_void_p)))
    return sp.csr_matrix((vs, ci, rp), shape=(r.value, c.value))


bad = 0
for case in range(n_cases):
    m_, k_, n_ = (int(10 ** rng.uniform(2.5, 5.2)) for _ in range(3))
    if os.environ.get('STRESS_HEAVY'):      # long rows in A: the strip / very-long-row paths
        k_ = int(10 ** rng.uniform(4.8, 5.6))
    nnza = int(min(10 ** rng.uniform(3.5, 6.2), m_ * min(k_ // 8, 1000) * 0.5))
    nnzb = int(min(10 ** rng.uniform(3.5, 6.2), k_ * min(n_ // 8, 1000) * 0.5))
    abt = rng.random() < 0.4
    try:
        A = draw(m_, k_, max(nnza, 1), rng)
        B = draw(n_, k_, max(nnzb, 1), rng) if abt else draw(k_, n_, max(nnzb, 1), rng)
    except Exception as e:
        print(f'case {case}: generator refused ({str(e)[:60]})', flush=True)
        continue
    Bm = B.T.tocsr() if abt else B
    products = int((A.astype(bool).astype(np.int64) @ np.diff(Bm.indptr).astype(np.int64)).sum())
    if products > 8e7:
        print(f'case {case}: skipped ({products} products)', flush=True)
        continue
    t0 = time.time()
    ref = A @ Bm
    ref.sort_indices()
    bound = (abs(A) @ abs(Bm)).tocsr()
    bound.sort_indices()
    t_ref = time.time() - t0
    ha, hb, hc = to_handle(A), to_handle(B), handle_t(0)
    t0 = time.time()
    check((lib.csrk_spgemm_abt if abt else lib.csrk_spgemm_ab)(ha, hb, C.byref(hc)))
    Cm = export(hc)
    t_gpu = time.time() - t0
    ref_order_ok = True
    order = C.c_int(0)
    check(lib.csrk_spgemm_get_order(C.byref(order)))
    if order.value == 1:
        # the reference's own column order (the default): bit for bit the oracle's raw arrays; then sorted for the SciPy comparison
        from oracle import oracle as O
        Bo = Bm.tocsr()
        _, _, orp, oci, _ = O.mult_ab((A.shape[0], A.shape[1], A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data),
                                      (Bo.shape[0], Bo.shape[1], Bo.indptr.astype(np.int32), Bo.indices.astype(np.int32), Bo.data))
        ref_order_ok = np.array_equal(Cm.indptr, orp) and np.array_equal(Cm.indices, oci)
        Cm.sort_indices()
    ok = ref_order_ok and Cm.shape == ref.shape and np.array_equal(Cm.indptr, ref.indptr) and np.array_equal(Cm.indices, ref.indices)
    if ok:
        ok = bool(np.all(np.abs(Cm.data - ref.data) <= 1e-12 * bound.data + 1e-300)) if np.array_equal(bound.indices, ref.indices) \
            else bool(np.allclose(Cm.data, ref.data, rtol=1e-9, atol=1e-9))
    bad += 0 if ok else 1
    print(f'case {case}: {"ABt" if abt else "AB "} {A.shape} nnz {A.nnz} (max row {int(np.diff(A.indptr).max())}) x {B.shape} nnz {B.nnz}: '
          f'{products} products -> {ref.nnz} entries, ok {ok}  (scipy {t_ref:.2f} s, here {t_gpu:.2f} s){"" if ok else "   <-- FAIL"}', flush=True)
    for h in (hc, hb, ha):
        check(lib.csrk_free(h))
print('FAILED cases:', bad)
sys.exit(1 if bad else 0)
