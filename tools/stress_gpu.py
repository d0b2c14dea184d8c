# GPU box: randomised stress of the planned paths at sizes between the unit tests and the full-size tests -- power-law and
# uniform matrices of random shape, skew and dtype; every result checked against an independent torch reduction (not the
# oracle: sizes up to 4e7 entries).  usage: python tools/stress_gpu.py [n_cases] [seed]
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t

dev = 'cuda'
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


def ref_spmv(rp, ci, vs, x):
    prod = (vs.to(torch.float64) if vs is not None else 1.0) * x[ci.long()]
    lens = (rp[1:] - rp[:-1]).long()
    y = torch.segment_reduce(prod, 'sum', lengths=lens, unsafe=True)
    b = torch.segment_reduce(prod.abs(), 'sum', lengths=lens, unsafe=True)
    return y, b


bad = 0
for case in range(n_cases):
    nr = int(10 ** rng.uniform(4.5 if os.environ.get('STRESS_BIG') else 3, 6.8))
    nc = int(10 ** rng.uniform(3, 6.7))
    lo = 6.0 if os.environ.get('STRESS_BIG') else 4.0
    nnz = int(min(10 ** rng.uniform(lo, 7.6), nr * min(nc // 8, 1000) * 0.5, 4e7))
    nnz = max(nnz, 1)
    alpha = float(rng.uniform(0.5, 1.5))
    maxdeg = int(10 ** rng.uniform(1.5, 6))
    uniform = rng.random() < 0.15
    f32 = rng.random() < 0.3
    structure = rng.random() < 0.1
    seed = int(rng.integers(1, 2 ** 31))
    t0 = time.time()
    try:
        m = synth.uniform_csr(nr, nc, nnz, seed=seed, device=dev) if uniform else \
            synth.powerlaw_csr(nr, nc, nnz, alpha=alpha, max_degree=maxdeg, seed=seed, device=dev)
    except Exception as e:      # (a shape the generator refuses: not what is under test)
        print(f'case {case}: generator refused ({type(e).__name__}: {str(e)[:80]})', flush=True)
        continue
    rp, ci = m['rowptrs'], m['colinds']
    nnz = int(ci.numel())
    vs = None if structure else (m['values'].to(torch.float32) if f32 else m['values'])
    rp32 = rp.to(torch.int32) if rp.dtype != torch.int32 else rp
    h = handle_t(0)
    vt = 0 if vs is None else (1 if f32 else 2)
    check(lib.csrk_create_device(nr, nc, nnz, rp32.data_ptr(), 0, ci.data_ptr(), vs.data_ptr() if vs is not None else None, vt,
                                 C.byref(h)))
    x = synth.dense_vector(nc, device=dev, seed=seed)
    ys = [torch.empty(nr, dtype=torch.float64, device=dev) for _ in range(3)]
    for y in ys:
        check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    yr, b = ref_spmv(rp, ci, vs, x)
    e1 = float(((ys[0] - yr).abs() / (b + 1e-300)).max()) if nr else 0.0
    e2 = float(((ys[1] - yr).abs() / (b + 1e-300)).max()) if nr else 0.0
    same = bool(torch.equal(ys[1].view(torch.int64), ys[2].view(torch.int64)))
    ok = e1 <= 1e-9 and e2 <= 1e-9 and same
    # transpose round trip (structure + values)
    t, tt = handle_t(0), handle_t(0)
    check(lib.csrk_transpose(h, 1, C.byref(t)))
    check(lib.csrk_transpose(t, 1, C.byref(tt)))
    d_rp, d_ci, d_vs = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.csrk_device_ptrs(tt, C.byref(d_rp), C.byref(d_ci), C.byref(d_vs)))
    rp2 = torch.as_tensor(_DevArray(d_rp.value, nr + 1, '<i4'), device=dev)
    tr_ok = bool(torch.equal(rp2, rp32))
    if nnz:
        ci2 = torch.as_tensor(_DevArray(d_ci.value, nnz, '<i4'), device=dev)
        tr_ok = tr_ok and bool(torch.equal(ci2, ci))
        if vs is not None:
            vs2 = torch.as_tensor(_DevArray(d_vs.value, nnz, '<f8'), device=dev)
            tr_ok = tr_ok and bool(torch.equal(vs2, vs.to(torch.float64)))
    check(lib.csrk_free(tt)); check(lib.csrk_free(t))
    # unit_rows against a torch reduction (norms at 1e-9 / 1e-5 relative)
    un_ok = True
    if vs is not None and nnz:
        vcopy = vs.clone()
        h2 = handle_t(0)
        check(lib.csrk_create_device(nr, nc, nnz, rp32.data_ptr(), 0, ci.data_ptr(), vcopy.data_ptr(), vt, C.byref(h2)))
        norms = torch.empty(nr, dtype=vs.dtype, device=dev)
        check(lib.csrk_unit_rows_device(h2, norms.data_ptr()))
        torch.cuda.synchronize()
        lens = (rp[1:] - rp[:-1]).long()
        nref = torch.segment_reduce(vs.to(torch.float64) ** 2, 'sum', lengths=lens, unsafe=True).sqrt()
        rel = 1e-5 if f32 else 1e-9
        un_ok = bool(((norms.to(torch.float64) - nref).abs() <= rel * nref + 1e-300).all())
        s2 = torch.segment_reduce(vcopy.to(torch.float64) ** 2, 'sum', lengths=lens, unsafe=True)
        nzr = nref > 0
        un_ok = un_ok and bool(((s2[nzr] - 1.0).abs() <= (1e-4 if f32 else 1e-9)).all())
        check(lib.csrk_free(h2))
    check(lib.csrk_free(h))
    allok = ok and tr_ok and un_ok
    bad += 0 if allok else 1
    print(f'case {case}: {"uniform" if uniform else f"alpha {alpha:.2f} maxdeg {maxdeg}"} {nr}x{nc} nnz {nnz} '
          f'{"structure" if structure else ("f32" if f32 else "f64")}: spmv first {e1:.1e} planned {e2:.1e} reproducible {same} '
          f'transpose {tr_ok} unit_rows {un_ok} ({time.time() - t0:.1f} s){"" if allok else "   <-- FAIL"}', flush=True)
    del m, rp, ci, vs, x, ys
    check(lib.csrk_trim_cache())
    torch.cuda.empty_cache()
print('FAILED cases:', bad)
sys.exit(1 if bad else 0)
