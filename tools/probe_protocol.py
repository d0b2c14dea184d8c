# scratch (GPU box): what a caller of the kernel PROTOCOL pays for mult_vec on the headline matrix -- host vectors in,
# host vector out (csr/csr.py:569-590) -- against the raw PCIe rates of this box.
#   python tools/probe_protocol.py
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth, CSR
from csr_amd.kernels import hip
from csr_amd._lib import lib, check


def wall(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


dev = torch.device('cuda', 0)
n = int(os.environ.get('N', 10_000_000)); nnz = n * 20
nb = n * 8
# raw rates
d = torch.empty(n, dtype=torch.float64, device=dev)
hp = torch.empty(n, dtype=torch.float64).pin_memory()
hq = np.random.default_rng(0).random(n)
hq_t = torch.from_numpy(hq)
print('pinned   H2D %.3f ms (min %.3f)' % wall(lambda: d.copy_(hp, non_blocking=True)), flush=True)
print('pinned   D2H %.3f ms (min %.3f)' % wall(lambda: hp.copy_(d, non_blocking=True)), flush=True)
print('pageable H2D %.3f ms (min %.3f)' % wall(lambda: d.copy_(hq_t)), flush=True)
print('pageable D2H %.3f ms (min %.3f)' % wall(lambda: hq_t.copy_(d)), flush=True)
print('pageable D2H into a fresh np.empty %.3f ms (min %.3f)' % wall(lambda: torch.from_numpy(np.empty(n)).copy_(d)), flush=True)
t0 = time.perf_counter(); z = np.empty(n); z[::512] = 0; print('touching a fresh 80 MB array: %.3f ms' % ((time.perf_counter() - t0) * 1e3))

m = synth.powerlaw_csr(n, n, nnz, device=dev)
A = CSR(n, n, nnz, np.array(m['rowptrs'].cpu().numpy()), np.array(m['colinds'].cpu().numpy()), np.array(m['values'].cpu().numpy()), _cast=False)
del m
x = np.array(synth.dense_vector(n, device=dev).cpu().numpy())
t0 = time.perf_counter(); y0 = A.mult_vec(x); print('first CSR.mult_vec (copies the matrix): %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter(); y1 = A.mult_vec(x); print('second (builds the plan): %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
print('CSR.mult_vec, cache warm: %.3f ms (min %.3f)' % wall(lambda: A.mult_vec(x)), flush=True)
h = hip.to_handle(A)
print('hip.mult_vec on a held handle: %.3f ms (min %.3f)' % wall(lambda: hip.mult_vec(h, x)), flush=True)
y = np.empty(n)
px, py = x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)
print('csrk_spmv into a reused y: %.3f ms (min %.3f)' % wall(lambda: check(lib.csrk_spmv(h.H, px, py))), flush=True)
x32 = x.astype(np.float32)
print('hip.mult_vec f32 x: %.3f ms (min %.3f)' % wall(lambda: hip.mult_vec(h, x32)), flush=True)
y2 = hip.mult_vec(h, x)
print('same bits as first call:', bool(np.array_equal(y1, y2)), ' max |y - y_first_call| %.3e' % float(np.abs(y2 - y0).max()))
hip.release_handle(h)
