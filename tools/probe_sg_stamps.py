# scratch (GPU box): per-unit timing of the numeric strip kernel (library built with -DCSRK_SG_STAMPS)
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
m = synth.movielens_like(device='cuda'); nc = m['ncols']
def sub(r1):
    rp = m['rowptrs'][:r1 + 1].contiguous(); e = int(rp[-1].item()); hh = handle_t(0)
    check(lib.csrk_create_device(r1, nc, e, rp.data_ptr(), 0, m['colinds'].data_ptr(), m['values'].data_ptr(), 2, C.byref(hh)))
    return hh, rp
ha, ka = sub(2000); hb, kb = sub(20000)
for i in range(2):
    c = handle_t(0); check(lib.csrk_spgemm_abt(ha, hb, C.byref(c))); torch.cuda.synchronize(); check(lib.csrk_free(c))
S = -(-20000 // 832); n = 2000 * S
buf = np.zeros(n * 8, dtype=np.uint64)
lib.csrk_debug_sg_stamps.argtypes = [C.c_void_p, C.c_int]
check(lib.csrk_debug_sg_stamps(buf.ctypes.data_as(C.c_void_p), buf.size))
b = buf.reshape(n, 8).astype(np.float64)
cyc, J, pos, ch, t0, pa, pb, pc = b.T
t0 -= t0[t0 > 0].min()
print('units', n, 'S', S, ' clock: s_memtime ticks (100 MHz => 10 ns each)')
print('sum ticks %.3g  max %.3g  mean %.3g' % (cyc.sum(), cyc.max(), cyc.mean()))
print('kernel span ticks %.3g' % ((t0 + cyc).max()))
o = np.argsort(-cyc)[:10]
for u in o: print('unit %6d J %5d pos %7d chunks %5d ticks %8d  setup %8d issue %8d apply %8d' % (u, J[u], pos[u], ch[u], cyc[u], pa[u], pb[u], pc[u]))
print('totals: setup %.3g issue %.3g apply %.3g of %.3g' % (pa.sum(), pb.sum(), pc.sum(), cyc.sum()))
print('all: ticks/chunk %.2f, ticks/J %.2f' % (cyc.sum() / ch.sum(), cyc.sum() / J.sum()))
# linear fit ticks ~ a*chunks + b*batches + c
A = np.stack([ch, np.ceil(J / 64), np.ones_like(J)], 1)
co = np.linalg.lstsq(A, cyc, rcond=None)[0]
print('fit: %.2f ticks/chunk + %.2f ticks/batch + %.1f' % tuple(co))
late = np.argsort(-(t0 + cyc))[:5]
for u in late: print('last-finishing unit %6d J %5d chunks %5d start %8d end %8d' % (u, J[u], ch[u], t0[u], t0[u] + cyc[u]))
