# scratch (GPU box): chunk handles (views of a rank's arrays at arbitrary entry offsets) against the oracle
import ctypes as C, numpy as np, torch, sys
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
from csr_amd.dist import chunk_cuts
from oracle import oracle as O
dev = 'cuda'
n = 2_500_000; nnz = 50_000_000
sh = synth.powerlaw_csr(n, n, nnz, device=dev)
rp, ci, vs = sh['rowptrs'], sh['colinds'], sh['values']
x = synth.dense_vector(n, device=dev)
ref = O.mult_vec(n, n, rp.cpu().numpy(), ci.cpu().numpy(), vs.cpu().numpy(), x.cpu().numpy())
bound = O.mult_vec(n, n, rp.cpu().numpy(), ci.cpu().numpy(), np.abs(vs.cpu().numpy()), np.abs(x.cpu().numpy()))
for K in (1, 2, 3):
    cuts = chunk_cuts(rp, K)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    keep = []
    for c in range(K):
        a, b = cuts[c], cuts[c + 1]
        e0, e1 = int(rp[a]), int(rp[b])
        rpc = (rp[a:b + 1] - rp[a]).contiguous(); cic, vsc = ci[e0:e1], vs[e0:e1]
        keep += [rpc, cic, vsc]
        h = handle_t(0)
        check(lib.csrk_create_device(b - a, n, e1 - e0, rpc.data_ptr(), 0, cic.data_ptr(), vsc.data_ptr(), 2, C.byref(h)))
        for call in range(3):
            check(lib.csrk_spmv_device(h, x.data_ptr(), y[a:b].data_ptr(), None))
            torch.cuda.synchronize()
            err = np.abs(y[a:b].cpu().numpy() - ref[a:b]) / (bound[a:b] + 1e-300)
            print(f'K={K} chunk {c} rows [{a},{b}) entries [{e0},{e1}) e0%4={e0 % 4} call {call}: max err/bound {err.max():.3e} at row {a + int(err.argmax())}', flush=True)
        check(lib.csrk_free(h))
