# scratch: local SpMV time of ONE rank's row range for world = 1, 2, 4, 8 (what the N-GPU bench computes per rank
# before the exchange); usage: PYTHONPATH=. python tools/probe_rank.py ["ENV=1 ENV2=2" ...]   (one pass per argument: the
# CSRK_* settings a plan is built under; none = defaults)
import ctypes as C, torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csr_amd import synth
from csr_amd._lib import lib, check, handle_t
dev = 'cuda'
n = 10_000_000; nnz = 200_000_000
x = synth.dense_vector(n, device=dev)
import os
base_env = dict(os.environ)
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
  for cfg in sys.argv[1:] or ['']:
    for k in [k for k in os.environ if k.startswith('CSRK_') and k not in base_env]:
        del os.environ[k]
    for kv in cfg.split():
        k, _, v = kv.partition('=')
        os.environ[k] = v
    if cfg:
        print(f'[{cfg}]')
    for rank in sorted({0}):
        sh = synth.powerlaw_csr(n, n, nnz, device=dev, rank=rank, world=world)
        rp, ci, vs = sh['rowptrs'], sh['colinds'], sh['values']
        nl = sh['row_end'] - sh['row_begin']
        h = handle_t(0)
        check(lib.csrk_create_device(nl, n, int(ci.numel()), rp.data_ptr(), 0, ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
        y = torch.empty(nl, dtype=torch.float64, device=dev)
        for _ in range(4): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(30): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        e1.record(); torch.cuda.synchronize()
        check(lib.csrk_spmv_profile_every(h, 1)); check(lib.csrk_spmv_profile_begin(h, 10))
        for _ in range(10): check(lib.csrk_spmv_device(h, x.data_ptr(), y.data_ptr(), None))
        torch.cuda.synchronize(); nr = C.c_int(0); km = (C.c_float * 4)(); check(lib.csrk_spmv_profile_end4(h, C.byref(nr), km))
        print('   kernels ms: light %.4f  tier0 %.4f  tier1 %.4f  stage %.4f' % tuple(km), flush=True)
        st = (C.c_int64 * 25)(); check(lib.csrk_spmv_plan_stats(h, st, 25))
        print(f'world {world} rank {rank}: rows {nl} nnz {int(ci.numel())}: {e0.elapsed_time(e1) / 30:.4f} ms  (tier0 rows {st[9]} entries {st[10]} slots {st[4]*512}, tier1 entries {st[13]}, light {st[3]}, pack {st[16]}, staged {st[24]})', flush=True)
        check(lib.csrk_free(h)); del sh, rp, ci, vs, y
