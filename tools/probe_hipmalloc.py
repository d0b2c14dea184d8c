#!/usr/bin/env python3
# scratch (GPU box): what a hipMalloc of a plan-sized array costs in a process shaped like bench.py (torch holding the
# matrix): sizes in MB from argv, each allocated, timed, touched by a memset and kept.
import ctypes as C, sys, time
import torch
hip = C.CDLL('libamdhip64.so')
dev = torch.device('cuda', 0)
hold = [torch.empty(int(1.6e9), dtype=torch.uint8, device=dev), torch.empty(int(0.8e9), dtype=torch.uint8, device=dev)]
tmp = torch.empty(int(3e9), dtype=torch.uint8, device=dev); del tmp      # (generation temporaries: back in torch's cache)
torch.cuda.synchronize()
kept = []
for mb in [float(a) for a in sys.argv[1:]] or [38, 296, 148, 1072, 268, 312, 156]:
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(int(mb * 1048576)))
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(int(mb * 1048576)))
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    print(f'hipMalloc {mb:8.1f} MB rc {rc}: {(t1 - t0) * 1e3:8.3f} ms, memset {(t2 - t1) * 1e3:8.3f} ms', flush=True)
    kept.append(p)
