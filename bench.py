#!/usr/bin/env python3
"""
Headline benchmark: fp64 SpMV (the reference's mult_vec, csr/kernels/numba/__init__.py:55-67)
on a 10M x 10M power-law CSR with nnz = 2e8, through the libcsrk C ABI on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in-process; for N > 1 the driver launches one rank per GPU
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) -- or, started plainly as
`python bench.py --gpus N`, bench.py starts that launcher itself as a child process before touching the GPU -- the matrix is
row-range partitioned (nnz balanced) with x replicated, and one step = local SpMV + the exchange
that completes y on every rank.  xGMI is point-to-point and the exchange of the 80 MB y is link-bound, so
how RCCL drives the links decides the step: before the warm-up the candidates of csr_amd/dist.py (padded
all-gather; the unpadded all-gather; point-to-point sends straight into y with the slice travelling while the tiers'
part of the product runs) are each timed for 8 steps and the fastest runs the timed region
(`multi_gpu.candidates_ms_per_step`; a candidate that fails, disagrees or exceeds --calibrate-seconds is dropped and the
padded all-gather is the fallback; `--collective allreduce` forces the all-reduce north_star names, `--collective NAME`
any other).
The matrix is FIXED as N grows ("scaling": "strong").

One JSON line is printed by rank 0:
  value      whole-job GFLOP/s = 2 * nnz / (max-over-ranks wall time per step), inputs resident
             in HBM before the timed region;
  roofline   the slowest of the SpMV's streaming kernels (light stream with its staging pass, accumulator
             tier 0, panel tier 1: DESIGN.md section 4): that kernel's own algorithmic bytes per launch
             (12 B per entry it processes + its row pointers / partials + x once) divided by its
             mean duration, measured live with hipEvent pairs recorded around that kernel on its
             launch stream during every 10th timed step (csrk_spmv_profile_every/begin/end4: a pair
             costs ~3 us on the stream, eight per step would add 4 % to the step); `all_kernels`
             lists every kernel, `whole_spmv` the same fraction for the whole 2.60 GB,
             `hbm_gbs_end_to_end` is 2.60 GB / step time;
             peak = 8000 GB/s (HBM3E spec); traffic = per-launch HBM bytes from the committed
             rocprofv3 PMC summary (profiles/), or null;
  cpu_baseline  the oracle's sequential restatement of the reference loop (oracle/csr_oracle.c,
             kind "port", 1 core) timed on this host on the same matrix, plus a full-size
             parity check of the GPU result against it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)      # 0.3 s of SpMVs: the first ~50 run 2 % slower (clocks, caches)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the matrix (testing only; INVALID as a result)')
    ap.add_argument('--alpha', type=float, default=1.1)
    ap.add_argument('--algo', default='auto', choices=['auto', 'merge', 'vector', 'scalar'])
    ap.add_argument('--collective', default='auto',
                    choices=['auto', 'allgather', 'allgatherv', 'allreduce', 'p2p-split'],
                    help='N > 1: how y is completed on every rank; auto = time the candidates before the warm-up and keep the fastest')
    ap.add_argument('--calibrate-seconds', type=float, default=20.0,
                    help='N > 1: a candidate whose build + 3 untimed steps take longer than this is dropped')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the N > 1 code path (process group on RCCL, exchange candidates, completeness checks) whatever N is: '
                         'with --gpus 1 it is the one-rank rehearsal of the multi-GPU run (tests/test_gpu_fullsize.py)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='N = 1: skip the `secondary` block (SpMM configs[2], transpose + A B^T configs[4], unit_rows on the '
                         'headline matrix: bench_secondary.py), which adds ~30 s after the SpMV has been timed')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline time budget')
    ap.add_argument('--secondary-seconds', type=float, default=240.0, help='N = 1: time limit of the secondary block (a child process)')
    ap.add_argument('--traffic-json', default=None, help='rocprofv3 PMC summary with per-launch HBM bytes')
    return ap.parse_args()


def load_traffic(path, workload, kernel):
    "(per-launch HBM bytes of `kernel`, the file they come from) from a committed PMC summary, or (None, None)"
    cands = [path] if path else []
    pdir = os.path.join(ROOT, 'profiles')
    if os.path.isdir(pdir):
        cands += sorted((os.path.join(pdir, f) for f in os.listdir(pdir) if f.endswith('_pmc_traffic.json')),
                        reverse=True)
    for p in cands:
        try:
            with open(p) as f:
                d = json.load(f)
            key = kernel
            if d.get('workload') == workload and key in d.get('hbm_bytes_per_launch', {}):
                return float(d['hbm_bytes_per_launch'][key]), os.path.relpath(p, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def _visible_gpus():
    "GPUs this process could use, counted WITHOUT touching the HIP runtime (sysfs KFD topology); None if unknown"
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(',') if t.strip() != ''])
    if not os.path.exists('/dev/kfd'):
        return 0
    top = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(top):
            with open(os.path.join(top, node, 'properties')) as f:
                for line in f:
                    k, _, v = line.partition(' ')
                    if k == 'simd_count' and int(v) > 0:
                        n += 1
        return n
    except (OSError, ValueError):
        return None


def self_launch(args):
    """
    `python bench.py --gpus N` with N > 1 and no launcher around it: start the one-rank-per-GPU job as a CHILD process
    (python -m torch.distributed.run ... bench.py <same flags>) before this process has imported torch or touched the
    GPU -- a process that has initialised HIP must never exec -- relay its output and exit with its code.
    """
    import socket
    import subprocess
    have = _visible_gpus()
    need = 1 if os.environ.get('BENCH_TEST_SHARE_GPU') == '1' else args.gpus      # (test hook: all ranks on GPU 0)
    if have is not None and have < need:
        sys.exit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this host '
                 f'(the product has no CPU fallback; N > 1 needs one MI355X per rank)')
    with socket.socket() as sk:                       # a free rendezvous port on the loopback
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    print(f'[bench] --gpus {args.gpus} without a launcher: starting {" ".join(cmd[1:7])} ... as a child process',
          file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                             # does not return
    import numpy as np
    import torch
    import torch.distributed as dist

    from csr_amd import synth
    from csr_amd._lib import lib, check, handle_t, SPMV_AUTO, SPMV_MERGE, SPMV_VECTOR, SPMV_SCALAR
    from csr_amd.dist import RowPartitionedSpMV, SplitPhaseRowPartitionedSpMV, hip_local_spmv, hip_local_spmv_parts

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: no GPU is visible (the product has no CPU fallback)')
    # BENCH_TEST_SHARE_GPU=1 (plumbing test on a 1-GPU box only): all ranks use GPU 0 over gloo, because
    # RCCL refuses two ranks on one device.  Never set by the driver.
    share = os.environ.get('BENCH_TEST_SHARE_GPU') == '1'
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    check(lib.csrk_set_device(dev_index))
    distd = world > 1 or args.force_dist      # the row-partitioned path with its exchange step
    if distd:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:           # (--force-dist without a launcher: one rank, any free port)
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        import datetime
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            # a collective that does not complete aborts the job after 5 minutes instead of holding the node
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(minutes=5))

    nrows = ncols = int(round(10_000_000 * args.scale))
    nnz = int(round(200_000_000 * args.scale))
    workload = f'spmv_powerlaw_{nrows}x{ncols}_nnz{nnz}_fp64'

    t_gen = time.time()
    shard = synth.powerlaw_csr(nrows, ncols, nnz, alpha=args.alpha, device=dev, rank=rank, world=world)
    x = synth.dense_vector(ncols, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    rp, ci, vs = shard['rowptrs'], shard['colinds'], shard['values']
    n_loc = shard['row_end'] - shard['row_begin']
    nnz_loc = int(ci.numel())

    h = handle_t(0)
    check(lib.csrk_create_device(n_loc, ncols, nnz_loc, rp.data_ptr(), int(rp.dtype == torch.int64),
                                 ci.data_ptr(), vs.data_ptr(), 2, C.byref(h)))
    algo_code = {'auto': SPMV_AUTO, 'merge': SPMV_MERGE, 'vector': SPMV_VECTOR, 'scalar': SPMV_SCALAR}[args.algo]
    check(lib.csrk_set_spmv_algo(h, algo_code))

    def barrier():
        if distd:
            dist.barrier()
        torch.cuda.synchronize()

    # the first call on a handle runs the plan-less kernel, the second builds the plan (tiers, streams, pack):
    # both timed here, outside the warm-up and the timed region
    local = hip_local_spmv(h.value)
    y_tmp = torch.zeros(max(n_loc, 1), dtype=torch.float64, device=dev)
    once = []
    for _ in range(2):
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        local(x, y_tmp[:n_loc])
        torch.cuda.synchronize()
        once.append((time.perf_counter() - t_w) * 1e3)
    first_ms, plan_ms = once
    # Set-up check, untimed and before the warm-up: the planned product is bitwise reproducible (no float atomic
    # decides an order anywhere in it) -- REPRO_CALLS products, every one compared bit for bit with the first.  It also
    # means the card has been running the SpMV for a quarter of a second when the warm-up starts (the first ~50 steps
    # of a cold card run ~2 % slower: clocks, caches), however few warm-up steps the caller asks for.
    REPRO_CALLS = 300
    y_ref = y_tmp[:n_loc].clone()
    n_differ = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(REPRO_CALLS):
        local(x, y_tmp[:n_loc])
        n_differ += (y_tmp[:n_loc].view(torch.int64) != y_ref.view(torch.int64)).any().to(torch.int64)
    torch.cuda.synchronize()
    reproducible = int(n_differ.item()) == 0
    if not reproducible:
        sys.exit(f'[bench rank {rank}] the planned SpMV is not bitwise reproducible: {int(n_differ.item())} of {REPRO_CALLS} products differ')
    del y_tmp, y_ref

    # N > 1: the ways of completing y on every rank (csr_amd/dist.py)
    def make_op(name):
        "-> (operator, handles it owns)"
        if not distd or name in ('allgather', 'allgatherv', 'allreduce'):
            return RowPartitionedSpMV(shard['bounds'], rank, world, local, dev,
                                      mode=name if distd else 'allgather'), []
        assert name == 'p2p-split'
        run_part, cut_rows = hip_local_spmv_parts(h.value, dev)
        return SplitPhaseRowPartitionedSpMV(shard['bounds'], rank, world, run_part, cut_rows, dev), []

    calibration = None
    if distd and args.collective == 'auto':
        # measure, don't guess: the exchange is bound by the point-to-point links and by how well RCCL drives them,
        # which a 1-GPU box cannot show.  Every candidate runs 3 untimed steps (its chunk handles build their
        # plans) and 8 timed ones; the slowest rank's time decides, so every rank picks the same one.
        calibration, best = {}, None
        ref_sum = None
        # the padded all-gather (the safe fallback, always first), the unpadded all-gather and the split-phase
        # point-to-point form: none builds another plan
        names = ('allgather', 'allgatherv', 'p2p-split')
        for name in names:
            cand, hs, err = None, [], None
            try:
                t_build = time.perf_counter()
                cand, hs = make_op(name)
                for _ in range(3):
                    yc = cand.step(x)
                torch.cuda.synchronize()
                t_build = time.perf_counter() - t_build
                if t_build > args.calibrate_seconds and name != 'allgather':
                    err = f'build + 3 steps took {t_build:.1f} s (budget {args.calibrate_seconds:.0f} s)'
                # every candidate must produce the first one's y: bit for bit when it runs the same plan, to 1e-9 of
                # max |y| when its chunks have plans of their own (their tiers cut the sums differently)
                if err:
                    pass
                elif ref_sum is None:
                    ref_sum = yc.clone()
                elif hs:
                    dmax, ymax = float((yc - ref_sum).abs().max().item()), float(ref_sum.abs().max().item())
                    if not dmax <= 1e-9 * ymax:
                        err = f'result differs from the all-gather form: max |dy| = {dmax:.3e}, max |y| = {ymax:.3e}'
                elif not torch.equal(yc, ref_sum):
                    err = 'result differs from the all-gather form (same plan: must be identical)'
            except Exception as e:                       # e.g. a backend without this exchange (gloo test hook)
                err = f'{type(e).__name__}: {e}'[:200]
            if err:
                print(f'[bench rank {rank}] exchange candidate {name} dropped: {err}', file=sys.stderr, flush=True)
            bad = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bad.item() > 0:
                calibration[name] = err or 'failed on another rank'
                for hc in hs:
                    check(lib.csrk_free(hc))
                continue
            barrier()
            t_c = time.perf_counter()
            for _ in range(8):
                cand.step(x)
            barrier()
            t = torch.tensor([time.perf_counter() - t_c], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            calibration[name] = round(float(t.item()) / 8 * 1e3, 4)
            if best is None or calibration[name] < calibration[best[0]]:
                if best is not None:
                    for hc in best[2]:
                        check(lib.csrk_free(hc))
                best = (name, cand, hs)
            else:
                for hc in hs:
                    check(lib.csrk_free(hc))
            del cand
        if best is None:
            sys.exit(f'no exchange candidate ran: {calibration}')
        collective, op, op_handles = best
        del ref_sum
    else:
        collective = args.collective if distd else 'none'
        if collective == 'auto':
            collective = 'none'
        op, op_handles = make_op(collective)
    # the handle whose kernels are timed for the roofline: the rank's row range, or its first chunk
    fallback_note = None
    while True:
        hp = op_handles[0] if op_handles else h
        # Kernel event pairs cost ~3 us apiece on the stream (~25 us for a product with four timed kernels).  During the
        # warm-up every kernel is timed on every step; the timed region then brackets only the kernel that was the
        # slowest there (the roofline's dominant kernel), on every 10th step of a long run.
        events = not os.environ.get('BENCH_NO_KERNEL_EVENTS')
        k_warm, n_warm = None, 0
        if events and args.warmup > 0:
            check(lib.csrk_spmv_profile_channels(hp, 0xf))
            check(lib.csrk_spmv_profile_every(hp, 1))
            check(lib.csrk_spmv_profile_begin(hp, args.warmup + 2))
        for _ in range(args.warmup):
            op.step(x)
        barrier()
        chan_mask = 0xf
        if events and args.warmup > 0:
            n_w, k_w = C.c_int(0), (C.c_float * 4)(0.0, 0.0, 0.0, 0.0)
            check(lib.csrk_spmv_profile_end4(hp, C.byref(n_w), k_w))
            k_warm, n_warm = [float(v) for v in k_w], n_w.value
            if n_warm > 0 and max(k_warm) > 0:
                chan_mask = 1 << max(range(4), key=lambda c: k_warm[c])      # the slowest kernel is the one bracketed in the timed region
        if events:
            every = 10 if args.steps >= 20 else (5 if args.steps >= 10 else 1)
            check(lib.csrk_spmv_profile_channels(hp, chan_mask))
            check(lib.csrk_spmv_profile_every(hp, every))
            check(lib.csrk_spmv_profile_begin(hp, args.steps // every + 2))
        op.timing = distd
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = op.step(x)
        barrier()
        elapsed = time.perf_counter() - t0
        n_rec, k_live = C.c_int(0), (C.c_float * 4)(0.0, 0.0, 0.0, 0.0)
        if events:
            check(lib.csrk_spmv_profile_end4(hp, C.byref(n_rec), k_live))
            check(lib.csrk_spmv_profile_channels(hp, 0xf))
        # per kernel: the timed region's mean where it was bracketed there, else the warm-up's
        k_ms2 = [float(k_live[c]) if (chan_mask >> c) & 1 and k_live[c] > 0 else (k_warm[c] if k_warm else 0.0) for c in range(4)]
        k_src = ['timed region' if (chan_mask >> c) & 1 and k_live[c] > 0 else ('warm-up' if k_warm else None) for c in range(4)]
        compute_ms = None
        exchange_ok = None
        if not distd:
            break
        # every rank must hold the same complete y: the wrapping int64 sum of the bit patterns of a rank's own slice,
        # summed over ranks, equals that of the whole vector on every rank (order-independent, exact)
        own = y[shard['row_begin']:shard['row_end']].view(torch.int64).sum().reshape(1)
        tot = own.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        agree = (y.view(torch.int64).sum().reshape(1) == tot).to(torch.float64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        exchange_ok = bool(agree.item() > 0)
        if exchange_ok:
            t = torch.tensor([elapsed, op.compute_ms()], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, compute_ms = float(t[0].item()), float(t[1].item())
            break
        if collective == 'allgather':
            sys.exit(f'EXCHANGE FAILURE: rank {rank} does not hold the complete y after {collective}')
        # the chosen form passed its check before the warm-up and failed now: report it and time the plain
        # all-gather form instead of leaving the run without a result
        print(f'[bench rank {rank}] y incomplete after the timed region with {collective}: falling back to allgather',
              file=sys.stderr, flush=True)
        fallback_note = f'{collective} failed the completeness check after the timed region'
        for hc in op_handles:
            check(lib.csrk_free(hc))
        collective = 'allgather'
        op, op_handles = make_op(collective)

    algo_name = lib.csrk_spmv_algo_name(hp).decode()
    n_tiles, tile_items = C.c_int64(0), C.c_int32(0)
    check(lib.csrk_spmv_plan_info(hp, C.byref(n_tiles), C.byref(tile_items)))

    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz / (elapsed / args.steps) / 1e9
    st = (C.c_int64 * 34)()
    check(lib.csrk_spmv_plan_stats(hp, st, 34))
    if op_handles:
        # the roofline block describes the first chunk's handle (the kernels that were timed)
        i_r, i_c, i_n, i_p, i_v = C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int(0), C.c_int(0)
        check(lib.csrk_info(hp, C.byref(i_r), C.byref(i_c), C.byref(i_n), C.byref(i_p), C.byref(i_v)))
        n_loc, nnz_loc = i_r.value, i_n.value
    n_heavy, nnz_path = int(st[2]), int(st[3])
    # Algorithmic bytes of ONE launch of each streaming kernel on this rank:
    # colinds 4 B + values 8 B per entry it processes; the tile kernel also reads one row pointer and
    # writes one y entry per row; a panel kernel reads one 4-B row pointer and writes one 8-B partial per
    # (column block, row) pair of its tier; each kernel reads x once.
    # st[20]: the short rows run as the light stream (else the merge-path tile kernel); tier 0 is in accumulator form
    # (st[9] = its rows; one row pointer + one y entry each).
    if int(st[20]):
        light = {'kernel': 'spmv_lstream_kernel', 'role': 'short rows: one wavefront per 512-entry tile of the light stream'}
    else:
        light = {'kernel': f'spmv_{algo_name}_kernel', 'role': 'merge-path tiles'}
    # algorithmic_bytes: SURVEY.md section 8(d)'s per-unit figures (12 B per entry, one row pointer + one y entry per row,
    # x once) for the part of the matrix the kernel serves.  stream_bytes: what the kernel's PRIVATE stream really holds
    # for those entries (tier 0: 10 B per entry + 4 B per 512-entry tile, one 32-KiB x window per segment, its partial
    # sums; the others 12 B per entry) -- the algorithmic rate of a kernel whose stream is narrower than the CSR arrays
    # can exceed the copy rate, its stream rate cannot.
    whole_bytes = nnz_loc * 12 + (n_loc + 1) * rp.element_size() + n_loc * 8 + ncols * 8
    kernels = [dict(light, ms=k_ms2[0], events=k_src[0], entries=nnz_path,
                    algorithmic_bytes=nnz_path * 12 + (n_loc + 1) * rp.element_size() + n_loc * 8 + ncols * 8,
                    stream_bytes=nnz_path * 12 + int(st[22]) * 4 + n_loc * 8 + int(st[24]) * 8)]
    if int(st[10]):
        t0 = {'kernel': 'spmv_acc_kernel', 'role': 'tier 0 (longest rows): x window + one accumulator per row in LDS, 10-B entries (f64 value + 16-bit column/row-step word)'}
        t0_stream = int(st[4]) * 512 * 10 + int(st[4]) * 4 + ncols * 8 + 256 * int(st[9]) * 8
        kernels.append(dict(t0, ms=k_ms2[1], events=k_src[1], entries=int(st[10]),
                            algorithmic_bytes=int(st[10]) * 12 + int(st[9]) * 12 + ncols * 8, stream_bytes=t0_stream))
    if int(st[13]):
        kernels.append({'kernel': 'spmv_panel_kernel<tier1>', 'role': 'tier 1 (mid rows): (block, row) pairs, x window in L2',
                        'ms': k_ms2[2], 'events': k_src[2], 'entries': int(st[13]),
                        'algorithmic_bytes': int(st[13]) * 12 + int(st[12]) * 12 + ncols * 8,
                        'stream_bytes': int(st[13]) * 12 + int(st[12]) * 12 + ncols * 8})
    if int(st[24]):
        # the cold-staging pass: no algorithmic bytes of its own (it re-orders x for the light stream); listed so that
        # the kernels add up to the SpMV, never the dominant one unless it really is the slowest
        kernels.append({'kernel': 'ls_stage_kernel', 'role': 'cold staging: x of the unpacked and packed columns copied into the '
                        'light stream\'s order through LDS windows (overhead pass)', 'ms': k_ms2[3], 'events': k_src[3], 'entries': int(st[24]),
                        'algorithmic_bytes': 0, 'stream_bytes': (int(st[24]) + int(st[16])) * 14 + ncols * 8})
    traffic_all = {}
    traffic_src = None
    for k in kernels:
        k['achieved_gbs'] = round(k['algorithmic_bytes'] / (k['ms'] * 1e-3) / 1e9, 1) if k['ms'] > 0 else 0.0
        k['stream_gbs'] = round(k['stream_bytes'] / (k['ms'] * 1e-3) / 1e9, 1) if k['ms'] > 0 else 0.0
        k['ms'] = round(k['ms'], 4)
        tr, src = load_traffic(args.traffic_json, workload, k['kernel']) if world == 1 else (None, None)
        k['traffic'] = tr
        if tr:
            traffic_src = src
            traffic_all[k['kernel']] = tr
            k['traffic_gbs'] = round(tr / (k['ms'] * 1e-3) / 1e9, 1) if k['ms'] > 0 else 0.0
            k['traffic_over_algorithmic'] = round(tr / k['algorithmic_bytes'], 3) if k['algorithmic_bytes'] else None
    # the dominant kernel: the slowest one (they run one after the other on one stream)
    dom = max(kernels, key=lambda k: k['ms'])
    k_sum_ms = sum(k['ms'] for k in kernels)
    frac_step = round(whole_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
    notes = []
    if dom['stream_gbs'] > 6290.0 * 1.1:
        notes.append('accounting check: the dominant kernel\'s own-stream rate exceeds the measured copy rate by more than 10 %')
    roofline = {
        'bound': 'hbm', 'kernel': dom['kernel'], 'role': dom['role'], 'achieved': dom['achieved_gbs'], 'peak': HBM_PEAK_GBS,
        'unit': 'GB/s', 'frac': round(dom['achieved_gbs'] / HBM_PEAK_GBS, 4),
        'traffic': dom['traffic'],
        # where `traffic` comes from: HBM counters cannot be read inside this run; they are the per-launch means of separate
        # rocprofv3 --pmc passes over this same command (tools/collect_profiles.sh), committed under profiles/
        'traffic_source': (traffic_src + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)')
                          if dom['traffic'] else None,
        # HIP events on the launch stream: the dominant kernel inside the timed region; the other kernels during the warm-up steps
        'kernel_ms': dom['ms'], 'kernel_ms_events': dom.get('events'), 'launches_timed': n_rec.value if dom.get('events') == 'timed region' else n_warm,
        'algorithmic_bytes': dom['algorithmic_bytes'],
        # the same kernel priced by the bytes its private stream holds and by the counter traffic: neither can exceed
        # the copy rate (6.29 TB/s measured, MI355X_MICROARCH.md) by much; the algorithmic rate above can, when the
        # stream is narrower than the CSR arrays it replaces
        'stream_bytes': dom['stream_bytes'], 'achieved_stream_gbs': dom['stream_gbs'],
        'frac_stream_of_measured_copy_peak_6290': round(dom['stream_gbs'] / 6290.0, 4),
        # ... and by the counter traffic over this run's kernel time: what the kernel really moves, against the 8 TB/s spec
        'frac_by_counter_traffic': round(dom['traffic'] / (dom['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                   if dom['traffic'] and dom['ms'] else None,
        # THE number BASELINE.json asks for: the whole 2.60 GB SpMV over the step (wall: small kernels and launch gaps
        # included), as a fraction of the 8 TB/s roofline; target >= 0.60
        'frac_whole_spmv_over_step': frac_step,
        'whole_spmv_ms': round(ms_per_step, 4), 'whole_spmv_algorithmic_bytes': whole_bytes,
        'target_frac': 0.60, 'target_met': bool(frac_step >= 0.60),
        'all_kernels': kernels,
        'whole_spmv': {'algorithmic_bytes': whole_bytes,
                       'streaming_kernels_ms': round(k_sum_ms, 4),
                       'frac_over_streaming_kernels': round(whole_bytes / (k_sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_sum_ms > 0 else None,
                       'frac_over_step': frac_step,
                       'traffic_bytes_all_kernels': sum(traffic_all.values()) if traffic_all else None},
        'notes': notes,
    }

    out = {
        'metric': 'spmv_gflops', 'value': round(gflops, 2), 'unit': 'GFLOP/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': workload, 'nrows': nrows, 'ncols': ncols, 'nnz': nnz,
                   'row_degree': f'power-law alpha={args.alpha}, max {int(min(1_000_000, ncols // 8))}',
                   'columns': 'Zipf(1.0) popularity over a permuted column space, distinct+sorted per row',
                   'algo': algo_name, 'tile_items': tile_items.value, 'tiles': n_tiles.value,
                   'rows_in_panels': n_heavy, 'tier0': {'min_entries': int(st[6]), 'column_block': int(st[7]), 'entries': int(st[10]), 'pairs': int(st[9]),
                             'accumulator_workgroups': int(st[27])},
                   'tier1': {'min_entries': int(st[14]), 'column_block': int(st[15]), 'rows': int(st[11]), 'entries': int(st[13]), 'pairs': int(st[12])},
                   'hot_column_cache': {'columns': int(st[16]), 'entry_share_sampled': round(int(st[17]) / 1e6, 4), 'slots': int(st[19])},
                   'cold_staged_entries': int(st[24]),
                   'parallelism': f'row-partition x{world}',
                   'collective': collective},
        'hbm_gbs_end_to_end': round((nnz * 12 + (nrows + 1) * 4 + nrows * 8 + ncols * 8) / (elapsed / args.steps) / 1e9, 1),
        # the whole SpMV against the roofline: algorithmic bytes of one product / step time / 8 TB/s (north_star's 0.60 target;
        # roofline.frac below is the dominant kernel's own)
        'whole_spmv_frac_of_hbm_roofline': (roofline or {}).get('frac_whole_spmv_over_step'),
        'roofline': roofline,
        'gen_seconds': round(t_gen, 2),
        # one-off costs on this handle, outside the timed region: the first call (plan-less kernel) and the second
        # (which builds the SpMV plan before launching)
        'first_call_ms': None if first_ms is None else round(first_ms, 2),
        'plan_build_call_ms': None if plan_ms is None else round(plan_ms, 2),
        'setup_check': {'bitwise_reproducible_products': REPRO_CALLS, 'ok': reproducible},
        # device memory: the CSR arrays this rank holds, and the SpMV plan built beside them (private streams of the
        # three tiers, cold-staging lists, tables)
        'matrix_bytes': nnz_loc * 12 + (n_loc + 1) * rp.element_size(), 'plan_bytes': int(st[25]),
        'plan_over_matrix': round(int(st[25]) / max(1, nnz_loc * 12 + (n_loc + 1) * rp.element_size()), 3),
        # where the plan's bytes are: the private streams re-state the matrix almost 1 : 1 -- tier 0 in 10 B
        # per entry, tier 1 and the light stream in 12 -- because the kernels read THEM, never the CSR arrays, after the
        # first product; the arrays stay for export, the other operations and plan-less algorithms
        'plan_bytes_by_part': {'tier0_accumulator_stream': int(st[29]), 'tier1_pair_panel': int(st[30]), 'light_stream': int(st[31]),
                               'cold_staging_and_pack': int(st[32]), 'tables': int(st[33])},
    }
    if compute_ms is not None:
        # per step: the slowest rank's local SpMV (device events) and what the exchange adds on top
        out['multi_gpu'] = {'exchange': collective, 'fallback': fallback_note, 'y_complete_and_identical_on_every_rank': exchange_ok, 'candidates_ms_per_step': calibration,
                            'local_spmv_ms_max_over_ranks': round(compute_ms, 4),
                            'exchange_ms': round(ms_per_step - compute_ms, 4),
                            'kernel_only_gflops': round(2.0 * nnz / (compute_ms * 1e-3) / 1e9, 1) if compute_ms > 0 else None,
                            # kernel-only and end-to-end rates side by side (the driver computes
                            # the scaling efficiency itself from `value` at each N; these are the two numerators)
                            'end_to_end_gflops': round(gflops, 1),
                            'backend': dist.get_backend() if distd else None}
        if world == 1:
            # one-rank rehearsal: the exchanged y must be the plain product, bit for bit
            y_plain = torch.empty_like(y)
            local(x, y_plain[:n_loc])
            torch.cuda.synchronize()
            out['multi_gpu']['y_equals_plain_product_bitwise'] = bool(torch.equal(y.view(torch.int64), y_plain.view(torch.int64)))

    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O       # the checker / baseline, never the thing measured above
        rp_h, ci_h, vs_h = rp.cpu().numpy(), ci.cpu().numpy(), vs.cpu().numpy()
        x_h = x.cpu().numpy()
        y_gpu = y.cpu().numpy()
        y_cpu = O.mult_vec(nrows, ncols, rp_h, ci_h, vs_h, x_h)          # warm-up + parity reference
        reps, t_cpu = 0, 0.0
        while reps < 1 or (t_cpu < args.cpu_seconds and reps < 10):
            t1 = time.perf_counter()
            O.mult_vec(nrows, ncols, rp_h, ci_h, vs_h, x_h)
            t_cpu += time.perf_counter() - t1
            reps += 1
        bound = O.mult_vec(nrows, ncols, rp_h, ci_h, np.abs(vs_h), np.abs(x_h))
        err = np.abs(y_gpu - y_cpu)
        worst = float(np.max(err / (bound + 1e-300)))
        out['cpu_baseline'] = {
            'value': round(2.0 * nnz / (t_cpu / reps) / 1e9, 3), 'unit': 'GFLOP/s', 'cores': 1, 'kind': 'port',
            'sample': f'the full {nrows}x{ncols} nnz={nnz} matrix, {reps} timed passes of orc_mult_vec_i32 '
                      f'({t_cpu / reps:.2f} s each)',
            'host_cpus': os.cpu_count(),
        }
        # two more CPU figures, clearly NOT the reference's kernel: the same row loop shared out
        # over this process's CPUs with OpenMP (bit-identical result), and SciPy's single-threaded csr_matvec
        also = {}
        try:
            nthr = len(os.sched_getaffinity(0))
            try:      # a container's CPU share (cgroup v2 quota), when it is smaller than the affinity mask
                q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
                if q != 'max':
                    nthr = max(1, min(nthr, int(int(q) / int(per))))
            except (OSError, ValueError):
                pass
            nthr = min(nthr, int(os.environ.get('BENCH_CPU_THREADS', '16')))      # a GPU box gives one GPU 16 CPUs' worth
            y_par = O.mult_vec_rows_parallel(nrows, ncols, rp_h, ci_h, vs_h, x_h, nthr)      # warm-up
            t1 = time.perf_counter()
            for _ in range(3):
                O.mult_vec_rows_parallel(nrows, ncols, rp_h, ci_h, vs_h, x_h, nthr)
            t_par = (time.perf_counter() - t1) / 3
            also['openmp_row_parallel_port'] = {'value': round(2.0 * nnz / t_par / 1e9, 3), 'unit': 'GFLOP/s', 'cores': nthr,
                                                'bit_identical_to_port': bool(np.array_equal(y_par, y_cpu))}
        except Exception as e:
            also['openmp_row_parallel_port'] = {'error': str(e)[:200]}
        try:
            import scipy.sparse as sps
            A_sp = sps.csr_matrix((vs_h, ci_h, rp_h), shape=(nrows, ncols))
            A_sp @ x_h
            t1 = time.perf_counter()
            y_sp = A_sp @ x_h
            t_sp = time.perf_counter() - t1
            also['scipy_csr_matvec'] = {'value': round(2.0 * nnz / t_sp / 1e9, 3), 'unit': 'GFLOP/s', 'cores': 1,
                                        'max_abs_diff_vs_port_over_bound': float(np.max(np.abs(y_sp - y_cpu) / (bound + 1e-300)))}
            del A_sp
        except Exception as e:
            also['scipy_csr_matvec'] = {'error': str(e)[:200]}
        out['cpu_baseline']['also_not_the_reference'] = also
        out['parity'] = {'max_abs_err_over_sum_abs_terms': worst, 'tolerance': 1e-6, 'ok': bool(worst <= 1e-6),
                         'rows_bit_identical': float(np.mean(y_gpu == y_cpu))}
        if worst > 1e-6:
            print(json.dumps(out))
            sys.exit('PARITY FAILURE: GPU result differs from the oracle')

    # (the timed handles go first: their plans' buffers return to the library's pool, which the second handle below draws on)
    for hc in op_handles:
        check(lib.csrk_free(hc))
    check(lib.csrk_free(h))
    if world == 1 and not distd:
        # What a caller WITHOUT a kept handle gets (the reference makes one per product, csr/csr.py:580-583): the first
        # product on a fresh handle runs the plan-less merge-path kernel over the CSR arrays as they are.  Timed here on a
        # second handle over the same resident arrays (kernels already loaded), outside the timed region.
        h2 = handle_t(0)
        check(lib.csrk_create_device(n_loc, ncols, nnz_loc, rp.data_ptr(), int(rp.dtype == torch.int64),
                                     ci.data_ptr(), vs.data_ptr(), 2, C.byref(h2)))
        y2 = torch.empty(n_loc, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.csrk_spmv_device(h2.value, x.data_ptr(), y2.data_ptr(), None))
        e1.record()
        torch.cuda.synchronize()
        pl_ms = e0.elapsed_time(e1)
        # ... and its second product builds the plan again, this time out of the library's pool: what the builders themselves
        # cost (the figure above, taken in a fresh process, also holds the runtime's first allocations -- a hipMalloc of tier
        # 0's 1 GB array has been seen to take 30-40 ms in some processes and 0.03 in others: DESIGN.md section 4)
        t_w = time.perf_counter()
        check(lib.csrk_spmv_device(h2.value, x.data_ptr(), y2.data_ptr(), None))
        torch.cuda.synchronize()
        plan2_ms = (time.perf_counter() - t_w) * 1e3
        check(lib.csrk_free(h2))
        del y2
        roofline['without_a_plan'] = {
            'what': 'first product on a handle: spmv_merge_kernel on the raw CSR arrays, x gathered from HBM / L2',
            'ms': round(pl_ms, 4), 'frac_whole_spmv': round(whole_bytes / (pl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'plan_build_call_ms': None if plan_ms is None else round(plan_ms, 2), 'plan_bytes': int(st[25]),
            'planned_products_to_amortise_the_plan': None if plan_ms is None or pl_ms <= ms_per_step else
            int(np.ceil(plan_ms / (pl_ms - ms_per_step))),
            'plan_build_call_ms_from_the_pool': round(plan2_ms, 2),
            'planned_products_to_amortise_it_from_the_pool': None if pl_ms <= ms_per_step else
            int(np.ceil(plan2_ms / (pl_ms - ms_per_step)))}
    if world == 1 and not distd and not args.no_secondary and args.scale == 1.0:
        # the other BASELINE configs, AFTER the SpMV has been timed and checked (nothing above depends on this):
        # each entry carries ms, algorithmic bytes, frac of 8 TB/s, a parity flag and the oracle's time on a stated sample
        # (a Python error anywhere in here must not cost the line that has been timed and verified above)
        # It runs as a CHILD process under a time limit: a hang or a GPU fault among its five kernel families cannot cost
        # the line above, which this process still holds (never a re-exec: a child, started after this process's own GPU
        # work is done and its matrix freed).
        import subprocess
        del y, rp, ci, vs, x
        try:
            check(lib.csrk_trim_cache())
            torch.cuda.empty_cache()
        except Exception:                     # noqa: BLE001
            pass
        cmd = [sys.executable, os.path.join(ROOT, 'bench_secondary.py'), '--product-ms', repr(float(ms_per_step))]
        try:
            child = subprocess.Popen(cmd, stdout=subprocess.PIPE, cwd=ROOT)
            try:
                so, _ = child.communicate(timeout=args.secondary_seconds)
                lines = [ln for ln in so.decode(errors='replace').splitlines() if ln.startswith('{')]
                if child.returncode == 0 and lines:
                    out['secondary'] = json.loads(lines[-1])
                else:
                    out['secondary'] = {'error': f'the secondary process ended with code {child.returncode} and {len(lines)} result line(s)'}
            except subprocess.TimeoutExpired:
                child.kill()
                child.communicate()
                out['secondary'] = {'error': f'the secondary process was stopped after {args.secondary_seconds:.0f} s'}
        except Exception as e:                # noqa: BLE001
            out['secondary'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distd:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
