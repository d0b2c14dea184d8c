#!/bin/bash
# GPU box: where a plan build's 17 ms go.  One rocprofv3 kernel trace of tools/sweep_inproc.py (10 steps); the kernels between
# the plan-less product and the first planned one are listed in time order with the idle time before each (host work, syncs).
#   tools/probe_plan_build.sh [OUT]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/planbuild}
rm -rf $OUT; mkdir -p $OUT
SWEEP_STEPS=10 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 tools/sweep_inproc.py "" > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name']
i0 = next(i for i, r in enumerate(rows) if 'csrk::spmv_merge_kernel' in name(r))
i1 = next(i for i, r in enumerate(rows) if 'csrk::spmv_acc_kernel' in name(r))
seg = rows[i0:i1 + 1]
t0 = int(seg[0]['Start_Timestamp'])
prev_end = None
busy = 0
agg = {}
print(f'{"t_us":>9s} {"gap_us":>8s} {"dur_us":>9s}  kernel')
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    n = name(r).split('(')[0].replace('void ', '').replace('csrk::', '')[:70]
    print(f'{(s - t0) / 1e3:9.1f} {gap:8.1f} {(e - s) / 1e3:9.1f}  {n}')
    busy += e - s
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev_end = e if prev_end is None else max(prev_end, e)
span = (int(seg[-1]['Start_Timestamp']) - t0) / 1e3
print(f'span {span:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {span - busy / 1e3 + (int(seg[-1]["End_Timestamp"]) - int(seg[-1]["Start_Timestamp"])) / 1e3:.1f} us')
print('-- by kernel')
for n, (c, u) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{u:9.1f} us {c:4d} x  {n}')
PY
