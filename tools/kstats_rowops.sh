#!/bin/bash
# GPU box: per-kernel durations of unit_rows / center_rows (rocprofv3 --kernel-trace --stats over tools/probe_rowops.py)
set -u
TAG=${1:-ks_rowops}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/probe_rowops.py > $OUT/run.log 2>&1
grep -v "^W2026\|amdgpu.ids" $OUT/run.log | tail -8
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'csrk::' in r['Name']:
        print(f"{r['Name'].split('csrk::')[1][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.2f} max_us {float(r['MaxNs'])/1e3:10.2f}")
PY
