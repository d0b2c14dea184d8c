#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) of the headline SpMV under one environment setting.
#   tools/kstats.sh TAG ["CSRK_X=1 CSRK_Y=2"]
set -u
TAG=${1:-ks}
CFG=${2:-}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
SWEEP_STEPS=${SWEEP_STEPS:-100} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/sweep_inproc.py "$CFG" > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'csrk::' in r['Name'] and int(r['Calls']) >= 50:
        print(f"{r['Name'].split('csrk::')[1][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f}")
PY
