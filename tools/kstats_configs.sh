#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/bench_configs.py WHAT (spmm | transpose | abt | all)
set -u
TAG=${1:-ks_cfg}
WHAT=${2:-all}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/bench_configs.py $WHAT > $OUT/run.log 2>&1
grep "^{" $OUT/run.log | cut -c1-400
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'csrk::' in r['Name'] and float(r['TotalDurationNs']) > 50e3:
        print(f"{r['Name'].split('csrk::')[1][:64]:64s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
PY
