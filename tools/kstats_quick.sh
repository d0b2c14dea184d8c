#!/bin/bash
# GPU box: like kstats.sh, several settings in a row, epilogue/panel lines only.   tools/kstats_quick.sh "CFG1" "CFG2" ...
export TMPDIR=/tmp
i=0
for CFG in "$@"; do
  i=$((i+1)); OUT=gpurun_out/kq$i; rm -rf $OUT; mkdir -p $OUT
  env $CFG SWEEP_STEPS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/sweep_inproc.py "" > $OUT/run.log 2>&1
  echo "== $CFG"; tail -1 $OUT/run.log
  python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'csrk::' in r['Name'] and int(r['Calls']) >= 50:
        print(f"   {r['Name'].split('csrk::')[1][:40]:40s} avg_us {float(r['AverageNs'])/1e3:9.2f}")
PY
done
