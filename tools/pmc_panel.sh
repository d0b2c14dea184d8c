#!/bin/bash
# GPU box: TCP_TCC_READ_REQ / cache accesses of the headline SpMV's kernels under one library build.
#   tools/pmc_panel.sh TAG [LIB]
set -u
TAG=${1:-pmc}
export TMPDIR=/tmp
[ -n "${2:-}" ] && export CSRK_LIBRARY=$2
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
SWEEP_STEPS=20 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tcp -- python3 tools/sweep_inproc.py "" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/tcp/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'csrk::spmv' not in k and 'ls_stage' not in k: continue
    k = k.split('csrk::')[1][:50]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
