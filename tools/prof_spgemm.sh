#!/bin/bash
# scratch (GPU box): per-kernel times of the two power-law SpGEMM products and the MovieLens-shaped A B^T block, for the
# default library and the variants named on the command line (csr_amd/libcsrk_<NAME>.so)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sg
rm -rf $OUT; mkdir -p $OUT
for v in default "$@"; do
  if [ $v = default ]; then unset CSRK_LIBRARY; else export CSRK_LIBRARY=$GRAFT_REPO_ROOT/csr_amd/libcsrk_$v.so; fi
  for w in ${SG_WORKLOADS:-probe_spgemm_noorc probe_abt2}; do
    PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v-$w -- python $GRAFT_REPO_ROOT/tools/$w.py > $OUT/$v-$w.log 2>&1 || tail -3 $OUT/$v-$w.log
    echo "== $v $w"; grep " ms" $OUT/$v-$w.log
    python - <<PY
import csv, glob
for f in glob.glob('$OUT/$v-$w/*/*_kernel_stats.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'csrk::' in r['Name']]
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:12]:
        print('   %-60s calls %4s avg %9.1f us total %9.1f us' % (r['Name'].split('csrk::')[1][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
  done
done
