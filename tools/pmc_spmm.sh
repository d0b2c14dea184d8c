#!/bin/bash
# scratch (GPU box): memory counters for the dense-panel SpMM kernels (BASELINE configs[2])
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_spmm
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/bench_configs.py spmm > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'csrk::spmm' not in n and 'csrk::mm_' not in n: continue
        k = n.split('csrk::')[1].split('(')[0][:40]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()): print(f'{k:42s} {c:30s} {sum(v)/len(v):.4g}  (n={len(v)})')
PY
