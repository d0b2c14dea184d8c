#!/bin/bash
# scratch (GPU box): SQ / LDS counters for the SpGEMM kernels of the MovieLens-shaped A B^T block
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_abt
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/bench_configs.py abt > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'csrk::sg_strip' not in n and 'csrk::sg_lds' not in n: continue
        k = n.split('csrk::')[1].split('(')[0][:40]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()): print(f'{k:42s} {c:26s} {sum(v)/len(v):.4g}')
PY
