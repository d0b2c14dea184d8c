#!/bin/bash
# scratch (GPU box): SQ counters for the SpMV kernels; usage: tools/pmc_light.sh TAG [ENV=VAL ...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
for e in "$@"; do export "$e"; done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'csrk::' not in n or not any(t in n for t in ('spmv', 'ls_stage', 'acc_reduce')): continue
        k = n.split('csrk::')[1].split('(')[0][:40]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
ks = sorted(set(k for k, _ in agg))
cs = sorted(set(c for _, c in agg))
for k in ks:
    print(k)
    for c in cs:
        v = agg.get((k, c))
        if v: print(f'   {c:28s} {sum(v)/len(v):.4g}')
PY
