#!/bin/bash
# scratch (GPU box): fabric-side counters of the light path's kernels; usage: tools/pmc_stage.sh TAG [ENV=VAL ...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
for e in "$@"; do export "$e"; done
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; }
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'ls_stage' not in n and 'spmv_lstream' not in n and 'hot_pack' not in n: continue
        k = n.split('csrk::')[1].split('(')[0][:40]
        agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for k, c in sorted(agg):
    v = agg[(k, c)]
    print(f'{k:24s} {c:28s} {sum(v)/len(v):.4g}')
PY
