#!/bin/bash
# scratch (GPU box): memory-side counters for the SpMV kernels; usage: tools/pmc_mem.sh TAG [ENV=VAL ...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
rm -rf $OUT; mkdir -p $OUT
for e in "$@"; do export "$e"; done
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; }
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
        if v: print(f'   {c:34s} {sum(v)/len(v):.4g}')
PY
