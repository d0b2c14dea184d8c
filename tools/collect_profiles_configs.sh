#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and FETCH / WRITE PMC passes for tools/bench_configs.py, ONE
# COLLECTION PER CONFIG (unit_rows on the headline matrix, BASELINE configs[2] dense-panel SpMM, configs[4] transpose, A B^T),
# so that a kernel shared by several operations -- rx_scatter_kernel serves transposes, from_coo, SpGEMM's sorts and the SpMM
# plan -- is averaged over one workload's launches only.
#   tools/collect_profiles_configs.sh TAG [--only "spmm transpose"]
# Output: gpurun_out/${TAG}cfg/<config>/...   Summarise with tools/summarise_profiles_configs.py TAG.
set -u
TAG=${1:-r01}
CONFIGS="unit_rows spmm transpose abt"
if [ "${2:-}" = "--only" ]; then CONFIGS=$3; fi
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}cfg
rm -rf $OUT; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/tree_stamp.py > $OUT/tree.txt      # the sources these profiles are taken from
cd /tmp
# a plain run first: its JSON lines are profiles/${TAG}_configs.json (timings without the profiler attached)
python3 $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/plain.log 2>&1
for c in $CONFIGS; do
  mkdir -p $OUT/$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/kt -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c/kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$c/fetch -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$c/write -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c/write.log 2>&1
  if [ $c = spmm ]; then
    # matrix-core and L1-fill counters (north_star asks for the SpMM's MFMA utilisation from rocprof: the kernels issue none)
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$c/mfma -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c/mfma.log 2>&1
    rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/$c/tcp -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c/tcp.log 2>&1
  fi
  echo "$c: $(grep -c config $OUT/$c/kt.log) line(s)"
done
