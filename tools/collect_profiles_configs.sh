#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and FETCH/WRITE PMC passes for
# tools/bench_configs.py (unit_rows on the headline matrix, BASELINE configs[2] dense-panel SpMM, configs[4] transpose + A B^T).
# Output: gpurun_out/$1cfg/...   Summarise with tools/summarise_profiles_configs.py.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}cfg
rm -rf $OUT; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/tree_stamp.py > $OUT/tree.txt      # the sources these profiles are taken from
cd /tmp
# a plain run first: its JSON lines are profiles/${TAG}_configs.json (timings without the profiler attached)
python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/write.log 2>&1
# matrix-core and L1-fill counters (north_star asks for the SpMM's MFMA utilisation from rocprof: the kernels issue none)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -- python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/mfma.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcp -- python $GRAFT_REPO_ROOT/tools/bench_configs.py all > $OUT/tcp.log 2>&1
grep -c config $OUT/kt.log
