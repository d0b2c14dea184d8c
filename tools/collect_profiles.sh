#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and separate PMC passes for bench.py.
# Output: gpurun_out/$1/...   Summarise afterwards with tools/summarise_profiles.py.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
python3 tools/tree_stamp.py > $OUT/tree.txt      # the sources these profiles are taken from
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err      # the driver's command line (secondary block included)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --no-cpu-baseline --no-secondary > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/tcc -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/tcc.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tcp -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/tcp.log 2>&1
tail -c 400 $OUT/bench.json
