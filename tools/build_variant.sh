#!/bin/bash
# scratch: build csr_amd/libcsrk_<NAME>.so from the same sources with extra compiler flags (kernel experiments);
# use it with CSRK_LIBRARY=csr_amd/libcsrk_<NAME>.so.   usage: tools/build_variant.sh NAME -DFOO=1 ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
O=csr_amd/build_$NAME; mkdir -p $O
for f in csr_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -c $f -o $O/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o csr_amd/libcsrk_$NAME.so $O/*.o
echo csr_amd/libcsrk_$NAME.so
