for lib in csr_amd/libcsrk.so csr_amd/libcsrk_nt2.so; do
  echo "== $lib"
  CSRK_LIBRARY=$PWD/$lib python tools/bench_configs.py spmm 2>&1 | grep -o '"ms": [0-9.]*' | head -1
  CSRK_LIBRARY=$PWD/$lib ROWOPS_SHAPE=headline python tools/probe_rowops.py 2>&1 | grep headline | cut -c1-60
done
