export SWEEP_STEPS=200
timeout -k 10 300 python tools/sweep_inproc.py "" > gpurun_out/sweep_g.log 2>&1
for v in w16 w12 w12b; do CSRK_LIBRARY=csr_amd/libcsrk_$v.so timeout -k 10 300 python tools/sweep_inproc.py "" 2>&1 | sed "s/^/[$v] /" >> gpurun_out/sweep_g.log; done
cat gpurun_out/sweep_g.log
