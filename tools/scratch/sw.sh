export SWEEP_STEPS=200
for v in rndnt ynt bothnt; do CSRK_LIBRARY=csr_amd/libcsrk_$v.so timeout -k 10 300 python tools/sweep_inproc.py "" 2>&1 | grep defaults | sed "s/^/[$v] /"; done
timeout -k 10 300 python tools/sweep_inproc.py "" 2>&1 | grep defaults
