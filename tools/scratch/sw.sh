export SWEEP_STEPS=200
for v in sa sb sc sd; do CSRK_LIBRARY=csr_amd/libcsrk_$v.so timeout -k 10 300 python tools/sweep_inproc.py "" 2>&1 | grep defaults | sed "s/^/[$v] /" | cut -c1-110; done
timeout -k 10 300 python tools/sweep_inproc.py "" 2>&1 | grep defaults | cut -c1-110
