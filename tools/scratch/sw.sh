SG_NA=500 SG_NB=5000 CSRK_LIBRARY=csr_amd/libcsrk_sgst.so python tools/probe_sg_stamps.py 2>&1 | sed -n 2,8p
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_random_sweep.py -x -q -m gpu -k "spgemm or mult or abt or multiply" 2>&1 | tail -3
python tools/bench_configs.py abt 2>&1 | grep "^{" | cut -c1-200
