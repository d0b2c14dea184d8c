python -m pytest tests/test_gpu_spmv.py -m gpu -x -q > gpurun_out/t_spmv.log 2>&1; tail -2 gpurun_out/t_spmv.log
python tools/probe_rank.py 2>&1 | grep -o "world [0-9] rank 0.*ms \|tier0 [0-9.]*" | tr '\n' ' '; echo
SWEEP_STEPS=200 timeout -k 10 600 python tools/sweep_inproc.py "" "CSRK_ACC_WGS=254" "CSRK_ACC_WGS=248" 2>&1 | grep -v amdgpu | cut -c1-110
