timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_random_sweep.py -x -q -m gpu -k "unit or center or sweep" 2>&1 | tail -4 && \
timeout -k 10 120 python tools/probe_rowops.py 2>&1 | tail -4
