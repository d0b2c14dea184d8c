timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "unit or center" 2>&1 | tail -8
