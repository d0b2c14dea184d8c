timeout -k 10 300 python -m pytest tests/test_abi.py -x -q -m gpu 2>&1 | tail -5
