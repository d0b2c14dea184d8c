for c in 1 0 1 0; do echo "conc $c"; CSRK_RS_CONCURRENT=$c ROWOPS_SHAPE=both timeout -k 10 120 python tools/probe_rowops.py 2>&1 | grep wall | cut -c1-60; done
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_random_sweep.py -x -q -m gpu -k "unit or center or sweep" 2>&1 | tail -3
