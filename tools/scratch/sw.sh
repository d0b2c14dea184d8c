timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "unit or center" 2>&1 | tail -5 && \
ROWOPS_SHAPE=both bash tools/kstats_rowops.sh ks_cf3 2>&1 | grep -v "^E2026\|^W2026" | tail -24
