timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_random_sweep.py -x -q -m gpu -k "spgemm or mult or abt or ab or multiply" 2>&1 | tail -4 && \
bash tools/kstats_configs.sh ks_ab2 ab 2>&1 | grep -v "^E2026\|^W2026" | grep "config\|sg_list_rows\|sg_count" && \
python tools/scratch/abt_small.py 2>&1 | grep abt
