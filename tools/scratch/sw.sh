timeout -k 10 900 python -m pytest tests/test_gpu_spmv.py -x -q -m gpu 2>&1 | tail -4 && \
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], [(k['kernel'][5:12],k['ms']) for k in r['all_kernels']])"; done
