set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/gpu_tests.log; echo "rc=$?" >> gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b_final.log 2>&1; tail -1 gpurun_out/b_final.log | cut -c1-600
