#!/bin/bash
# scratch (GPU box): bench.py under a list of environment settings; usage: tools/sweep_env.sh "A=1 B=2" "A=3" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], [ (k['kernel'][:24], k['ms']) for k in d['roofline']['all_kernels']])"
done
