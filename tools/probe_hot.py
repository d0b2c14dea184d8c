#!/usr/bin/env python3
"""Sweep the hot-column pack size (CSRK_HOT_SLOTS) on the headline matrix: one subprocess per setting."""
import json, os, subprocess, sys
for slots in sys.argv[1:] or ['0', '65536', '262144', '1048576']:
    env = dict(os.environ)
    if slots == '0':
        env['CSRK_SPMV_HOT'] = '0'
    else:
        env['CSRK_HOT_SLOTS'] = slots
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '30', '--no-cpu-baseline'], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print(slots, d['value'], d['ms_per_step'], d['config'].get('hot_column_cache'), d.get('parity', {}).get('ok'),
          [(k['kernel'][:24], k['ms']) for k in d['roofline']['all_kernels']], flush=True)
