#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>cfg/ (tools/collect_profiles_configs.sh) into committed files under profiles/."""
import collections, csv, glob, json, os, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = f'gpurun_out/{tag}cfg'
keep = ('csrk::spmm', 'csrk::mm_', 'csrk::rx_', 'csrk::rowptr_from', 'csrk::sg_', 'csrk::so_', 'csrk::row_')
with open(f'profiles/{tag}_configs.json', 'w') as f:      # the plain (unprofiled) run's lines
    for ln in open(f'{src}/plain.log'):
        if ln.startswith('{'):
            f.write(ln)
rows = list(csv.DictReader(open(newest(f'{src}/kt/*/*_kernel_stats.csv'))))
with open(f'profiles/{tag}_configs_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    cols = ['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev']
    w.writerow(cols)
    for r in rows:
        if any(k in r['Name'] for k in keep):
            w.writerow([r[k] for k in cols])


def pmc(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f'{src}/{d}/*/*_counter_collection.csv'))):
        if any(k in r['Kernel_Name'] for k in keep):
            name = r['Kernel_Name'].split('csrk::')[1].split('(')[0]
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}

allc = {}
for d in ('fetch', 'write'):
    allc.update(pmc(d))
traffic = {}
for (k, c), v in allc.items():
    traffic[k] = traffic.get(k, 0.0) + v * 1024.0 * (2.0 if c == 'FETCH_SIZE' else 1.0)   # KB; 128-B reads tallied as 64 B
other = {}
for d in ('mfma', 'tcp'):
    try:
        for (k, c), v in pmc(d).items():
            other.setdefault(k, {})[c] = round(v)
    except (ValueError, OSError):
        pass
json.dump({'workload': 'tools/bench_configs.py all (unit_rows on the headline matrix, configs[2] SpMM, configs[4] transpose + A B^T blocks, power-law A B)',
           'hbm_bytes_per_launch': {k: round(v) for k, v in sorted(traffic.items())},
           'counters_per_launch': other,
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace only; mean per launch; '
                     'read bytes = 2 x FETCH_SIZE x 1024 (128-B requests tallied at 64 B on gfx950), writes = WRITE_SIZE x 1024'},
          open(f'profiles/{tag}_configs_pmc_traffic.json', 'w'), indent=1)
print(json.dumps({k: round(v / 1e6, 1) for k, v in sorted(traffic.items())}))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from record_tree import record      # noqa: E402
record(tag, 'configs', src)
