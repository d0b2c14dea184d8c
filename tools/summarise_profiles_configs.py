#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>cfg/ (tools/collect_profiles_configs.sh: one collection per config) into committed files under profiles/."""
import collections, csv, glob, json, os, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = f'gpurun_out/{tag}cfg'
keep = ('csrk::spmm', 'csrk::mm_', 'csrk::rx_', 'csrk::rowptr_from', 'csrk::sg_', 'csrk::so_', 'csrk::row_', 'csrk::dense_', 'csrk::hr_')
configs = [c for c in ('unit_rows', 'spmm', 'transpose', 'abt') if os.path.isdir(f'{src}/{c}')]
with open(f'profiles/{tag}_configs.json', 'w') as f:      # the plain (unprofiled) run's lines
    for ln in open(f'{src}/plain.log'):
        if ln.startswith('{'):
            f.write(ln)
cols = ['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev']
with open(f'profiles/{tag}_configs_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Config'] + cols)
    for c in configs:
        for r in csv.DictReader(open(newest(f'{src}/{c}/kt/*/*_kernel_stats.csv'))):
            if any(k in r['Name'] for k in keep):
                w.writerow([c] + [r[k] for k in cols])


def pmc(c, d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f'{src}/{c}/{d}/*/*_counter_collection.csv'))):
        if any(k in r['Kernel_Name'] for k in keep):
            name = r['Kernel_Name'].split('csrk::')[1].split('(')[0]
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

by_config, launches, other = {}, {}, {}
for c in configs:
    traffic, n = {}, {}
    for d in ('fetch', 'write'):
        for (k, cn), (v, cnt) in pmc(c, d).items():
            traffic[k] = traffic.get(k, 0.0) + v * 1024.0 * (2.0 if cn == 'FETCH_SIZE' else 1.0)   # KB; 128-B reads tallied as 64 B
            n[k] = cnt
    by_config[c] = {k: round(v) for k, v in sorted(traffic.items())}
    launches[c] = n
    for d in ('mfma', 'tcp'):
        try:
            for (k, cn), (v, _) in pmc(c, d).items():
                other.setdefault(c, {}).setdefault(k, {})[cn] = round(v)
        except (ValueError, OSError):
            pass
flat = {}
for c in configs:      # (bench_secondary.py reads the SpMM's kernels from this flat table)
    for k, v in by_config[c].items():
        flat.setdefault(k, v)
json.dump({'workload': 'tools/bench_configs.py, one rocprofv3 collection per config: ' + ', '.join(configs),
           'by_config': by_config, 'launches_averaged': launches,
           'hbm_bytes_per_launch': {k: flat[k] for k in sorted(flat)},
           'counters_per_launch': other,
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace only, per config; mean per launch; '
                     'read bytes = 2 x FETCH_SIZE x 1024 (128-B requests tallied at 64 B on gfx950), writes = WRITE_SIZE x 1024; '
                     'hbm_bytes_per_launch = the same numbers in one table (a kernel that runs under several configs: the first config\'s)'},
          open(f'profiles/{tag}_configs_pmc_traffic.json', 'w'), indent=1)
print(json.dumps({c: {k: round(v / 1e6, 1) for k, v in by_config[c].items()} for c in configs}))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from record_tree import record      # noqa: E402
record(tag, 'configs', src)
