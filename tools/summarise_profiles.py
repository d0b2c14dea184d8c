#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/collect_profiles.sh) into committed files under profiles/."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    "gpurun merges into gpurun_out/ without deleting earlier files: take the most recent match"
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = f'gpurun_out/{tag}'
os.makedirs('profiles', exist_ok=True)
bench = json.loads(open(f'{src}/bench.json').read().strip().splitlines()[-1])
# kernel stats: keep our kernels + the total
rows = list(csv.DictReader(open(newest(f'{src}/kt/*/*_kernel_stats.csv'))))
with open(f'profiles/{tag}_spmv_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
    for r in rows:
        if 'csrk::' in r['Name']:
            w.writerow([r[k] for k in ['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev']])
def pmc(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f'{src}/{d}/*/*_counter_collection.csv'))):
        if 'csrk::spmv' in r['Kernel_Name'] or 'csrk::panel' in r['Kernel_Name'] or 'csrk::acc_' in r['Kernel_Name'] \
                or 'csrk::hot_pack' in r['Kernel_Name'] or 'csrk::ls_stage_kernel' in r['Kernel_Name']:
            full = r['Kernel_Name'].split('csrk::')[1].split('(')[0]
            name = full.split('<')[0]
            if name == 'spmv_panel_kernel':      # (bench.py's name for the pair kernel)
                name += '<tier1>'
            agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}
allc = {}
for d in ('fetch', 'write', 'tcc', 'tcp'):
    allc.update(pmc(d))
with open(f'profiles/{tag}_spmv_pmc_counters.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'counter', 'mean_per_launch'])
    for (k, c), v in sorted(allc.items()):
        w.writerow([k, c, f'{v:.6g}'])
traffic = {}
for (k, c), v in allc.items():
    if c in ('FETCH_SIZE', 'WRITE_SIZE'):
        traffic[k] = traffic.get(k, 0.0) + v * 1024.0 * (2.0 if c == 'FETCH_SIZE' else 1.0)   # KB; 128-B reads tallied as 64 B
out = {'workload': bench['config']['workload'], 'hbm_bytes_per_launch': traffic,
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slots), each with '
                 '--kernel-trace only; mean over the recorded launches. '
                 'FETCH_SIZE tallies every L2->fabric read request at 64 B, but on gfx950 the vector path issues 128-B '
                 'requests (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ for these kernels, measured), so read bytes = 2 x '
                 'FETCH_SIZE x 1024 (the correction of MI355X_MICROARCH.md, HBM section); writes = WRITE_SIZE x 1024.'}
json.dump(out, open(f'profiles/{tag}_spmv_pmc_traffic.json', 'w'), indent=1)
# the bench line of the same collection run, with the traffic these passes measured for its dominant kernel
if bench['roofline']['kernel'] in traffic:      # (the line was printed before this summary existed: it cites the round before)
    bench['roofline']['traffic'] = traffic[bench['roofline']['kernel']]
    bench['roofline']['traffic_source'] = f'profiles/{tag}_spmv_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same collection run)'
    for k in bench['roofline'].get('all_kernels', []):
        if k.get('kernel') in traffic:
            k['traffic'] = traffic[k['kernel']]
            k['traffic_gbs'] = round(k['traffic'] / (k['ms'] * 1e-3) / 1e9, 1) if k.get('ms') else None
            if k.get('algorithmic_bytes'):
                k['traffic_over_algorithmic'] = round(k['traffic'] / k['algorithmic_bytes'], 3)
    if 'whole_spmv' in bench['roofline']:
        bench['roofline']['whole_spmv']['traffic_bytes_all_kernels'] = sum(
            v for k, v in traffic.items() if k in ('ls_stage_kernel', 'spmv_acc_kernel', 'spmv_panel_kernel<tier1>', 'spmv_lstream_kernel', 'spmv_epilogue_kernel'))
open(f'profiles/{tag}_bench.json', 'w').write(json.dumps(bench) + '\n')
print(json.dumps(out['hbm_bytes_per_launch']))
print({k: v for k, v in bench['roofline'].items() if k != 'all_kernels'})
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from record_tree import record      # noqa: E402
record(tag, 'spmv', src)
