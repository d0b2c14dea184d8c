#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the unit_rows / center_rows kernels, per shape
export TMPDIR=/tmp
OUT=gpurun_out/pmc_rowops
rm -rf $OUT; mkdir -p $OUT
for shape in headline ml; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ROWOPS_SHAPE=$shape rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${shape}_$c -- python3 tools/probe_rowops.py > $OUT/${shape}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for shape in ('headline', 'ml'):
    agg = collections.defaultdict(list)
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob(f'gpurun_out/pmc_rowops/{shape}_{c}/*/*_counter_collection.csv')[0]
        for r in csv.DictReader(open(f)):
            if 'csrk::row_' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('csrk::')[1].split('(')[0]
                agg[(k, c)].append(float(r['Counter_Value']))
    tr = {}
    for (k, c), v in agg.items():
        tr[k] = tr.get(k, 0.0) + sum(v) / len(v) * 1024.0 * (2.0 if c == 'FETCH_SIZE' else 1.0)
    out[shape] = {k: round(v) for k, v in sorted(tr.items())}
    out[shape + '_total_unit_rows'] = round(sum(v for k, v in tr.items() if 'true' in k or 'class_count' in k or 'chunk' in k))
json.dump({'hbm_bytes_per_launch': out,
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only over tools/probe_rowops.py; mean per launch; read bytes = 2 x FETCH_SIZE x 1024 (128-B requests tallied at 64 B on gfx950), writes = WRITE_SIZE x 1024',
           'algorithmic_bytes': {'headline': 2 * 200_000_000 * 8 + 10_000_000 * 12, 'ml': 2 * 25_000_095 * 8 + 162_541 * 12}},
          open('gpurun_out/pmc_rowops/r03_rowops_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(out)[:1500])
PY
