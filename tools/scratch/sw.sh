export TMPDIR=/tmp
python tools/probe_rank.py 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/ks_rank8; mkdir -p gpurun_out/ks_rank8
WORLDS=8 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_rank8/kt -- python3 tools/probe_rank.py > gpurun_out/ks_rank8/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ks_rank8/kt/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'csrk::' in r['Name'] and int(r['Calls']) >= 30:
        print(f"{r['Name'].split('csrk::')[1][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f}")
PY
