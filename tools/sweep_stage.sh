#!/bin/bash
# scratch (GPU box): env sweeps of bench.py
O=gpurun_out/stage; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --cpu-seconds 1 > $O/$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.log") if l.startswith("{")][-1])
    print("$tag", d["ms_per_step"], [(k["kernel"][5:12],k["ms"]) for k in d["roofline"]["all_kernels"]], d["config"].get("cold_staged_entries"), d["config"]["hot_column_cache"]["columns"], d["parity"]["ok"], d["parity"]["rows_bit_identical"], flush=True)
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.log").read()[-1500:])
PY
}
run win512k A=1
run nowin CSRK_LS_WINDOW=0
run win128k CSRK_HOT_SLOTS=131072
run win64k CSRK_HOT_SLOTS=65536
run win32k CSRK_HOT_SLOTS=32768
