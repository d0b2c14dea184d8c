#!/bin/bash
# scratch (GPU box): env sweeps of bench.py
O=gpurun_out/stage; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.log") if l.startswith("{")][-1])
    print("$tag", d["ms_per_step"], [(k["kernel"][5:12],k["ms"]) for k in d["roofline"]["all_kernels"]], d["config"].get("cold_staged_entries"), d["config"]["hot_column_cache"]["columns"], flush=True)
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.log").read()[-1500:])
PY
}
run base A=1
run lsuc CSRK_LS_UC=1
run accuc CSRK_ACC_UC=1
run bothuc CSRK_LS_UC=1 CSRK_ACC_UC=1
run lsuc1m CSRK_LS_UC=1 CSRK_HOT_SLOTS=1048576
