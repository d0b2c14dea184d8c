#!/bin/bash
# scratch (GPU box): cold staging on/off, order of the staging pass, pack sizes
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
run off CSRK_LS_STAGE=0
run first CSRK_LS_STAGE=1
run late CSRK_LS_STAGE=1 CSRK_LS_STAGE_FIRST=0
run first1m CSRK_LS_STAGE=1 CSRK_HOT_SLOTS=1048576
run off2 CSRK_LS_STAGE=0
run first2 CSRK_LS_STAGE=1
