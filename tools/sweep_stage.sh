#!/bin/bash
# scratch (GPU box): env sweeps of bench.py
O=gpurun_out/stage; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.log") if l.startswith("{")][-1])
    print("$tag", d["ms_per_step"], [(k["kernel"][5:12],k["ms"]) for k in d["roofline"]["all_kernels"]], flush=True)
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.log").read()[-1500:])
PY
}
run t1hot A=1
run t1off CSRK_T1_HOT=0
run t1hot_tpw2 CSRK_PANEL_TPW1=2
