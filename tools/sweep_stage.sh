#!/bin/bash
# scratch (GPU box): env sweeps of bench.py
O=gpurun_out/stage; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 50 --warmup 5 --cpu-seconds 1 > $O/$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.log") if l.startswith("{")][-1])
    print("$tag", d["ms_per_step"], [(k["kernel"][5:12],k["ms"]) for k in d["roofline"]["all_kernels"]], d["parity"]["ok"], d["config"]["tier1"]["entries"], d["config"].get("cold_staged_entries"), flush=True)
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.log").read()[-1500:])
PY
}
run not1 CSRK_TIERB_MIN=0
run t1_512 CSRK_TIERB_MIN=512
run t1_384 CSRK_TIERB_MIN=384
