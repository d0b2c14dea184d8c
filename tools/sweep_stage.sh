#!/bin/bash
# scratch (GPU box): env sweeps of bench.py
O=gpurun_out/stage; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/$tag.log 2>&1; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$tag.log") if l.startswith("{")][-1])
    print("$tag", d["ms_per_step"], [(k["kernel"][5:12],k["ms"]) for k in d["roofline"]["all_kernels"]], d["config"]["tier1"]["entries"], flush=True)
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.log").read()[-1500:])
PY
}
run t128 A=1
run t96 CSRK_TIERB_MIN=96
run t160 CSRK_TIERB_MIN=160
run t192 CSRK_TIERB_MIN=192
run p1m CSRK_HOT_SLOTS=1048576
run t128b A=1
