#!/bin/bash
# round 4, call e: the whole GPU suite after the pruning + quick bench lines of the three models
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 ) 2>&1 | tail -3; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "host", round(d["host_enqueue_ms_per_step"], 2), "loss", round(d["loss"], 4))
print("   kernels", {k: v for k, v in list(d["kernel_breakdown_us_per_step"].items())[:22]})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $ARGS $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
ARGS="";               run nrms NR_X=0
ARGS="--model NAML";   run naml NR_X=0
ARGS="--model LSTUR";  run lstur NR_X=0
