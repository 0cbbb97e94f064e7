#!/bin/bash
# round 4, call t: qkv_proj with a counted wait + raw barrier between weight chunks (NR_PROJ_RAWB=1) against __syncthreads()
export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
NR_PROJ_RAWB=1 timeout 600 python -m pytest tests/test_proj_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "proj or golden_base or mind_shape or dropout_matches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for R in 0 1 0 1; do echo -n "NR_PROJ_RAWB=$R: "; NR_PROJ_RAWB=$R timeout 120 python tools/prof_kernel.py proj_train 2>/dev/null | tail -1; done
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
kb = d["kernel_breakdown_us_per_step"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "loss", round(d["loss"], 4), {k: v for k, v in kb.items() if 'proj' in k})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; m=$2; shift 2; env "$@" timeout 600 python bench.py --model $m $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
run NRMS_rawb0_a NRMS NR_PROJ_RAWB=0; run NRMS_rawb1_a NRMS NR_PROJ_RAWB=1; run NRMS_rawb0_b NRMS NR_PROJ_RAWB=0; run NRMS_rawb1_b NRMS NR_PROJ_RAWB=1
