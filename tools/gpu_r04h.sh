#!/bin/bash
# round 4, call h: A/B of the raw barrier behind attn_bwd's write-out (NR_ATTN_RAWB)
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
NR_ATTN_RAWB=1 timeout 900 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "attn_bwd or golden_base or mind_shape or bench_scale or dropout_matches" > $O/pytest_h.log 2>&1; tail -4 $O/pytest_h.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "loss", round(d["loss"], 4), {k: v for k, v in list(d["kernel_breakdown_us_per_step"].items())[:4]})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
run nrms_rawb0_a NR_ATTN_RAWB=0; run nrms_rawb1_a NR_ATTN_RAWB=1; run nrms_rawb0_b NR_ATTN_RAWB=0; run nrms_rawb1_b NR_ATTN_RAWB=1
