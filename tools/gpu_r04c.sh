#!/bin/bash
# round 4, call b: the general ring GEMMs and the LSTUR step graph: parity tests, then A/B bench lines (hand-written vs library GEMMs; graph vs eager)
export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_graph_gpu.py tests/test_rccl_gpu.py tests/test_lstur_gpu.py tests/test_naml_gpu.py tests/test_conv_grad_unquantised_gpu.py -m gpu -q > $O/pytest_b.log 2>&1; tail -12 $O/pytest_b.log
timeout 600 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "attn_bwd or golden_base or mind_shape" > $O/pytest_b2.log 2>&1; tail -5 $O/pytest_b2.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "host", round(d["host_enqueue_ms_per_step"], 2), "loss", round(d["loss"], 4))
print("   kernels", {k: v for k, v in list(d["kernel_breakdown_us_per_step"].items())[:16]})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $ARGS $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
ARGS="--model LSTUR";              run lstur_hand NR_GEMM_HAND=15;  run lstur_lib NR_GEMM_HAND=0
ARGS="--model LSTUR --no-graph";   run lstur_hand_eager NR_GEMM_HAND=15
ARGS="--model NAML";               run naml_hand NR_GEMM_HAND=15;   run naml_lib NR_GEMM_HAND=0
ARGS="";                           run nrms_hand NR_GEMM_HAND=15;   run nrms_lib NR_GEMM_HAND=11
