#!/bin/bash
# round 4, call d: conv data gradient in GEMM form, pooling gradients on the ring kernel: parity tests + A/B bench lines
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_rccl_gpu.py tests/test_naml_gpu.py tests/test_lstur_gpu.py tests/test_conv_grad_unquantised_gpu.py tests/test_model_gpu.py -m gpu -q -k "conv_dgrad or rccl or naml or lstur or conv_weight or standalone or golden_base" > $O/pytest_d.log 2>&1; tail -8 $O/pytest_d.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "eager", d.get("ms_per_step_eager"), "host", round(d["host_enqueue_ms_per_step"], 2), "loss", round(d["loss"], 4))
print("   kernels", {k: v for k, v in list(d["kernel_breakdown_us_per_step"].items())[:18]})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $ARGS $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
ARGS="--model NAML";   run naml_63 NR_GEMM_HAND=63;   run naml_31 NR_GEMM_HAND=31; run naml_15 NR_GEMM_HAND=15
ARGS="--model LSTUR";  run lstur_63 NR_GEMM_HAND=63;  run lstur_15 NR_GEMM_HAND=15
ARGS="";               run nrms_63 NR_GEMM_HAND=63;   run nrms_15 NR_GEMM_HAND=15
