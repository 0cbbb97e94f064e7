#!/bin/bash
# Round-6 closing pass on ONE box: the profile pass (tools/gpu_r06_profiles.sh), its JSON products copied into profiles/ of the box's checkout so
# that the bench lines taken right after quote traffic / MFMA-busy figures of THIS library, then the bench lines (tools/gpu_r06_final.sh).
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_naml_gpu.py tests/test_config_knobs_gpu.py -x -q 2>&1 | tail -3
( time bash tools/gpu_r06_profiles.sh r06prof2 ) 2>&1 | tail -40
for f in traffic.json mfma_busy.json step_traffic.json; do [ -f gpurun_out/r06prof2/$f ] && cp gpurun_out/r06prof2/$f profiles/$f; done
( time bash tools/gpu_r06_final.sh r06final2 ) 2>&1 | tail -60
