#!/bin/bash
# Ship a TEMPORARY copy of the reference's src/ (git-ignored .ref_scratch/, removed again on exit) with the gpurun snapshot and run
# tools/gpu_r06_launcher.sh on the GPU box.  Nothing of the reference is committed.  Extra commands to run in the same call: "$@".
set -u
cd "$(dirname "$0")/.."
trap 'rm -rf .ref_scratch' EXIT
rm -rf .ref_scratch && mkdir -p .ref_scratch && cp -r /root/reference/src .ref_scratch/src
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-2400} -- "bash tools/gpu_r06_launcher.sh r06_launcher; ${*:-true}"
