#!/bin/bash
# Ship a TEMPORARY copy of the reference's src/ (git-ignored .ref_scratch/, removed again on exit) with the gpurun snapshot and run the
# reference's unchanged train.py / evaluate.py through the launcher on the GPU box.  Nothing of the reference is committed.
set -u
cd "$(dirname "$0")/.."
trap 'rm -rf .ref_scratch' EXIT
rm -rf .ref_scratch && mkdir -p .ref_scratch && cp -r /root/reference/src .ref_scratch/src
/usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_launcher_run.sh r02_launcher'
