#!/bin/bash
# Round-6 closing pass, second edition (after the conv data-gradient GEMM and the pooling changes): profile pass -> JSON products into profiles/ ->
# bench lines -> the whole GPU suite
export TMPDIR=/tmp
bash tools/gpu_r06_closing.sh 2>&1 | tail -120 > gpurun_out/r06_closing2_tail.txt
tail -70 gpurun_out/r06_closing2_tail.txt
( time timeout 3600 python -m pytest tests -m gpu -q --timeout 2400 ) > gpurun_out/r06_pytest_gpu_full.txt 2>&1
tail -8 gpurun_out/r06_pytest_gpu_full.txt
