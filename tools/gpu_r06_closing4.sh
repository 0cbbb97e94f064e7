#!/bin/bash
# Round-6 closing pass, fourth edition (final library of the third session: conv data gradient in pairs order, the stream form of nr_dx_gemm and the GEMM form of the conv forward behind their switches): profile pass ->
# JSON products into profiles/ -> bench lines of the five workloads -> the whole GPU suite, all on ONE box
export TMPDIR=/tmp
( time bash tools/gpu_r06_profiles.sh r06prof4 ) 2>&1 | tail -40 > gpurun_out/r06_closing4_tail.txt
for f in traffic.json mfma_busy.json step_traffic.json; do [ -f gpurun_out/r06prof4/$f ] && cp gpurun_out/r06prof4/$f profiles/$f; done
( time bash tools/gpu_r06_final.sh r06final4 ) 2>&1 | tail -80 >> gpurun_out/r06_closing4_tail.txt
tail -90 gpurun_out/r06_closing4_tail.txt
( time timeout 3600 python -m pytest tests -m gpu -q --timeout 2400 ) > gpurun_out/r06_pytest_gpu_full4.txt 2>&1
tail -8 gpurun_out/r06_pytest_gpu_full4.txt
