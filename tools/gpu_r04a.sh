#!/bin/bash
# round 4, call a: new parity tests at bench scale, default bench line with the multi-seed / training parity legs, gather probe under rocprofv3
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_training_parity_gpu.py tests/test_proj_gpu.py -m gpu -q -x -k "bench_scale or train_to_the_same or oracle_at_bench" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "bench_scale or step_counter" > $O/pytest_new2.log 2>&1; tail -5 $O/pytest_new2.log
( time timeout 900 python bench.py > $O/bench_line_NRMS_small.json 2> $O/bench_line_NRMS_small.err ) 2>&1 | tail -3
tail -c 600 $O/bench_line_NRMS_small.err
bash tools/gather_prof.sh $O > $O/gather_prof.log 2>&1; cat $O/gather_rocprof.txt
