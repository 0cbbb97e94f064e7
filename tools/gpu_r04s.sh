#!/bin/bash
# round 4, call s: the whole GPU suite + smoke() on the library with the flat pooling backward as the default
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
