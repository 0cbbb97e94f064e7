#!/bin/bash
# last GPU pass of round 6 on the final library: smoke(), the default bench line, the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r06last
mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
( time timeout 1200 python bench.py ) > $O/bench_default.txt 2> $O/bench_default.err; tail -3 $O/bench_default.txt | cut -c1-600
grep '^{' $O/bench_default.txt | tail -1 > $O/bench_line_default.json
( time timeout 3600 python -m pytest tests -m gpu -q --timeout 2400 ) > $O/pytest_gpu_full.txt 2>&1
tail -8 $O/pytest_gpu_full.txt
