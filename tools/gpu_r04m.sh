#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
timeout 300 python tools/pool3_phases.py --timeline > $O/pool3_phases.txt 2> $O/pool3_phases.err; cat $O/pool3_phases.txt; tail -3 $O/pool3_phases.err
