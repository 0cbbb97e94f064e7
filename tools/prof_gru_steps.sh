#!/bin/bash
# Per-launch averages of the GRU step kernels (rocprofv3 kernel trace over tools/prof_gru.py).  Usage: bash tools/prof_gru_steps.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats -d /tmp/pg -o pg --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_gru.py 2>&1 | grep fwd | tail -1
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
grep gru_ $f | cut -c1-140
