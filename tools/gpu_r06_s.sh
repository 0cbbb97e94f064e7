#!/bin/bash
# nineteenth GPU pass of round 6: which Python lines still issue torch device operations inside a training step (three models)
export TMPDIR=/tmp
O=gpurun_out/r06s
mkdir -p $O
for M in NRMS NAML LSTUR; do
  timeout 300 python tools/diag_glue2.py $M small > $O/glue2_$M.txt 2>&1
  echo "== $M"; grep -v Warning $O/glue2_$M.txt | tail -60
done
