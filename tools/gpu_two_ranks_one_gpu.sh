#!/bin/bash
# Two ranks on the single GPU of a gpurun box, gloo as the transport (RCCL refuses two ranks on one device): exercises the N > 1 code path
# of bench.py (ranks leaving the group, large shapes, touched-row exchange, bucket overlap) for hangs and crashes.  The numbers are NOT
# scaling measurements -- both ranks share one GPU and the exchange goes through host memory.
set -u
mkdir -p gpurun_out
export NR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
# two processes on ONE GPU: the persistent GRU sweeps need all 256 CUs of the device to themselves (two such kernels interleaved on the CUs would
# each wait for workgroups the other keeps out until the bounded waits give up and raise the error word): the step launches here
export NR_GRU_PERSIST=0
for m in NRMS NAML LSTUR; do
  timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
      bench.py --gpus 2 --steps 10 --warmup 3 --model $m > gpurun_out/two_ranks_$m.log 2>&1
  echo "rc[$m]=$?" | tee -a gpurun_out/two_ranks_$m.log
  tail -c 1500 gpurun_out/two_ranks_$m.log
done
