#!/bin/bash
# The reference's UNCHANGED src/train.py and src/evaluate.py through the launcher on one MI355X.  The reference checkout is
# not part of this repository: tools/run_launcher_on_gpu.sh places a temporary, git-ignored copy of its src/ under .ref_scratch/ for the
# duration of the gpurun call and removes it afterwards.
export TMPDIR=/tmp
export PYTHONWARNINGS=ignore
O=$PWD/gpurun_out/${1:-r02_launcher}
REF=$PWD/.ref_scratch/src
mkdir -p $O
for M in NRMS LSTUR NAML; do
  RUN=/tmp/run_$M
  rm -rf $RUN; mkdir -p $RUN
  python - <<PY
import sys
sys.path.insert(0, "$PWD")
from news_recommendation_amd import synth
synth.write_reference_dataset("$RUN", n_news=4000, n_users=3000, n_train=19200 if "$M" == "NRMS" else 6400, n_val_impr=1500, num_words=70976, seed=1)
PY
  STEPS=$([ $M = NRMS ] && echo 100 || echo 50)
  echo "=== $M: unchanged train.py (reference DataLoader, torch.optim.Adam, reference evaluate() for validation every $STEPS batches) ===" | tee -a $O/launcher_$M.log
  ( time timeout 900 python -m news_recommendation_amd.launcher train --reference $REF --workdir $RUN --model $M --set num_batches_validate=$STEPS num_batches_show_loss=50 ) 2>&1 \
     | tr '\r' '\n' | grep -v "it/s\]\|^$\|amdgpu.ids" | tail -40 >> $O/launcher_$M.log
  echo "=== $M: unchanged evaluate.py on ./data/test ===" | tee -a $O/launcher_$M.log
  ( time timeout 600 python -m news_recommendation_amd.launcher evaluate --reference $REF --workdir $RUN --model $M ) 2>&1 | tr '\r' '\n' | grep -v "it/s\]\|^$\|amdgpu.ids" | tail -12 >> $O/launcher_$M.log
  echo "=== $M: evaluate with the batched driver (--fast-eval), same checkpoint ===" | tee -a $O/launcher_$M.log
  ( time timeout 600 python -m news_recommendation_amd.launcher evaluate --reference $REF --workdir $RUN --model $M --fast-eval ) 2>&1 | grep -v "amdgpu.ids" | tail -8 >> $O/launcher_$M.log
  ls $RUN/checkpoint/$M | tail -3 >> $O/launcher_$M.log
  tail -30 $O/launcher_$M.log
done
