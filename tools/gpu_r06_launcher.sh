#!/bin/bash
# the reference's UNCHANGED src/train.py and src/evaluate.py on MI355X with the CURRENT library, all three models, on a
# LEARNABLE synthetic tree (synth.write_reference_dataset(learnable=True)): >= 300 training steps with validation and a checkpoint, then
# evaluate.py; then, on the SAME checkpoint and data, the reference's own evaluate() with the reference's own model on the host cores against
# the same function with the engine's model (tools/eval_both_ways.py) -> |dAUC|, |dnDCG@10| (north_star: within 1e-3).
# The reference checkout is not part of this repository: tools/run_r06_launcher_on_gpu.sh places a temporary, git-ignored copy of its src/
# under .ref_scratch/ for the duration of the gpurun call.
export TMPDIR=/tmp
export PYTHONWARNINGS=ignore
O=$PWD/gpurun_out/${1:-r06_launcher}
REF=$PWD/.ref_scratch/src
mkdir -p $O
for M in ${MODELS:-NRMS NAML LSTUR}; do
  RUN=/tmp/run_$M
  rm -rf $RUN; mkdir -p $RUN
  python - <<PY
import sys
sys.path.insert(0, "$PWD")
from news_recommendation_amd import synth
synth.write_reference_dataset("$RUN", n_news=4000, n_users=3000, n_train=25600, n_val_impr=10000, num_words=20000, seed=1, learnable=True)
PY
  L=$O/launcher_unchanged_train_evaluate_$M.log
  SET="num_words=20000 num_users=3001 learning_rate=0.0005 num_batches_validate=200 num_batches_show_loss=100"
  echo "=== $M: unchanged train.py (reference DataLoader, torch.optim.Adam, reference evaluate() for validation every 200 batches); config knobs: $SET ===" | tee -a $L
  ( time timeout 1500 python -m news_recommendation_amd.launcher train --reference $REF --workdir $RUN --model $M --set $SET ) 2>&1 \
     | tr '\r' '\n' | grep -v "it/s\]\|s/it\]\|^$\|amdgpu.ids" | tail -30 >> $L
  echo "=== $M: unchanged evaluate.py on ./data/test ===" | tee -a $L
  ( time timeout 600 python -m news_recommendation_amd.launcher evaluate --reference $REF --workdir $RUN --model $M --set num_words=20000 num_users=3001 ) 2>&1 | tr '\r' '\n' | grep -v "it/s\]\|s/it\]\|^$\|amdgpu.ids" | tail -12 >> $L
  ls $RUN/checkpoint/$M | tail -3 >> $L
  echo "=== $M: the same checkpoint, the reference's evaluate() both ways ===" | tee -a $L
  for W in engine reference; do
    timeout 1500 python tools/eval_both_ways.py --which $W --reference $REF --workdir $RUN --model $M --set num_words=20000 num_users=3001 2>$O/eval_${M}_$W.err | grep '^{' | tee -a $L > $O/eval_${M}_$W.json
  done
  python - <<PY | tee -a $L
import json
e = json.load(open("$O/eval_${M}_engine.json")); r = json.load(open("$O/eval_${M}_reference.json"))
d = {"model": "$M", "checkpoint": e["checkpoint"], "engine": {k: e[k] for k in ("auc", "mrr", "ndcg5", "ndcg10", "device", "model_package", "seconds")},
     "reference_cpu": {k: r[k] for k in ("auc", "mrr", "ndcg5", "ndcg10", "device", "model_package", "seconds", "threads")},
     "abs_diff": {k: abs(e[k] - r[k]) for k in ("auc", "mrr", "ndcg5", "ndcg10")}}
d["within_1e-3"] = bool(d["abs_diff"]["auc"] <= 1e-3 and d["abs_diff"]["ndcg10"] <= 1e-3)
json.dump(d, open("$O/both_ways_$M.json", "w"), indent=1)
print(json.dumps(d))
PY
  tail -25 $L
done
