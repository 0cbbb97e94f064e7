#!/usr/bin/env python
"""The north_star's target sentence, end to end: ONE checkpoint (written by the reference's unchanged train.py running on the engine), ONE data
tree, the reference's unchanged ``evaluate()`` (src/evaluate.py:171-272) called twice --

    --which engine      model = the engine's drop-in ``model.<NAME>`` on cuda:0 (sys.path as news_recommendation_amd/launcher.py sets it)
    --which reference   model = the reference's OWN ``model.<NAME>`` on the host cores (GPUs hidden before torch is imported)

-- and the four metrics printed at full precision as one JSON line (evaluate.py's own __main__ prints four decimals).  Test / measurement
infrastructure: the reference checkout exists on the GPU box only as the temporary copy tools/gpu_r06_launcher.sh ships."""
import argparse
import importlib
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument('--which', choices=['engine', 'reference'], required=True)
ap.add_argument('--reference', required=True, help="the reference's src/ directory")
ap.add_argument('--workdir', required=True)
ap.add_argument('--model', default='NRMS')
ap.add_argument('--split', default='test')
ap.add_argument('--set', nargs='*', default=[], metavar='KNOB=VALUE', help="override attributes of the reference's config class in memory (as the launcher does)")
a = ap.parse_args()
if a.which == 'reference':
    os.environ['HIP_VISIBLE_DEVICES'] = ''
    os.environ['CUDA_VISIBLE_DEVICES'] = ''
os.environ['MODEL_NAME'] = a.model
sys.dont_write_bytecode = True
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = os.path.abspath(a.reference)
paths = [ref, repo] + ([os.path.join(repo, 'news_recommendation_amd', 'dropin')] if a.which == 'engine' else [])
for p in paths:
    sys.path.insert(0, p)
from news_recommendation_amd.launcher import install_shims, apply_overrides   # noqa: E402  (np.Inf, tensorboard stub, torch.load allow-list: SURVEY 8 c2)
install_shims()
os.chdir(a.workdir)
apply_overrides(a.model, a.set)
import torch   # noqa: E402
import evaluate as ref_eval   # noqa: E402  the reference's module (its `Model` is whichever model package sys.path resolves first)
from train import latest_checkpoint   # noqa: E402

model_mod = importlib.import_module(f'model.{a.model}')
origin = os.path.relpath(model_mod.__file__, repo) if model_mod.__file__.startswith(repo) else model_mod.__file__
model = ref_eval.Model(ref_eval.config).to(ref_eval.device)
path = latest_checkpoint(os.path.join('./checkpoint', a.model))
model.load_state_dict(torch.load(path, map_location=ref_eval.device)['model_state_dict'])      # (the checkpoint was written from cuda:0; evaluate.py's own bare torch.load needs a GPU)
model.eval()
t0 = time.perf_counter()
auc, mrr, n5, n10 = ref_eval.evaluate(model, f'./data/{a.split}', ref_eval.config.num_workers)
print(json.dumps({"which": a.which, "model": a.model, "model_package": origin, "device": str(ref_eval.device), "checkpoint": path, "split": a.split,
                  "auc": float(auc), "mrr": float(mrr), "ndcg5": float(n5), "ndcg10": float(n10), "seconds": time.perf_counter() - t0,
                  "threads": torch.get_num_threads()}))
