#!/usr/bin/env python
"""Diagnostics of the statistical training-parity leg (GPU box): where does a difference between the engine's and the reference's trained
models come from?  Variants of bench.train_parity_fixture on the NRMS fixture, 8 dropout streams each:
  base         EngineAdam, p = 0.2; the trained weights also ranked by the CPU oracle (training vs scoring)
  torch_adam   torch.optim.Adam on the engine's gradients (optimiser vs gradients)
  p0           dropout off on both sides: ONE deterministic run each, engine vs the CPU oracle trained here with p = 0 (dropout vs the rest)
Writes gpurun_out/train_parity_diag.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from oracle import train_parity as tp   # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
out = {}
keep = ("engine_auc", "mean_engine_auc", "mean_reference_auc", "diff_auc", "z_auc", "engine_last10_loss", "engine_weights_scored_by_oracle_auc", "optimizer", "dropout")
which = sys.argv[1:] or ['base', 'torch_adam', 'p0']
if 'base' in which:
    r = bench.train_parity_fixture(dev, 'NRMS', 8, oracle_scored=True)
    out['base'] = {k: r.get(k) for k in keep}
    print('base', json.dumps(out['base']), flush=True)
if 'torch_adam' in which:
    r = bench.train_parity_fixture(dev, 'NRMS', 8, optimizer='torch')
    out['torch_adam'] = {k: r.get(k) for k in keep}
    print('torch_adam', json.dumps(out['torch_adam']), flush=True)
if 'p0' in which:
    r = bench.train_parity_fixture(dev, 'NRMS', 1, p_drop=0.0, oracle_scored=True)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_parity', 'nrms.npz'))
    task = tp.task_from_arrays(z)
    st0 = tp.init_state(task["num_words"])
    trained, losses = tp.train_oracle(task, st0, lr=1e-3, p_drop=0.0, torch_seed=0)
    om = tp.eval_metrics(task, tp.oracle_eval_scores(task, trained))
    out['p0'] = {"engine_auc": r["engine_auc"], "engine_last10_loss": r["engine_last10_loss"], "engine_weights_scored_by_oracle_auc": r.get("engine_weights_scored_by_oracle_auc"),
                 "oracle_auc": float(om[0]), "oracle_last10_loss": float(np.mean(losses[-10:]))}
    print('p0', json.dumps(out['p0']), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'train_parity_diag.json'), 'w') as f:
    json.dump(out, f, indent=1)
