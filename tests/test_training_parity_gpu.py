"""Statistical training parity against the REAL reference.

tests/golden/train_parity/{nrms,naml,lstur}.npz were written in the build container by oracle/make_golden_train_parity.py: a teacher-labelled
task and the held-out AUC / nDCG@10 of the reference's OWN model classes (imported from its checkout) trained on it with torch's dropout and
torch.optim.Adam -- the loop body of src/train.py:202-233 -- from a seeded initial state, EIGHT torch seeds each.  Here the engine (bf16
operands, its counter-based dropout, EngineAdam; LSTUR: persistent GRU sweeps, row-sparse lazy Adam, user masking 0.5) trains on the same
batches from the same initial state with as many dropout streams of its own, and the two samples are compared:

    |mean_engine - mean_reference|  <  3 * sqrt(s_e^2 / n_e + s_r^2 / n_r)        for AUC and for nDCG@10,

NRMS with THIRTY-TWO seeds a side: the first eight a side gave diff = -3.6e-3 at z = -2.3 (engine below the reference in 51 of 64 pairs).  Two
CPU experiments settled what that was (tools/engine_masks_experiment.py, tools/bf16_bias_experiment.py, profiles/r06_train_parity_analysis.txt):
the fp32 oracle trained with the ENGINE's masks lands on the engine's AUCs seed by seed (mean 0.75207 vs 0.75204) -- so the engine's compute
tracks the reference's math over 200 steps, and the difference is a property of the eight mask STREAMS, not of the arithmetic (the oracle with
every operand rounded to bf16 and torch's own masks lands on the reference: +8e-5) -- and 24 more reference seeds moved the reference's mean from
0.75566 to 0.75432 (sd 3.2e-3): the first eight were a high draw.

a bound DERIVED from the measured spreads (r05's legs used floors of 5e-3 .. 1.5e-2 that could not see a 1e-2 deficit).  The signed difference
and the number of (engine, reference) pairs with the engine below are in the record (gpurun_out/train_parity_fixture_<model>.json, copied to
profiles/r06_train_parity_*.json)."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize('model_name', ['NRMS', 'NAML', 'LSTUR'])
def test_engine_and_reference_train_to_the_same_auc(model_name):
    import bench
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    r = bench.train_parity_fixture(dev, model_name, engine_seeds=32 if model_name == 'NRMS' else 8)
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f'train_parity_fixture_{model_name}.json'), 'w') as f:
            json.dump(r, f, indent=1)
    assert len(r["reference_auc"]) >= 8 and len(r["engine_auc"]) >= 8, r
    gain = r["mean_reference_auc"] - r["auc_init"]
    assert gain > 0.03, r                                                # the reference's loop learns the task ...
    assert min(r["engine_auc"]) > r["auc_init"] + 0.6 * gain, r          # ... and so does the engine, on every dropout stream
    assert abs(r["z_auc"]) < 3.0, (r["diff_auc"], r["stderr_diff_auc"], r)
    assert abs(r["z_ndcg10"]) < 3.0, (r["diff_ndcg10"], r["stderr_diff_ndcg10"], r)
    # the two trainers also end at the same training loss (mean of the last ten steps, averaged over the seeds)
    le = sum(r["engine_last10_loss"]) / len(r["engine_last10_loss"])
    lr_ = sum(r["reference_last10_loss"]) / len(r["reference_last10_loss"])
    assert abs(le - lr_) < 0.05, (le, lr_)
