"""GPU-resident training data + fast trainer.  CPU (build container): default_config mirrors the reference's
config.py and TrainData reproduces BaseDataset's samples.  GPU: a short run trains, validates with the batched evaluator,
writes a reference-format checkpoint and resumes from it."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/src'

DATA_VS_REFERENCE = r'''
import os, sys
import numpy as np, torch
sys.dont_write_bytecode = True
root, ref, workdir, model_name = sys.argv[1:5]
os.environ['MODEL_NAME'] = model_name
sys.path.insert(0, ref); sys.path.insert(0, root)
os.chdir(workdir)
import config as ref_config
import dataset as ref_dataset                                   # the reference's dataset.py (read-only)
from news_recommendation_amd import default_config
from news_recommendation_amd.data_fast import TrainData, split_batch
rc = getattr(ref_config, model_name + 'Config'); dc = getattr(default_config, model_name + 'Config')
for k in dir(dc):
    if not k.startswith('_'):
        assert getattr(rc, k) == getattr(dc, k), k
ds = ref_dataset.BaseDataset('data/train/behaviors_parsed.tsv', 'data/train/news_parsed.tsv')
td = TrainData('data/train/behaviors_parsed.tsv', 'data/train/news_parsed.tsv', rc, 'cpu')
assert len(ds) == len(td)
idx = torch.tensor([0, 3, len(ds) - 1, 7])
b = td.batch(idx)
b['cand'], b['click'] = split_batch(b)          # [B, C, ...] / [B, N, ...] views of the stacked layout
for j, i in enumerate(idx.tolist()):
    it = ds[i]
    for c in range(len(it['candidate_news'])):
        for a in rc.dataset_attributes['news']:
            assert torch.equal(torch.as_tensor(it['candidate_news'][c][a]), b['cand'][a][j, c]), (i, c, a)
    for n in range(rc.num_clicked_news_a_user):
        for a in rc.dataset_attributes['news']:
            assert torch.equal(torch.as_tensor(it['clicked_news'][n][a]), b['click'][a][j, n]), (i, n, a)
    if 'user' in rc.dataset_attributes['record']:
        assert it['user'] == int(b['user'][j]) and it['clicked_news_length'] == int(b['length'][j])
shard = TrainData('data/train/behaviors_parsed.tsv', 'data/train/news_parsed.tsv', rc, 'cpu', rank=1, world=2)
assert len(shard) == len(ds) // 2 and torch.equal(shard.cand[0], td.cand[1])
print('data ok', model_name, len(ds))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize('model_name', ['NRMS', 'NAML', 'LSTUR'])
def test_train_data_matches_reference_dataset(tmp_path, model_name):
    from news_recommendation_amd import synth
    synth.write_reference_dataset(str(tmp_path))
    p = subprocess.run([sys.executable, '-c', DATA_VS_REFERENCE, ROOT, REF, str(tmp_path), model_name],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert p.returncode == 0 and 'data ok' in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize('model_name', ['NRMS', 'NAML', 'LSTUR'])
def test_fast_trainer_trains_validates_checkpoints_and_resumes(tmp_path, model_name):
    from news_recommendation_amd import synth, train_fast
    synth.write_reference_dataset(str(tmp_path), n_news=300, n_train=512, n_val_impr=40, num_words=500)
    over = ['batch_size=64', 'num_words=500', 'num_users=41', 'num_categories=30', 'learning_rate=0.002', 'num_epochs=6',
            'num_batches_show_loss=4', 'num_batches_validate=16']
    cfg = train_fast.load_config(model_name, None, over)
    lines = []
    cwd = os.getcwd()
    try:
        torch.manual_seed(0)
        r = train_fast.train(model_name, cfg, str(tmp_path), log=lines.append)
        assert r['steps'] == 48
        losses = [float(l.split('current loss ')[1].split(',')[0]) for l in lines if 'current loss' in l]
        assert losses[-1] < losses[0] - 0.05, losses                      # it learns the (fixed) synthetic clicks
        assert sum('validation AUC' in l for l in lines) == 3
        ck = sorted(os.listdir(os.path.join(tmp_path, 'checkpoint', model_name)))
        assert ck and all(c.startswith('ckpt-') and c.endswith('.pth') for c in ck)
        sd = torch.load(os.path.join(tmp_path, 'checkpoint', model_name, ck[-1]), map_location='cpu')
        assert set(sd) == {'model_state_dict', 'optimizer_state_dict', 'step', 'early_stop_value'}      # train.py:264-277
        assert set(sd['model_state_dict']) == set(r['model'].state_dict())
        lines.clear()
        os.chdir(cwd)
        r2 = train_fast.train(model_name, cfg, str(tmp_path), max_steps=2, log=lines.append)
        assert any('Load saved parameters' in l for l in lines) and r2['steps'] == 2
    finally:
        os.chdir(cwd)


def test_load_checkpoint_accepts_reference_format_with_weights_only(tmp_path):
    """train.py:264-277 saves 'early_stop_value' as a numpy scalar; torch >= 2.6's default weights_only=True rejects it.  The trainer's
    loader (and the launcher's process-wide allow-list) admit exactly the numpy scalar / dtype types, nothing else."""
    import numpy as np
    from news_recommendation_amd import train_fast
    path = str(tmp_path / 'ckpt-7.pth')
    torch.save({'model_state_dict': {'w': torch.arange(4.0)}, 'optimizer_state_dict': {'state': {}, 'param_groups': []}, 'step': 7,
                'early_stop_value': -np.float64(0.61)}, path)
    with pytest.raises(Exception):
        torch.load(path, map_location='cpu', weights_only=True)
    ck = train_fast.load_checkpoint(path, 'cpu')
    assert ck['step'] == 7 and float(ck['early_stop_value']) == -0.61 and torch.equal(ck['model_state_dict']['w'], torch.arange(4.0))

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    torch.save({'x': Evil()}, path)
    with pytest.raises(Exception):
        train_fast.load_checkpoint(path, 'cpu')


@pytest.mark.gpu
def test_failed_persistent_sweep_stays_on_record_and_gates_the_optimiser():
    """the error word of a persistent GRU sweep was erased by the next launch's memset.  A forced give-up (nr_debug_gru_fault:
    workgroup 0 never arrives, its team mates' bounded waits run out) must stay visible through a CLEAN sweep that follows, and while it does no
    optimiser kernel applies an update."""
    from news_recommendation_amd import _capi, ops_gru
    lib = _capi.load()
    dev = torch.device('cuda', 0)
    B, N, Hd = 64, 6, 900
    if not (lib.nr_gru_persist_enabled(B, Hd, N) & 1):
        pytest.skip("the persistent GRU sweeps are not enabled on this device / under this NR_GRU_PERSIST")
    ops_gru.fault_clear()
    gru = torch.nn.GRU(Hd, Hd, batch_first=True).to(dev)
    x = torch.randn(B, N, Hd, device=dev) * 0.3
    length = torch.full((B,), N, dtype=torch.int64)
    with torch.no_grad():
        good = ops_gru.gru_last_state(x, None, length, gru).clone()
        assert ops_gru.fault_state()[:2] == (0, 0)
        assert lib.nr_debug_gru_fault(0, 1) == 0                     # the NEXT forward sweep fails
        ops_gru.gru_last_state(x, None, length, gru)
        st = ops_gru.fault_state()
        assert st[0] & 2 and st[1] == 0, st                          # "a wait gave up" on the forward sweep
        again = ops_gru.gru_last_state(x, None, length, gru).clone() # a clean sweep: correct output ...
        assert torch.equal(again, good)
        assert ops_gru.fault_state()[:2] == st[:2]                   # ... and the evidence is still there
        with pytest.raises(RuntimeError, match='persistent GRU sweep failed'):
            ops_gru.persist_check()
        # the optimiser is gated: nr_adam_flat drops the gradient and moves nothing; the skipped step index is recorded
        p, g = torch.randn(1024, device=dev), torch.randn(1024, device=dev)
        m, v = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
        from news_recommendation_amd.optim import AdamSchedule
        sched = AdamSchedule(1e-3, (0.9, 0.999), dev)
        p0 = p.clone()
        args = lambda step: (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 1024, sched.table.data_ptr(), step, 0.9, 0.999, 1e-8, 1.0, 1,
                             torch.cuda.current_stream().cuda_stream)
        assert lib.nr_adam_flat(*args(7)) == 0
        assert torch.equal(p, p0) and not g.any() and not m.any() and ops_gru.fault_state()[2] == 7
        ops_gru.fault_clear()
        g.normal_()
        assert lib.nr_adam_flat(*args(7)) == 0
        assert not torch.equal(p, p0) and ops_gru.fault_state() == (0, 0, 0, 0)


@pytest.mark.gpu
def test_fast_trainer_repeats_the_steps_of_a_failed_persistent_sweep(tmp_path, monkeypatch):
    """a sweep that gives up must never be trained on, and the step is repeated.  The 6th training forward sweep of an LSTUR
    run is made to fail; the trainer looks every 4 steps: at step 8 it finds optimiser steps 6-8 skipped, switches to the step-per-launch
    kernels and repeats them; the run ends with all 12 optimiser steps taken and a falling loss."""
    from news_recommendation_amd import _capi, ops_gru, synth, train_fast
    lib = _capi.load()
    if not (lib.nr_gru_persist_enabled(64, 900, 50) & 1):
        pytest.skip("the persistent GRU sweeps are not enabled on this device / under this NR_GRU_PERSIST")
    monkeypatch.setenv('NR_GRU_PERSIST', '3')
    synth.write_reference_dataset(str(tmp_path), n_news=300, n_train=512, n_val_impr=40, num_words=500)
    over = ['batch_size=64', 'num_words=500', 'num_users=41', 'num_categories=30', 'learning_rate=0.002', 'num_epochs=2',
            'num_batches_show_loss=4', 'num_batches_validate=1000']
    cfg = train_fast.load_config('LSTUR', None, over)
    lines = []
    cwd = os.getcwd()
    ops_gru.fault_clear()
    try:
        torch.manual_seed(0)
        assert lib.nr_debug_gru_fault(0, 6) == 0
        r = train_fast.train('LSTUR', cfg, str(tmp_path), max_steps=12, log=lines.append)
    finally:
        os.chdir(cwd)
        lib.nr_debug_gru_fault(0, 0)
    msg = [l for l in lines if 'persistent GRU sweep failed' in l]
    assert len(msg) == 1 and 'optimiser step 6' in msg[0] and 'repeating 3 step(s)' in msg[0], lines
    assert r['steps'] == 12 and r['optimizer_steps'] == 12
    assert os.environ['NR_GRU_PERSIST'] == '0'
    assert ops_gru.fault_state() == (0, 0, 0, 0)
    losses = [float(l.split('current loss ')[1].split(',')[0]) for l in lines if 'current loss' in l]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    avg = [float(l.split('average loss: ')[1].split(',')[0]) for l in lines if 'current loss' in l]
    assert all(np.isfinite(avg)), avg                                   # the failed steps' losses are not in the running mean
    assert all(torch.isfinite(p).all() for p in r['model'].parameters())
