"""The RCCL code path of EngineAdam on the one GPU a test box has: a process group of ONE rank over backend 'nccl' (= RCCL on ROCm) with
the optimiser forced onto its multi-rank branch (force_dist) -- the table bucket's collective started from inside the backward on RCCL's
stream, the small bucket, the touched-row all-gather, and the reduce-scatter / sharded-Adam / all-gather form.  Collectives over one rank
are identities, so the parameters after a few training steps must equal the single-process fast path up to the run-to-run noise of the
embedding scatter (rows that span several chunks of the sorted token list are combined with fp32 atomics: the summation order, hence the
last bit, varies between two runs of the SAME path -- measured here with a second single-process run).  What the test exercises is the
stream ordering between the engine's kernels (current stream) and RCCL's internal stream, and the allocator hand-off of the buffers the
collectives touch (net-new vs src/train.py:227-233, which has no distributed code): a collective that ran before its input was complete,
or an update that ran before the collective, shows up as a difference of the order of the update itself (1e-3), not of the last bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
mode, model_name, out = sys.argv[2], sys.argv[3], sys.argv[4]
from bench import Workload, make_cfg
from news_recommendation_amd import optim
optim.TABLE_MIN_NUMEL = 1 << 18                   # the reduced word table below is still "the table bucket"
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
if mode != 'local':
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{sys.argv[5]}', rank=0, world_size=1)
cfg = make_cfg(model_name, 'small', vocab=5000)
cfg.num_news, cfg.num_users = 3000, 700
wl = Workload(model_name, cfg)
model = wl.make_model(seed=11).to(dev).train()
opt = optim.EngineAdam(model, lr=1e-3, row_sparse=('user_embedding.weight',) if model_name == 'LSTUR' else (),
                       force_dist=(mode != 'local'), table_rs=(mode == 'rs'))
assert opt._dist_on() == (mode != 'local')
assert any(r.name != 'small' for r in opt.regions)
batches = wl.batches(0, 3, 64, dev)
target = torch.zeros(64, dtype=torch.long, device=dev)
crit = torch.nn.CrossEntropyLoss()
torch.manual_seed(123)                            # dropout seeds are drawn from torch's CPU generator: same masks in every mode
losses = []
for i in range(4):
    loss = crit(wl.forward(model, batches[i % 3]), target)
    loss.backward()
    opt.step()
    losses.append(float(loss.item()))
    assert not opt.flat_g.any()
if mode != 'local':
    assert opt.comm_bytes and all(v > 0 for v in opt.comm_bytes.values()), opt.comm_bytes
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
opt.gather_state()                                # collective when the moments are sharded (mode 'rs'); a no-op otherwise
osd = opt.state_dict()
for i, st in osd['state'].items():
    sd[f'opt/{i}/exp_avg'] = st['exp_avg'].cpu().numpy()
    sd[f'opt/{i}/exp_avg_sq'] = st['exp_avg_sq'].cpu().numpy()
sd['losses'] = np.array(losses)
np.savez(out, **sd)
if mode != 'local':
    dist.destroy_process_group()
print('ok', mode, model_name, losses)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(mode, model_name, tmp_path):
    out = str(tmp_path / f'{mode}_{model_name}.npz')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, mode, model_name, out, str(_free_port())], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    return dict(np.load(out))


@pytest.mark.parametrize('model_name', ['NRMS', 'LSTUR'])
def test_engine_adam_over_rccl_world1_equals_local_path(tmp_path, model_name):
    local = _run('local', model_name, tmp_path)
    again = _run('local', model_name, tmp_path)
    assert np.isfinite(local['losses']).all()
    noise = max(float(np.abs(again[k].astype(np.float64) - local[k]).max()) for k in local)
    tol = max(4 * noise, 2e-6)                     # parameters move by ~1e-3 per step (lr): a mis-ordered exchange is 3 orders above this
    for mode in ('ar', 'rs'):                      # all-reduce form, reduce-scatter + sharded Adam + all-gather form
        got = _run(mode, model_name, tmp_path)
        assert set(got) == set(local)
        for k in local:
            err = float(np.abs(got[k].astype(np.float64) - local[k]).max())
            assert err <= tol, f'{model_name} / {mode}: {k} differs from the single-process path by {err:.3g} (run-to-run noise {noise:.3g})'


CHILD_SEG = r'''
import os, sys, numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
mode, model_name, out = sys.argv[2], sys.argv[3], sys.argv[4]
overlap, table_rs = sys.argv[6] == '1', sys.argv[7] == '1'
from bench import Workload, make_cfg
from news_recommendation_amd import ops, optim
from news_recommendation_amd.graph import SegmentedStep
ops.new_seed = lambda: 0x1234ABCD5678            # same base seeds in both modes: the masks differ from step to step through the counter only
optim.TABLE_MIN_NUMEL = 1 << 18
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{sys.argv[5]}', rank=0, world_size=1)
cfg = make_cfg(model_name, 'small', vocab=5000)
cfg.num_news, cfg.num_users = 3000, 700
wl = Workload(model_name, cfg)
model = wl.make_model(seed=11).to(dev).train()
opt = optim.EngineAdam(model, lr=1e-3, row_sparse=('user_embedding.weight',) if model_name == 'LSTUR' else (), force_dist=True, table_rs=table_rs)
B = 64
batches = wl.batches(0, 3, B, dev)
target = torch.zeros(B, dtype=torch.long, device=dev)
crit = torch.nn.CrossEntropyLoss()
flat = lambda b: [b[s][a] for s in ('cand', 'click') for a in wl.attrs] + ([b['user'], b['length'].to(dev)] if model_name == 'LSTUR' else [])
def fwd_bwd(*xs):
    n = len(wl.attrs)
    cand, click = dict(zip(wl.attrs, xs[:n])), dict(zip(wl.attrs, xs[n:2 * n]))
    if model_name == 'LSTUR':
        logits = model.forward_ids(xs[2 * n], xs[2 * n + 1].clone(), cand, click)
    else:
        logits = model.forward_ids(cand['title'], click['title'])
    loss = crit(logits, target)
    loss.backward()
    return loss
g = SegmentedStep(fwd_bwd, flat(batches[0]), opt, warmup=2, overlap=overlap)
assert opt.t == 2 and not opt.overlap and ops.defer_wgrad == overlap and (g.graph_w is not None) == overlap
losses = []
for i in range(5):
    xs = flat(batches[i % 3])
    loss = g(*xs) if mode == 'graph' else g.eager_step(*xs)
    losses.append(float(loss.item()))
    assert not opt.flat_g.any() and all(not st.pending for st in opt.sparse)
assert opt.t == 7 and int(g.ctr.item()) == 7, (opt.t, int(g.ctr.item()))
assert opt.comm_bytes and all(v > 0 for v in opt.comm_bytes.values()), opt.comm_bytes
g.close()
opt.gather_state()
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
osd = opt.state_dict()
for i, st in osd['state'].items():
    sd[f'opt/{i}/exp_avg_sq'] = st['exp_avg_sq'].cpu().numpy()
    assert float(st['step']) == 7.0
assert not ops._deferred
sd['losses'] = np.array(losses)
np.savez(out, **sd)
dist.destroy_process_group()
print('ok', mode, model_name, losses)
'''


def _run_seg(mode, model_name, tmp_path, overlap=True, table_rs=False):
    out = str(tmp_path / f'seg_{mode}_{model_name}_{int(overlap)}{int(table_rs)}.npz')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-c', CHILD_SEG, ROOT, mode, model_name, out, str(_free_port()), str(int(overlap)), str(int(table_rs))], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    return dict(np.load(out))


@pytest.mark.parametrize('model_name,overlap,table_rs', [('NRMS', True, False), ('NRMS', True, True), ('NRMS', False, False), ('LSTUR', True, False)])
def test_segmented_step_graphs_over_rccl_equal_eager_steps(tmp_path, model_name, overlap, table_rs):
    """graph.SegmentedStep on a process group over RCCL (world 1, multi-rank branch forced): [forward + backward + row staging] and [Adam]
    replayed as two HIP graphs with the collectives issued between them == the eager data-parallel step on the same device step counter,
    up to the run-to-run noise of the embedding scatter's fp32 atomics.  What it exercises: capture next to RCCL's watchdog thread, the
    stream ordering graph A -> collectives on RCCL's stream -> graph B, static row-exchange buffers, the row-sparse Adam under the counter."""
    # overlap: the THREE-segment mode (r06): graph A ends with the embedding scatter, the table exchange (all-reduce, or reduce-scatter with
    # table_rs) is started behind it, graph W replays the postponed weight-gradient GEMMs under it, graph B is Adam (+ the gather of the table)
    eager, graph = _run_seg('eager', model_name, tmp_path, overlap, table_rs), _run_seg('graph', model_name, tmp_path, overlap, table_rs)
    again = _run_seg('eager', model_name, tmp_path, overlap, table_rs)
    assert np.isfinite(eager['losses']).all() and len(set(np.round(eager['losses'], 6))) > 1
    assert set(eager) == set(graph)
    noise = max(float(np.abs(again[k].astype(np.float64) - eager[k]).max()) for k in eager)
    tol = max(4 * noise, 2e-6)
    for k in eager:
        err = float(np.abs(eager[k].astype(np.float64) - graph[k]).max())
        assert err <= tol, f'{model_name}: {k} differs between segment replays and eager steps by {err:.3g} (run-to-run noise {noise:.3g})'
