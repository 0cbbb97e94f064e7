"""HIP graph of the training step (news_recommendation_amd/graph.py): replays == the eager loop on the same device step counter -- new
dropout masks and the right Adam step index at every replay -- for NRMS, NAML and LSTUR (whose row-sparse Adam, whole-row user mask and history
lengths all stay on the device).  "Equal" up to the run-to-run noise of the embedding
scatter's fp32 atomics (measured with a second eager run): a replay that reused a mask or a step index would be off by the size of an
update (1e-3), not by last bits."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
mode, model_name, out = sys.argv[2], sys.argv[3], sys.argv[4]
from bench import Workload, make_cfg
from news_recommendation_amd import ops, optim
from news_recommendation_amd.graph import StepGraph
ops.new_seed = lambda: 0x1234ABCD5678            # same base seeds in both modes: the masks differ from step to step through the counter only
optim.TABLE_MIN_NUMEL = 1 << 18
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
cfg = make_cfg(model_name, 'small', vocab=4000)
cfg.num_news = 2500
wl = Workload(model_name, cfg)
model = wl.make_model(seed=7).to(dev).train()
opt = wl.make_optimizer(model)
B = 48
batches = wl.batches(0, 3, B, dev)
target = torch.zeros(B, dtype=torch.long, device=dev)
crit = torch.nn.CrossEntropyLoss()
flat = lambda b: [b[s][a] for s in ('cand', 'click') for a in wl.attrs] + ([b['user'], b['length'].to(dev)] if model_name == 'LSTUR' else [])
def step_fn(*xs):
    n = len(wl.attrs)
    cand, click = dict(zip(wl.attrs, xs[:n])), dict(zip(wl.attrs, xs[n:2 * n]))
    if model_name == 'LSTUR':          # device-resident user ids and history lengths: nothing in the step touches the host (graph.py)
        logits = model.forward_ids(xs[2 * n], xs[2 * n + 1].clone(), cand, click)
    else:
        logits = model.forward_ids(cand['title'], click['title']) if model_name == 'NRMS' else model.forward_ids(cand, click)
    loss = crit(logits, target)
    loss.backward()
    opt.step()
    return loss
g = StepGraph(step_fn, flat(batches[0]), opt, warmup=2)
assert opt.t == 2
losses = []
for i in range(5):
    xs = flat(batches[i % 3])
    loss = g(*xs) if mode == 'graph' else g.eager_step(*xs)
    losses.append(float(loss.item()))
    assert not opt.flat_g.any()
    assert all(not st.pending for st in opt.sparse)
assert opt.t == 7 and int(g.ctr.item()) == 7, (opt.t, int(g.ctr.item()))
# masks are reproducible through the exported-mask entry point under the counter, and differ from step to step
m7 = torch.empty(4096, device=dev); lib = g.lib
lib.nr_dropout_mask(m7.data_ptr(), 4096, 0.2, 99, 1, torch.cuda.current_stream().cuda_stream)
lib.nr_step_counter_add(g.ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
m8 = torch.empty(4096, device=dev)
lib.nr_dropout_mask(m8.data_ptr(), 4096, 0.2, 99, 1, torch.cuda.current_stream().cuda_stream)
assert not torch.equal(m7, m8) and abs(float(m7.mean()) - 0.8) < 0.05
g.close()
m0 = torch.empty(4096, device=dev); m0b = torch.empty(4096, device=dev)
lib.nr_dropout_mask(m0.data_ptr(), 4096, 0.2, 99, 1, torch.cuda.current_stream().cuda_stream)
lib.nr_dropout_mask(m0b.data_ptr(), 4096, 0.2, 99, 1, torch.cuda.current_stream().cuda_stream)
assert torch.equal(m0, m0b) and not torch.equal(m0, m7)
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
osd = opt.state_dict()
for i, st in osd['state'].items():
    sd[f'opt/{i}/exp_avg_sq'] = st['exp_avg_sq'].cpu().numpy()
    assert float(st['step']) == 7.0
sd['losses'] = np.array(losses)
np.savez(out, **sd)
print('ok', mode, losses)
'''


def _run(mode, model_name, tmp_path):
    out = str(tmp_path / f'{mode}_{model_name}.npz')
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, mode, model_name, out], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    return dict(np.load(out))


@pytest.mark.parametrize('model_name', ['NRMS', 'NAML', 'LSTUR'])
def test_step_graph_replays_equal_eager_steps(tmp_path, model_name):
    eager, graph = _run('eager', model_name, tmp_path), _run('graph', model_name, tmp_path)
    again = _run('eager', model_name, tmp_path)
    assert np.isfinite(eager['losses']).all() and len(set(np.round(eager['losses'], 6))) > 1
    assert set(eager) == set(graph)
    noise = max(float(np.abs(again[k].astype(np.float64) - eager[k]).max()) for k in eager)
    tol = max(4 * noise, 2e-6)
    for k in eager:
        err = float(np.abs(eager[k].astype(np.float64) - graph[k]).max())
        assert err <= tol, f'{model_name}: {k} differs between graph replays and eager steps by {err:.3g} (run-to-run noise {noise:.3g})'
