"""Pin the oracle (oracle/nrms_numpy.py, oracle/nrms_torch.py, oracle/metrics.py)
against golden vectors produced by the imported reference (oracle/make_golden.py)."""
import os
import numpy as np
import pytest
import torch

from oracle import nrms_numpy as onp
from oracle import metrics as om
from oracle.nrms_torch import OracleNRMS
from oracle.make_golden import CASES


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f'nrms_{name}.npz'))
    V, d, H, Q, B, C, N, L, seed = CASES[name]
    rng = np.random.default_rng(seed)
    params = onp.random_nrms_params(rng, V, d, Q, np.float32, emb_std=0.5)
    return g, params, (V, d, H, Q, B, C, N, L)


@pytest.mark.parametrize('name', ['tiny', 'base'])
@pytest.mark.parametrize('prec', ['f64', 'f32'])
def test_numpy_forward_backward_matches_reference(golden_dir, name, prec):
    g, params, (V, d, H, Q, B, C, N, L) = _load(golden_dir, name)
    dt = np.float64 if prec == 'f64' else np.float32
    p = {k: v.astype(dt) for k, v in params.items()}
    logits, cache = onp.nrms_forward(g['cand_ids'], g['click_ids'], p, H)
    tol = dict(rtol=1e-9, atol=1e-11) if prec == 'f64' else dict(rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(logits, g[f'{prec}_logits'], **tol)
    np.testing.assert_allclose(cache['user_vec'], g[f'{prec}_user_vec'], **tol)
    np.testing.assert_allclose(cache['cand_vec'].reshape(B * C, -1), g[f'{prec}_news_vec'], **tol)
    loss, dlogits = onp.cross_entropy_target0(logits)
    np.testing.assert_allclose(loss, g[f'{prec}_loss'], **tol)
    grads = onp.nrms_backward(dlogits.astype(dt), cache, p, H)
    gtol = dict(rtol=1e-7, atol=1e-10) if prec == 'f64' else dict(rtol=2e-3, atol=2e-6)
    checked = 0
    for k in p:
        if f'{prec}_grad/{k}' in g:
            np.testing.assert_allclose(grads[k], g[f'{prec}_grad/{k}'], err_msg=k, **gtol)
            checked += 1
        else:
            np.testing.assert_allclose(np.linalg.norm(grads[k].astype(np.float64)),
                                       g[f'{prec}_gradnorm/{k}'], rtol=gtol['rtol'])
            np.testing.assert_allclose(grads[k][:8, :16], g[f'{prec}_gradslice/{k}'], **gtol)
            if k.endswith('word_embedding.weight'):
                assert np.all(grads[k][0] == 0) and np.all(g[f'{prec}_gradrow0/{k}'] == 0)
                np.testing.assert_allclose(grads[k].sum(axis=1), g[f'{prec}_gradrowsum/{k}'],
                                           rtol=gtol['rtol'], atol=1e-5 if prec == 'f32' else 1e-10)
            checked += 1
    assert checked == len(p)


@pytest.mark.parametrize('name', ['tiny', 'base'])
def test_torch_port_matches_reference(golden_dir, name):
    g, params, (V, d, H, Q, B, C, N, L) = _load(golden_dir, name)
    m = OracleNRMS(V, d, H, Q, 0.2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})  # same keys as the reference
    m.eval()
    cand = [{'title': torch.from_numpy(g['cand_ids'][:, j])} for j in range(C)]
    click = [{'title': torch.from_numpy(g['click_ids'][:, j])} for j in range(N)]
    logits = m(cand, click)
    np.testing.assert_allclose(logits.detach().numpy(), g['f32_logits'], rtol=1e-5, atol=1e-6)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(B, dtype=torch.long))
    loss.backward()
    for k, p in m.named_parameters():
        if f'f32_grad/{k}' in g:
            np.testing.assert_allclose(p.grad.numpy(), g[f'f32_grad/{k}'], rtol=1e-4, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(p.grad.numpy()[:8, :16], g[f'f32_gradslice/{k}'], rtol=1e-4, atol=1e-6)
    nv = m.get_news_vector({'title': torch.from_numpy(g['cand_ids'].reshape(-1, L))})
    np.testing.assert_allclose(nv.detach().numpy(), g['f32_news_vec'], rtol=1e-5, atol=1e-6)
    pr = m.get_prediction(nv[:C], torch.from_numpy(g['f32_user_vec'][0]))
    np.testing.assert_allclose(pr.detach().numpy(), g['f32_pred0'], rtol=1e-4, atol=1e-5)


def test_dropout_mask_semantics_against_torch():
    """F.dropout keep/scale semantics used by the oracle's explicit masks
    (src/model/NRMS/news_encoder.py:38-45)."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(4, 6, 8)).astype(np.float32)
    torch.manual_seed(3)
    y = torch.nn.functional.dropout(torch.from_numpy(x), p=0.2, training=True).numpy()
    keep = (y != 0).astype(np.float32)
    np.testing.assert_allclose(x * keep * np.float32(1 / 0.8), y, rtol=1e-6)


def test_metrics_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    off = 0
    for n, ref in zip(g['lens'], g['res']):
        y, s = g['y'][off:off + n], g['s'][off:off + n]
        off += n
        got = om.single_impression_metrics(y, s)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0, equal_nan=True)
