"""Pin oracle/naml_torch.py and oracle/lstur_torch.py against golden vectors produced by the imported reference
(oracle/make_golden_naml_lstur.py): outputs, loss and every parameter gradient, fp64 (tight) and fp32."""
import os
import numpy as np
import pytest
import torch

from oracle.naml_torch import OracleNAML, random_naml_params
from oracle.lstur_torch import OracleLSTUR, random_lstur_params
from oracle.make_golden_naml_lstur import NAML_CASES, LSTUR_CASES, as_lists


def _check_grads(g, tag, model, gtol):
    checked = 0
    for k, p in model.named_parameters():
        gr = p.grad.detach().numpy()
        if f'{tag}_grad/{k}' in g:
            np.testing.assert_allclose(gr, g[f'{tag}_grad/{k}'], err_msg=k, **gtol)
        else:
            np.testing.assert_allclose(np.linalg.norm(gr.astype(np.float64)), g[f'{tag}_gradnorm/{k}'], rtol=gtol['rtol'], err_msg=k)
            np.testing.assert_allclose(gr.reshape(gr.shape[0], -1)[:8, :16], g[f'{tag}_gradslice/{k}'], err_msg=k, **gtol)
            if 'embedding' in k:
                assert np.all(gr[0] == 0) and np.all(g[f'{tag}_gradrow0/{k}'] == 0), k      # padding_idx=0 row gets no gradient
                np.testing.assert_allclose(gr.sum(axis=1), g[f'{tag}_gradrowsum/{k}'], rtol=gtol['rtol'],
                                           atol=1e-4 if tag == 'f32' else 1e-10, err_msg=k)
        checked += 1
    assert checked == len(list(model.named_parameters()))


def _tols(tag):
    if tag == 'f64':
        return dict(rtol=1e-9, atol=1e-11), dict(rtol=1e-7, atol=1e-10)
    return dict(rtol=3e-4, atol=3e-5), dict(rtol=3e-3, atol=3e-6)


@pytest.mark.parametrize('name', list(NAML_CASES))
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_naml_oracle_matches_reference(golden_dir, name, tag):
    c = NAML_CASES[name]
    g = np.load(os.path.join(golden_dir, f'naml_{name}.npz'))
    dt = torch.float64 if tag == 'f64' else torch.float32
    m = OracleNAML(c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], 0.2)
    m.load_state_dict(random_naml_params(c['seed'], c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q']))   # reference keys
    m = m.to(dt).eval()
    cand = {k[5:]: g[k] for k in g.files if k.startswith('cand_')}
    click = {k[6:]: g[k] for k in g.files if k.startswith('click_')}
    cl, hl = as_lists(cand, click)
    logits = m(cl, hl)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(c['B'], dtype=torch.long))
    loss.backward()
    tol, gtol = _tols(tag)
    np.testing.assert_allclose(logits.detach().numpy(), g[f'{tag}_logits'], **tol)
    np.testing.assert_allclose(loss.item(), g[f'{tag}_loss'], **tol)
    flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
    nv = m.get_news_vector(flat)
    np.testing.assert_allclose(nv.detach().numpy(), g[f'{tag}_news_vec'], **tol)
    uv = m.get_user_vector(torch.stack([m.get_news_vector(x) for x in hl], dim=1))
    np.testing.assert_allclose(uv.detach().numpy(), g[f'{tag}_user_vec'], **tol)
    np.testing.assert_allclose(m.get_prediction(nv[:c['C']], uv[0]).detach().numpy(), g[f'{tag}_pred0'], **tol)
    _check_grads(g, tag, m, gtol)


@pytest.mark.parametrize('name', list(LSTUR_CASES))
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_lstur_oracle_matches_reference(golden_dir, name, tag):
    c = LSTUR_CASES[name]
    g = np.load(os.path.join(golden_dir, f'lstur_{name}.npz'))
    dt = torch.float64 if tag == 'f64' else torch.float32
    m = OracleLSTUR(c['V'], c['d'], c['ncat'], c['nusers'], c['F'], c['window'], c['Q'], 0.2, 0.5, c['method'])
    m.load_state_dict(random_lstur_params(c['seed'], c['V'], c['d'], c['ncat'], c['nusers'], c['F'], c['window'], c['Q'], c['method']))
    m = m.to(dt).eval()
    cand = {k[5:]: g[k] for k in g.files if k.startswith('cand_')}
    click = {k[6:]: g[k] for k in g.files if k.startswith('click_')}
    cl, hl = as_lists(cand, click)
    user, length = torch.from_numpy(g['user']), torch.from_numpy(g['clicked_news_length'])
    logits = m(user, length, cl, hl)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(c['B'], dtype=torch.long))
    loss.backward()
    tol, gtol = _tols(tag)
    np.testing.assert_allclose(logits.detach().numpy(), g[f'{tag}_logits'], **tol)
    np.testing.assert_allclose(loss.item(), g[f'{tag}_loss'], **tol)
    flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
    np.testing.assert_allclose(m.get_news_vector(flat).detach().numpy(), g[f'{tag}_news_vec'], **tol)
    cv = torch.stack([m.get_news_vector(x) for x in hl], dim=1)
    np.testing.assert_allclose(m.get_user_vector(user, length, cv).detach().numpy(), g[f'{tag}_user_vec'], **tol)
    _check_grads(g, tag, m, gtol)
    assert torch.equal(length, torch.from_numpy(g['clicked_news_length']))     # the oracle does not mutate the caller's lengths
