"""The reference's src/config.py knobs away from their defaults (north_star: "same src/config.py knobs"): title / abstract / history
lengths that are not instantiated kernel lengths (zero-padded to 20 / 50 by the host, masked as attention keys, zero vectors for the
convolution, outside every pooling), negative_sampling_ratio != 2, and the `length` argument of MultiHeadSelfAttention.forward
(multihead_self.py:60-70).  Every case: logits and every gradient of the drop-in model vs the CPU fp32 oracle run on the SAME
un-padded inputs -- the oracle never sees the padding."""
import numpy as np
import pytest
import torch

from tests.test_model_gpu import rel_err, grad_floor

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_nrms_title_12_history_30_k4():
    """num_words_title = 12, num_clicked_news_a_user = 30, negative_sampling_ratio = 4 (config.py:21,22,27)."""
    from oracle import nrms_numpy as onp
    from oracle.nrms_torch import OracleNRMS
    from tests.test_model_gpu import build, as_lists, mind_batch
    V, B, C, N, L = 5000, 9, 5, 30, 12
    rng = np.random.default_rng(61)
    params = onp.random_nrms_params(rng, V, 300, 200, np.float32, emb_std=0.4)
    cand, click = mind_batch(rng, B, C=C, N=N, L=L, V=V)
    ref = OracleNRMS(V, 300, 15, 200, 0.2)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ref.eval()
    lr = ref(as_lists(cand), as_lists(click))
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(B, dtype=torch.long)).backward()
    m = build(V, 300, 15, 200, N, L, params).eval()
    lg = m(as_lists(cand), as_lists(click))
    assert lg.shape == (B, C)
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(B, dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1.5e-2
    gref = dict(ref.named_parameters())
    fl = grad_floor({k: v.grad.numpy() for k, v in gref.items()})
    for k, p in m.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), gref[k].grad.numpy(), fl)
        assert e < 5e-2, (k, e)
    # eval entry points at the same lengths (evaluate.py:198,226-230)
    with torch.no_grad():
        nv = m.get_news_vector({'title': torch.from_numpy(cand.reshape(-1, L))})
        nv_ref = ref.get_news_vector({'title': torch.from_numpy(cand.reshape(-1, L))})
        assert rel_err(nv.cpu().numpy(), nv_ref.numpy()) < 1.5e-2
        hv = torch.randn(4, N, 300)
        assert rel_err(m.get_user_vector(hv.to(DEV)).cpu().numpy(), ref.get_user_vector(hv).numpy()) < 1.5e-2


def test_lstur_title_16_history_30():
    from oracle.lstur_torch import random_lstur_params
    from oracle.make_golden_naml_lstur import as_lists, synth_batch
    from tests.test_lstur_gpu import MIND, build, oracle, check_grads
    c = dict(MIND, V=5000, nusers=301, B=6, seed=52, N=30, L=16, C=2, method='ini')
    params = random_lstur_params(52, c['V'], c['d'], c['ncat'], c['nusers'], c['F'], c['window'], c['Q'], 'ini', emb_std=0.3)
    rng = np.random.default_rng(52)
    cand, click, hist = synth_batch(rng, c, False)
    user = torch.from_numpy(rng.integers(0, c['nusers'], size=c['B']).astype(np.int64))
    length = torch.from_numpy(hist)
    cl, hl = as_lists(cand, click)
    ref = oracle(c, params)
    lr = ref(user, length.clone(), cl, hl)
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(c['B'], dtype=torch.long)).backward()
    m = build(c, params).eval()
    lg = m(user, length.clone(), cl, hl)
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(c['B'], dtype=torch.long, device=DEV)).backward()
    with torch.no_grad():
        l_plain = oracle(c, params, q_operands=False)(user, length.clone(), cl, hl)
    assert rel_err(lg.detach().cpu().numpy(), l_plain.numpy()) < 2e-2
    check_grads(m, ref, 5e-2)


def test_naml_title_14_abstract_33_history_25_k3():
    from oracle.naml_torch import OracleNAML, random_naml_params
    from oracle.make_golden_naml_lstur import as_lists, synth_batch
    from tests.test_naml_gpu import MIND, build, oracle_with_engine_operands, check_grads
    c = dict(MIND, V=5000, B=6, seed=43, N=25, L=14, La=33, C=4)
    params = random_naml_params(43, c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], emb_std=0.3)
    cand, click, _ = synth_batch(np.random.default_rng(43), c, True)
    cl, hl = as_lists(cand, click)
    plain = OracleNAML(c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], 0.2)
    plain.load_state_dict(params)
    with torch.no_grad():
        l_plain = plain.eval()(cl, hl)
    ref = oracle_with_engine_operands(c, params)
    lr = ref(cl, hl)
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(c['B'], dtype=torch.long)).backward()
    m = build(c, params).eval()
    lg = m(cl, hl)
    assert lg.shape == (c['B'], c['C'])
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(c['B'], dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), l_plain.numpy()) < 1e-3          # measured 1.5e-4
    check_grads(m, {k: p.grad.numpy() for k, p in ref.named_parameters()}, 6e-2)      # measured <= 2.5e-2


@pytest.mark.parametrize('attrs', [('title', 'subcategory'), ('abstract',), ('category', 'subcategory', 'title')])
def test_naml_view_subsets(attrs):
    """dataset_attributes['news'] subsets (src/model/NAML/news_encoder.py:63-84,100-114): same state_dict keys as the oracle built for
    the subset, logits and gradients vs that oracle; a single view bypasses the final attention."""
    from oracle.naml_torch import OracleNAML
    from oracle.make_golden_naml_lstur import synth_batch
    from tests.test_naml_gpu import MIND, make_cfg
    from news_recommendation_amd.dropin.model.NAML import NAML
    c = dict(MIND, V=3000, B=5, seed=47, N=20, C=3)
    cfg = make_cfg(c)
    cfg.dataset_attributes = {"news": list(attrs), "record": []}
    torch.manual_seed(47)
    ref = OracleNAML(c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], 0.2, attrs=attrs).eval()
    for te in ref.news_encoder.text_encoders.values():
        te.CNN.q_operands = True            # conv operands rounded where the engine rounds them: same relu masks (tests/test_naml_gpu.py)
    m = NAML(cfg)
    assert set(m.state_dict()) == set(ref.state_dict())
    m.load_state_dict(ref.state_dict())
    m = m.to(DEV).eval()
    cand, click, _ = synth_batch(np.random.default_rng(47), c, True)
    cand = {k: v for k, v in cand.items() if k in attrs}
    click = {k: v for k, v in click.items() if k in attrs}
    lists = lambda d: [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in d.items()} for j in range(next(iter(d.values())).shape[1])]
    cl, hl = lists(cand), lists(click)
    lr = ref(cl, hl)
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(c['B'], dtype=torch.long)).backward()
    lg = m(cl, hl)
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(c['B'], dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1e-3          # measured <= 2.3e-4
    rg = {k: p.grad.numpy() for k, p in ref.named_parameters()}
    fl = grad_floor(rg)
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), rg[k], fl) < 9e-2, k           # measured <= 5.6e-2 (abstract-only model: one view carries every gradient)


def _ref_mhsa(x, mod, length):
    """multihead_self.py:15-23,46-75 restated in torch fp32 (exp / (sum + 1e-8), key mask from `length`, no output projection)."""
    B, S, D = x.shape
    H, dk = mod.num_attention_heads, mod.d_k
    sp = lambda t: t.view(B, S, H, dk).transpose(1, 2)
    q, k, v = sp(mod.W_Q(x)), sp(mod.W_K(x)), sp(mod.W_V(x))
    e = torch.exp(q @ k.transpose(-1, -2) / np.sqrt(dk))
    if length is not None:
        e = e * (torch.arange(S)[None, :] < length.view(-1, 1))[:, None, None, :]
    a = e / (e.sum(-1, keepdim=True) + 1e-8)
    return (a @ v).transpose(1, 2).reshape(B, S, D)


@pytest.mark.parametrize('S', [20, 50, 13, 37])
def test_multihead_self_attention_length_argument(S):
    """MultiHeadSelfAttention.forward(Q, length=...) on its own: forward and input / weight gradients vs the fp32 restatement, for
    instantiated (20, 50) and padded (13, 37) sequence lengths."""
    import copy
    from news_recommendation_amd.dropin.model.general.attention.multihead_self import MultiHeadSelfAttention
    torch.manual_seed(S)
    cpu = MultiHeadSelfAttention(300, 15)
    gpu = copy.deepcopy(cpu).to(DEV)
    B = 5
    x = (torch.randn(B, S, 300) * 0.5).requires_grad_(True)
    length = torch.tensor([S, 1, max(1, S // 2), S - 1, 3])
    g = torch.randn(B, S, 300) * 0.1
    y_ref = _ref_mhsa(x, cpu, length)
    y_ref.backward(g)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = gpu(xg, length=length)
    assert y.shape == (B, S, 300)
    y.backward(g.to(DEV))
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1.5e-2
    assert rel_err(xg.grad.cpu().numpy(), x.grad.numpy()) < 4e-2
    fl = 2e-2 * float(cpu.W_V.bias.grad.abs().max())
    for (k, p), (_, pr) in zip(gpu.named_parameters(), cpu.named_parameters()):
        assert rel_err(p.grad.cpu().numpy(), pr.grad.numpy(), fl) < 5e-2, k


def test_additive_attention_any_length():
    """AdditiveAttention.forward on [batch, 9, 300] (padded to 20) vs additive.py:27-53 in fp32."""
    import copy
    from news_recommendation_amd.dropin.model.general.attention.additive import AdditiveAttention
    torch.manual_seed(3)
    cpu = AdditiveAttention(200, 300)
    gpu = copy.deepcopy(cpu).to(DEV)
    x = (torch.randn(7, 9, 300) * 0.5).requires_grad_(True)
    w = torch.softmax(torch.tanh(cpu.linear(x)) @ cpu.attention_query_vector, dim=1)
    ref = torch.bmm(w.unsqueeze(1), x).squeeze(1)
    g = torch.randn(7, 300)
    ref.backward(g)
    xg = x.detach().to(DEV).requires_grad_(True)
    out = gpu(xg)
    out.backward(g.to(DEV))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-2
    assert rel_err(xg.grad.cpu().numpy(), x.grad.numpy()) < 4e-2
