"""The model-geometry knobs of src/config.py away from the tuned instantiation: word_embedding_dim in {100, 200, 300},
num_attention_heads any divisor with d_k <= 32, num_filters in {256, 300, 400}, window_size in {1, 3, 5}, query_vector_dim up to 256 -- the
drop-in models on the general-geometry path (news_recommendation_amd/ops_generic.py, csrc/k_generic.h) against the CPU fp32 oracles, whose
code is dimension-generic (pinned to the imported reference at OTHER dimensions too: tests/golden/*_tiny.npz are d = 60 / 32 cases).
Also here: the kernels themselves on MI355X (the checks tests/test_generic_emu.py runs on the emulator), the cross-attention form of
MultiHeadSelfAttention and AdditiveAttention's tensorboard hook.  Tolerances: bf16 GEMM operands, fp32 everything else -- as on the tuned path."""
import copy

import numpy as np
import pytest
import torch

from tests import kernel_checks_generic as kg
from tests.test_model_gpu import rel_err, grad_floor

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def be():
    from tests.backends import GpuBackend
    return GpuBackend()


def test_kernels_attention(be):
    kg.check_attn(be, n_seq=37, S=20, H=10, dk=20)
    kg.check_attn(be, n_seq=9, S=64, H=4, dk=32, with_len=False)
    kg.check_attn(be, n_seq=5, S=50, H=25, dk=4)


def test_kernels_pooling_dropout_relu(be):
    kg.check_additive(be, n_seq=33, S=50, D=400, Q=256)
    kg.check_additive(be, n_seq=6, S=20, D=100, Q=200, valid=13)
    kg.check_dropout(be, n=4 * 100003)
    kg.check_relu(be, n=100001)
    kg.check_split_linear(be, n=159, D=100, N=256)


@pytest.mark.parametrize('w,D,F', [(1, 100, 256), (3, 200, 400), (5, 300, 256), (5, 100, 300)])
def test_kernels_convolution(be, w, D, F):
    kg.check_conv(be, n_seq=40, S=20, D=D, F=F, w=w)


def _grads_close(m, ref, bound, floor_scale=2e-2):
    rg = {k: p.grad.numpy() for k, p in ref.named_parameters()}
    fl = grad_floor(rg)
    worst = ('', 0.0)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        e = rel_err(p.grad.cpu().numpy(), rg[k], fl)
        worst = max(worst, (k, e), key=lambda kv: kv[1])
        assert e < bound, (k, e)
    return worst


@pytest.mark.parametrize('d,heads,qdim', [(100, 5, 200), (200, 10, 256), (300, 10, 200), (300, 15, 256), (200, 25, 64)])
def test_nrms_other_geometry(d, heads, qdim):
    """NRMS with word_embedding_dim / num_attention_heads / query_vector_dim off the tuned values (config.py:34,39,45): logits and every
    gradient vs OracleNRMS (multihead_self.py / additive.py restated for any dimension), eval and the three evaluation entry points."""
    from oracle import nrms_numpy as onp
    from oracle.make_golden import make_cfg
    from oracle.nrms_torch import OracleNRMS
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    from tests.test_model_gpu import as_lists, mind_batch
    V, B, C, N, L = 3000, 5, 3, 50, 20
    rng = np.random.default_rng(d + heads)
    params = onp.random_nrms_params(rng, V, d, qdim, np.float32, emb_std=0.4)
    cand, click = mind_batch(rng, B, C=C, N=N, L=L, V=V)
    ref = OracleNRMS(V, d, heads, qdim, 0.2)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ref.eval()
    lr = ref(as_lists(cand), as_lists(click))
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(B, dtype=torch.long)).backward()
    m = NRMS(make_cfg(V, d, heads, qdim, N, L, 0.2))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to(DEV).eval()
    lg = m(as_lists(cand), as_lists(click))
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(B, dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1.5e-2
    _grads_close(m, ref, 5e-2)
    with torch.no_grad():
        t = {'title': torch.from_numpy(cand.reshape(-1, L))}
        assert rel_err(m.get_news_vector(t).cpu().numpy(), ref.get_news_vector(t).numpy()) < 1.5e-2
        hv = torch.randn(4, N, d)
        assert rel_err(m.get_user_vector(hv.to(DEV)).cpu().numpy(), ref.get_user_vector(hv).numpy()) < 1.5e-2
    # training mode: the general path draws the engine's counter-based masks (reproducible from the seed, changing from call to call)
    m.train()
    torch.manual_seed(5)
    a = m(as_lists(cand), as_lists(click))
    torch.manual_seed(5)
    assert torch.equal(a, m(as_lists(cand), as_lists(click))) and not torch.equal(a, m(as_lists(cand), as_lists(click)))


def test_nrms_other_geometry_training_masks_match_oracle():
    """Train mode at d = 200 / 10 heads: the general path's dropout (both sites) reproduced in the oracle via nr_dropout_mask."""
    from oracle import nrms_numpy as onp
    from oracle.make_golden import make_cfg
    from oracle.nrms_torch import OracleNRMS
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    from tests.backends import GpuBackend
    from tests.kernel_checks import export_mask
    from tests.test_model_gpu import as_lists, mind_batch
    V, B, C, N, L, d, heads, qdim = 2000, 3, 3, 50, 20, 200, 10, 200
    rng = np.random.default_rng(8)
    params = onp.random_nrms_params(rng, V, d, qdim, np.float32, emb_std=0.4)
    cand, click = mind_batch(rng, B, C=C, N=N, L=L, V=V)
    m = NRMS(make_cfg(V, d, heads, qdim, N, L, 0.2))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to(DEV).train()
    torch.manual_seed(77)
    lg = m(as_lists(cand), as_lists(click))
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    T = B * (C + N)
    be = GpuBackend()
    m1 = export_mask(be, T * L * d, 0.2, seed, 1).reshape(T, L, d)
    m2 = export_mask(be, T * L * d, 0.2, seed, 2).reshape(T, L, d)
    keeps = []
    for j in range(C + N):
        idx = np.arange(B) * C + j if j < C else B * C + np.arange(B) * N + (j - C)
        keeps.append({'title1': torch.from_numpy(m1[idx]), 'title2': torch.from_numpy(m2[idx])})
    ref = OracleNRMS(V, d, heads, qdim, 0.2)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ref.train()
    lr = ref(as_lists(cand), as_lists(click), keeps)
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1.5e-2
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(B, dtype=torch.long)).backward()
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(B, dtype=torch.long, device=DEV)).backward()
    _grads_close(m, ref, 5e-2)


@pytest.mark.parametrize('d,F,window,qdim', [(100, 256, 1, 200), (200, 400, 5, 256), (300, 300, 5, 200), (300, 256, 3, 200)])
def test_naml_other_geometry(d, F, window, qdim):
    """NAML with word_embedding_dim / num_filters / window_size / query_vector_dim off the tuned values (config.py:34,39,54,55)."""
    from oracle.naml_torch import OracleNAML, random_naml_params
    from oracle.make_golden_naml_lstur import as_lists, synth_batch
    from tests.test_naml_gpu import build
    c = dict(V=3000, d=d, ncat=40, dcat=100, F=F, window=window, Q=qdim, C=3, N=50, L=20, La=50, B=3, seed=d + F)
    params = random_naml_params(c['seed'], c['V'], d, c['ncat'], c['dcat'], F, window, qdim, emb_std=0.3)
    cand, click, _ = synth_batch(np.random.default_rng(c['seed']), c, True)
    cl, hl = as_lists(cand, click)
    ref = OracleNAML(c['V'], d, c['ncat'], c['dcat'], F, window, qdim, 0.2)
    ref.load_state_dict(params)
    for te in ref.news_encoder.text_encoders.values():
        te.CNN.q_operands = True                                      # same relu masks as the engine's bf16 conv operands (tests/test_naml_gpu.py)
    ref.eval()
    lr = ref(cl, hl)
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(c['B'], dtype=torch.long)).backward()
    m = build(c, params).eval()
    lg = m(cl, hl)
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(c['B'], dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 5e-3
    _grads_close(m, ref, 6e-2)
    with torch.no_grad():
        flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
        assert rel_err(m.get_news_vector(flat).cpu().numpy(), ref.get_news_vector(flat).numpy()) < 5e-3
    m.train()                                                         # training mode runs (masks drawn per text encoder call)
    assert torch.isfinite(m(cl, hl)).all()


@pytest.mark.parametrize('F,window,method', [(256, 5, 'ini'), (400, 1, 'con')])
def test_lstur_other_geometry(F, window, method):
    """LSTUR with num_filters / window_size off the tuned values: the conv title encoder on the general path, category rows of width F, and
    the GRU at hidden sizes 3 F / 1.5 F other than 900 / 450 (step-per-launch kernels)."""
    from oracle.lstur_torch import random_lstur_params
    from oracle.make_golden_naml_lstur import as_lists, synth_batch
    from tests.test_lstur_gpu import MIND, build, oracle
    c = dict(MIND, V=3000, nusers=101, B=4, seed=F, F=F, window=window, method=method, ncat=40)
    params = random_lstur_params(c['seed'], c['V'], c['d'], c['ncat'], c['nusers'], F, window, c['Q'], method, emb_std=0.3)
    rng = np.random.default_rng(c['seed'])
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
    assert rel_err(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1e-2
    _grads_close(m, ref, 8e-2)           # measured: title_CNN.weight 6.0e-2 (relu masks of bf16 conv operands at B = 4), everything else <= 3e-2


def test_cross_attention_and_tensorboard_hook():
    """MultiHeadSelfAttention.forward(Q, K, V) with K, V != Q (multihead_self.py:46-58) and AdditiveAttention's writer hook (additive.py:40-49):
    used by none of the reference's models, part of the modules' interface."""
    from news_recommendation_amd.dropin.model.general.attention.multihead_self import MultiHeadSelfAttention
    from news_recommendation_amd.dropin.model.general.attention.additive import AdditiveAttention
    torch.manual_seed(11)
    cpu = MultiHeadSelfAttention(300, 15)
    gpu = copy.deepcopy(cpu).to(DEV)
    B, S = 4, 20
    Q, K, V = (torch.randn(B, S, 300) * 0.5 for _ in range(3))
    sp = lambda t: t.view(B, S, 15, 20).transpose(1, 2)
    q, k, v = sp(cpu.W_Q(Q)), sp(cpu.W_K(K)), sp(cpu.W_V(V))
    e = torch.exp(q @ k.transpose(-1, -2) / np.sqrt(20))
    ref = ((e / (e.sum(-1, keepdim=True) + 1e-8)) @ v).transpose(1, 2).reshape(B, S, 300)
    got = gpu(Q.to(DEV), K.to(DEV), V.to(DEV))
    assert rel_err(got.detach().cpu().numpy(), ref.detach().numpy()) < 1.5e-2

    class Writer:
        def __init__(self): self.calls = []
        def add_scalars(self, tag, d, step): self.calls.append((tag, {k: float(v) for k, v in d.items()}, step))
    wr = Writer()
    att = AdditiveAttention(200, 300, writer=wr, tag='views', names=['a', 'b', 'c', 'd']).to(DEV)
    x = torch.randn(6, 4, 300, device=DEV)
    for _ in range(10):
        out = att(x)
    assert out.shape == (6, 300) and len(wr.calls) == 1 and wr.calls[0][0] == 'views' and wr.calls[0][2] == 10
    w = torch.softmax(torch.tanh(att.linear(x)) @ att.attention_query_vector, dim=1).mean(dim=0)
    np.testing.assert_allclose([wr.calls[0][1][k] for k in 'abcd'], w.detach().cpu().numpy(), rtol=2e-2, atol=1e-3)
