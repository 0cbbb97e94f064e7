"""GPU parity of the drop-in NAML modules against (a) golden vectors from the imported reference (tests/golden/naml_base.npz,
oracle/make_golden_naml_lstur.py) and (b) the CPU torch oracle (oracle/naml_torch.py), incl. train mode with the kernels' own
dropout masks exported through nr_dropout_mask.  Tolerances: bf16 operands / fp32 accumulation, stated next to each check."""
import os
import numpy as np
import pytest
import torch

from oracle.naml_torch import OracleNAML, random_naml_params
from oracle.make_golden_naml_lstur import NAML_CASES, as_lists, synth_batch
from tests.test_model_gpu import rel_err, grad_floor

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make_cfg(c, p=0.2):
    class Cfg:
        dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}
        num_words, word_embedding_dim = c['V'], c['d']
        num_categories, category_embedding_dim = c['ncat'], c['dcat']
        num_filters, window_size, query_vector_dim = c['F'], c['window'], c['Q']
        dropout_probability = p
        num_clicked_news_a_user, num_words_title, num_words_abstract = c['N'], c['L'], c['La']
    return Cfg


def build(c, params, p=0.2):
    from news_recommendation_amd.dropin.model.NAML import NAML
    m = NAML(make_cfg(c, p))
    m.load_state_dict(params)                                    # reference key names (both aliases of the shared tables)
    return m.to(DEV)


def oracle_with_engine_operands(c, params, train=False):
    """The pinned oracle with the conv operands rounded to bf16 where the engine rounds them (OracleConv.q_operands): same
    relu masks as the engine, so gradient parity is not at the mercy of pre-activations that sit within rounding noise of 0."""
    ref = OracleNAML(c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], 0.2)
    ref.load_state_dict(params)
    for te in ref.news_encoder.text_encoders.values():
        te.CNN.q_operands = True
    return ref.train(train)


def check_grads(m, ref_grads, bound):
    fl = grad_floor(ref_grads)
    seen = set()
    for k, p in m.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), ref_grads[k], fl)
        assert e < bound, (k, e)
        seen.add(k)
    return seen


def test_golden_base_forward_and_grads(golden_dir):
    c = NAML_CASES['base']
    g = np.load(os.path.join(golden_dir, 'naml_base.npz'))
    params = random_naml_params(c['seed'], c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'])
    m = build(c, params).eval()
    assert set(m.state_dict()) == set(params)
    cand = {k[5:]: g[k] for k in g.files if k.startswith('cand_')}
    click = {k[6:]: g[k] for k in g.files if k.startswith('click_')}
    cl, hl = as_lists(cand, click)
    logits = m(cl, hl)
    assert logits.shape == (c['B'], c['C']) and logits.is_cuda
    # bf16 operands through two pooling levels: 1.5e-2 of the logit scale
    assert rel_err(logits.detach().cpu().numpy(), g['f32_logits']) < 1e-3          # measured 1.4e-4 (bounds: ~3x profiles/r05_measured_rel_err.txt, rounded up)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(c['B'], dtype=torch.long, device=DEV))
    loss.backward()
    # gradients vs the fp32 reference: recompute them with the pinned oracle (the golden file stores norms/slices for big tensors)
    ref = oracle_with_engine_operands(c, params)
    lr = ref(cl, hl)
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(c['B'], dtype=torch.long)).backward()
    np.testing.assert_allclose(lr.detach().numpy(), g['f32_logits'], rtol=0, atol=1e-2 * np.abs(g['f32_logits']).max())
    rg = {k: p.grad.numpy() for k, p in ref.named_parameters()}
    # the logits here are ~9 in magnitude with near-one-hot softmax: gradients carry the bf16 error of the logits -> 6e-2 bound
    check_grads(m, rg, 6e-2)              # measured <= 2.4e-2 (the pooling layers' bias gradients over their floor); 5e-2 below: <= 3.0e-2
    with torch.no_grad():
        flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
        nv = m.get_news_vector(flat)
        assert rel_err(nv.cpu().numpy(), g['f32_news_vec']) < 7e-3          # measured 2.3e-3
        cv = torch.stack([m.get_news_vector(x) for x in hl], dim=1)
        uv = m.get_user_vector(cv)
        assert rel_err(uv.cpu().numpy(), g['f32_user_vec']) < 3e-3          # measured 8.4e-4
        pr = m.get_prediction(nv[:c['C']], uv[0])
        assert pr.shape == (c['C'],) and np.abs(pr.cpu().numpy() - g['f32_pred0']).max() < 1.5e-2 * np.abs(g['f32_logits']).max()


MIND = dict(V=70976, d=300, ncat=275, dcat=100, F=300, window=3, Q=200, C=3, N=50, L=20, La=50)


def test_mind_shape_vs_torch_oracle():
    """MIND-small shapes at B=6 (ragged last workgroups everywhere): logits and every gradient vs the CPU fp32 oracle."""
    c = dict(MIND, B=6, seed=41)
    params = random_naml_params(41, c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], emb_std=0.3)
    cand, click, _ = synth_batch(np.random.default_rng(41), c, True)
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
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(c['B'], dtype=torch.long, device=DEV)).backward()
    assert rel_err(lg.detach().cpu().numpy(), l_plain.numpy()) < 1e-3          # vs the un-quantised fp32 reference math; measured 1.7e-4
    check_grads(m, {k: p.grad.numpy() for k, p in ref.named_parameters()}, 5e-2)
    we = m.news_encoder.text_encoders['title'].word_embedding.weight
    assert torch.all(we.grad[0] == 0)                                           # padding_idx row of the word table
    assert torch.all(m.news_encoder.element_encoders['category'].embedding.weight.grad[0] == 0)


def test_training_mode_dropout_matches_oracle_with_exported_masks():
    from tests.backends import GpuBackend
    from tests.kernel_checks import export_mask
    c = dict(MIND, V=3000, B=4, seed=42)
    B, C, N, L, La = c['B'], c['C'], c['N'], c['L'], c['La']
    params = random_naml_params(42, c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], emb_std=0.3)
    cand, click, _ = synth_batch(np.random.default_rng(42), c, True)
    cl, hl = as_lists(cand, click)
    m = build(c, params, p=0.2).train()
    torch.manual_seed(77)
    l1 = m(cl, hl)
    torch.manual_seed(77)
    assert torch.equal(l1, m(cl, hl))                             # same seed -> same masks
    assert not torch.equal(l1, m(cl, hl))
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # what ops.new_seed() drew
    T = B * (C + N)
    be = GpuBackend()
    ntok = T * L + T * La
    m1 = export_mask(be, ntok * 300, 0.2, seed, 1)
    m2 = export_mask(be, ntok * 300, 0.2, seed, 2)
    t1, t2 = m1[:T * L * 300].reshape(T, L, 300), m2[:T * L * 300].reshape(T, L, 300)
    a1, a2 = m1[T * L * 300:].reshape(T, La, 300), m2[T * L * 300:].reshape(T, La, 300)
    keeps = []
    for j in range(C + N):                                        # engine news order: candidates b*C+c, then clicked B*C + b*N + n
        idx = np.arange(B) * C + j if j < C else B * C + np.arange(B) * N + (j - C)
        keeps.append({'title1': torch.from_numpy(t1[idx]), 'title2': torch.from_numpy(t2[idx]),
                      'abstract1': torch.from_numpy(a1[idx]), 'abstract2': torch.from_numpy(a2[idx])})
    ref = oracle_with_engine_operands(c, params, train=True)
    lr = ref(cl, hl, keeps)
    assert rel_err(l1.detach().cpu().numpy(), lr.detach().numpy()) < 1e-3          # measured 1.2e-4
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(B, dtype=torch.long)).backward()
    m.zero_grad()
    torch.nn.CrossEntropyLoss()(l1, torch.zeros(B, dtype=torch.long, device=DEV)).backward()
    check_grads(m, {k: p.grad.numpy() for k, p in ref.named_parameters()}, 5e-2)


def test_optimizer_step_sees_live_parameters_and_text_encoder_alone():
    c = dict(MIND, V=2000, B=4, seed=43)
    params = random_naml_params(43, c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], emb_std=0.3)
    cand, click, _ = synth_batch(np.random.default_rng(43), c, True)
    cl, hl = as_lists(cand, click)
    m = build(c, params).eval()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    tgt = torch.zeros(c['B'], dtype=torch.long, device=DEV)
    loss0 = torch.nn.CrossEntropyLoss()(m(cl, hl), tgt)
    opt.zero_grad(); loss0.backward(); opt.step()
    l1 = m(cl, hl)
    assert torch.nn.CrossEntropyLoss()(l1, tgt).item() < loss0.item()
    ref = OracleNAML(c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'], 0.2)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref.eval()
    assert rel_err(l1.detach().cpu().numpy(), ref(cl, hl).detach().numpy()) < 1e-3          # measured 9.0e-5
    # TextEncoder.forward on its own
    te, rte = m.news_encoder.text_encoders['abstract'], ref.news_encoder.text_encoders['abstract']
    ids = torch.from_numpy(cand['abstract'][:, 0])
    with torch.no_grad():
        assert rel_err(te(ids).cpu().numpy(), rte(ids).numpy()) < 4e-3          # measured 1.3e-3
