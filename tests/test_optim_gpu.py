"""GPU parity of the optimiser / sort kernels and of EngineAdam on the real drop-in models, plus the MIND-LARGE-shaped parity cases
(BASELINE.json configs[3] / configs[4]: vocabulary 130,001, 711,223-row user table) that the multi-GPU bench runs."""
import numpy as np
import pytest
import torch

from oracle import adam_numpy as oadam
from oracle import nrms_numpy as onp
from oracle.nrms_torch import OracleNRMS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
F = np.float32


@pytest.fixture(scope='module')
def lib():
    from news_recommendation_amd import _capi
    return _capi.load()


def _sched(lr, betas, n):
    from news_recommendation_amd.optim import AdamSchedule
    return torch.from_numpy(AdamSchedule.host_table(lr, betas, n)).to(DEV)


def _st():
    return torch.cuda.current_stream().cuda_stream


def test_adam_flat_vs_oracle(lib):
    """One fused pass on 1,000,003 elements against the numpy restatement of torch's _single_tensor_adam (itself pinned against
    torch.optim.Adam in tests/test_optim_cpu.py).  gfx950 division / sqrt are correctly rounded, so 1 ulp is the budget."""
    n = 1_000_003
    rng = np.random.default_rng(0)
    p, g = rng.normal(size=n).astype(F), (rng.normal(size=n) * 10.0 ** rng.integers(-5, 1, size=n)).astype(F)
    m, v = (rng.normal(size=n) * 0.1).astype(F), (rng.random(size=n) * 0.01).astype(F)
    pe, me, ve = oadam.adam_step(p, g, m, v, 7, grad_scale=0.125)
    tp, tg, tm, tv = (torch.from_numpy(a).to(DEV) for a in (p, g, m, v))
    sched = _sched(1e-4, (0.9, 0.999), 16)
    assert lib.nr_adam_flat(tp.data_ptr(), tg.data_ptr(), tm.data_ptr(), tv.data_ptr(), n, sched.data_ptr(), 7, 0.9, 0.999, 1e-8, 0.125, 1, _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(tp.cpu().numpy(), pe, rtol=3e-7, atol=0)
    np.testing.assert_allclose(tm.cpu().numpy(), me, rtol=3e-7, atol=1e-12)
    np.testing.assert_allclose(tv.cpu().numpy(), ve, rtol=3e-7, atol=0)
    assert not tg.any()


def test_row_lazy_adam_equals_dense_bitwise_on_hardware(lib):
    """LSTUR-shaped rows (d = 900): 40 steps of random (duplicated, padded) row gradients through the lazy kernels, with reads that catch
    rows up in between, equal the dense kernel over the whole table BIT FOR BIT after flush()."""
    from news_recommendation_amd import ops
    rng = np.random.default_rng(1)
    R, d, T = 5000, 900, 40
    sched = _sched(1e-3, (0.9, 0.999), T + 2)
    p0 = torch.from_numpy(rng.normal(size=(R, d)).astype(F)).to(DEV)
    pd, md, vd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pl, ml, vl = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    last = torch.zeros(R, dtype=torch.int32, device=DEV)
    for t in range(1, T + 1):
        nb = 64
        ids = torch.from_numpy(rng.integers(0, R if t % 3 else 40, size=nb).astype(np.int64)).to(DEV)     # every third step: heavy duplication
        rows = torch.from_numpy(rng.normal(size=(nb, d)).astype(F)).to(DEV)
        read = torch.from_numpy(rng.integers(0, R, size=200).astype(np.int64)).to(DEV)
        assert lib.nr_row_adam_catchup(read.data_ptr(), read.numel(), pl.data_ptr(), ml.data_ptr(), vl.data_ptr(), last.data_ptr(), R, d,
                                       sched.data_ptr(), t - 1, 0.9, 0.999, 1e-8, _st()) == 0
        assert torch.equal(pl[read], pd[read]), f"rows not current before the forward of step {t}"
        ids_sorted, perm = ops.sort_ids(ids, R)
        # dense gradient built with the same summation order as the lazy kernel (sorted positions), pad row 0 skipped
        g = torch.zeros_like(p0)
        hs, hp = ids_sorted.cpu().numpy(), perm.cpu().numpy()
        rows_h = rows.cpu().numpy()
        gh = np.zeros((R, d), dtype=F)
        for i, pi in zip(hs, hp):
            if i > 0:
                gh[i] += rows_h[pi]
        g.copy_(torch.from_numpy(gh))
        assert lib.nr_adam_flat(pd.data_ptr(), g.data_ptr(), md.data_ptr(), vd.data_ptr(), R * d, sched.data_ptr(), t, 0.9, 0.999, 1e-8, 0.5, 1, _st()) == 0
        assert lib.nr_row_adam_step(ids_sorted.data_ptr(), perm.data_ptr(), nb, rows.data_ptr(), d, pl.data_ptr(), ml.data_ptr(), vl.data_ptr(),
                                    last.data_ptr(), R, d, sched.data_ptr(), t, 0.9, 0.999, 1e-8, 0.5, 0, _st()) == 0
    assert not torch.equal(pl, pd)
    assert lib.nr_row_adam_flush(pl.data_ptr(), ml.data_ptr(), vl.data_ptr(), last.data_ptr(), R, d, sched.data_ptr(), T, 0.9, 0.999, 1e-8, _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(pl, pd) and torch.equal(ml, md) and torch.equal(vl, vd)
    assert torch.equal(pl[0], p0[0])


@pytest.mark.parametrize('n,num_rows', [(542_720, 70_976), (542_720, 130_001), (1_899_520, 130_001), (300_001, 400_001), (53, 275), (1, 5)])
def test_sort_ids_vs_torch_stable_sort(n, num_rows):
    """Token streams of the bench shapes (Zipf ids, 45 % padding), 2- and 3-pass key widths, against torch's stable sort."""
    from news_recommendation_amd import ops, synth
    rng = np.random.default_rng(n % 1000)
    ids = synth.zipf_ids(rng, (n,), num_rows)
    ids[rng.random(n) < 0.45] = 0
    t = torch.from_numpy(ids).to(DEV)
    ids_sorted, perm = ops.sort_ids(t, num_rows)
    ref_sorted, ref_perm = torch.sort(t, stable=True)
    assert torch.equal(ids_sorted, ref_sorted) and torch.equal(perm, ref_perm)


def test_sort_ids_async_feeds_the_scatter():
    """ops.sort_ids_async (side stream) -> sorted_ids_ready: same result, usable from the current stream."""
    from news_recommendation_amd import ops
    t = torch.randint(0, 70976, (27136, 20), device=DEV)
    pack = ops.sort_ids_async(t, 70976)
    ids_sorted, perm = ops.sorted_ids_ready(pack)
    ref_sorted, ref_perm = torch.sort(t.reshape(-1), stable=True)
    assert torch.equal(ids_sorted, ref_sorted) and torch.equal(perm, ref_perm)


# ---- EngineAdam on the real drop-in models ------------------------------------------------------------------------------------------
class _Cfg:
    num_words = 3000
    num_categories = 40
    num_users = 500
    word_embedding_dim = 300
    category_embedding_dim = 100
    num_attention_heads = 15
    query_vector_dim = 200
    dropout_probability = 0.0
    num_clicked_news_a_user = 50
    num_words_title = 20
    num_words_abstract = 50
    negative_sampling_ratio = 2
    num_filters = 300
    window_size = 3
    long_short_term_method = 'ini'
    masking_probability = 0.0
    learning_rate = 1e-3
    dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}


def _nrms_batch(rng, B, V):
    cand = rng.integers(1, V, size=(B, 3, 20)).astype(np.int64)
    click = rng.integers(1, V, size=(B, 50, 20)).astype(np.int64)
    click[:, :10] = 0
    return torch.from_numpy(cand).to(DEV), torch.from_numpy(click).to(DEV)


def _lockstep(a, b, oa, ob, losses, sparse_name=None, steps=4):
    """a: torch.optim.Adam (zero_grad / backward / step, what train.py does); b: EngineAdam (flat buffers, in-place table scatter, fused
    update, optional row-sparse table).  Adam turns a gradient element that is rounding noise into a +-lr move, so two free-running
    replicas drift apart by design; here the replicas are kept in lock-step instead: every step (1) both compute their gradients from
    IDENTICAL weights -- they must agree to summation-order noise --, (2) a takes b's gradients and both optimisers step, (3) the updated
    parameters must agree to the last few ulp of the update."""
    for step in range(steps):
        oa.zero_grad()
        la, lb = losses(a), losses(b)
        la.backward()
        lb.backward()
        assert abs(la.item() - lb.item()) <= 1e-6 * max(1.0, abs(la.item())), (step, la.item(), lb.item())
        assert ob.check_views()
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            if k == sparse_name:
                ids = torch.cat([i for i, _ in ob.sparse[0].pending])
                rows = torch.cat([r for _, r in ob.sparse[0].pending])
                gb = torch.zeros_like(pa).index_add_(0, ids[ids > 0], rows[ids > 0])
            else:
                gb = pb.grad
            scale = float(pa.grad.abs().max()) + 1e-20
            assert float((pa.grad - gb).abs().max()) <= 2e-5 * scale, (step, k)
            pa.grad.copy_(gb)
        oa.step()
        ob.step()
        assert not ob.flat_g.any()
        if sparse_name is not None:
            ob.flush()
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg=f"{k} after step {step}")
            pa.data.copy_(pb.data)            # remove the ulp-level differences: the next step starts from identical weights again


def test_engine_adam_nrms_tracks_torch_adam():
    """Drop-in NRMS: in-place table gradients + fused flat Adam == returned gradients + torch.optim.Adam; the packed-weight caches follow the
    raw-kernel parameter updates (the losses of step n+1 agree); optimiser state round-trips into torch.optim.Adam."""
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    from news_recommendation_amd.optim import EngineAdam
    from news_recommendation_amd import ops
    torch.manual_seed(0)
    a = NRMS(_Cfg).to(DEV).train()
    b = NRMS(_Cfg).to(DEV).train()
    b.load_state_dict(a.state_dict())
    oa = torch.optim.Adam(a.parameters(), lr=1e-3)
    ob = EngineAdam(b, lr=1e-3)
    rng = np.random.default_rng(3)
    y = torch.zeros(6, dtype=torch.long, device=DEV)
    batches = [_nrms_batch(rng, 6, _Cfg.num_words) for _ in range(4)]
    it = {'a': iter(batches), 'b': iter(batches)}

    def losses(m):
        cand, click = next(it['a' if m is a else 'b'])
        if m is a:
            ops.invalidate_packed()           # a's parameters were overwritten through .data (no version bump)
        return torch.nn.functional.cross_entropy(m.forward_ids(cand, click), y)
    _lockstep(a, b, oa, ob, losses)
    probe = torch.optim.Adam(NRMS(_Cfg).to(DEV).parameters(), lr=1e-3)
    probe.load_state_dict(ob.state_dict())


def test_engine_adam_lstur_row_sparse_user_table():
    """LSTUR: user_embedding as a row-sparse table (no dense gradient, (id, row) pairs, lazy exact Adam) vs torch.optim.Adam's dense update of
    the same table, users repeating across and within steps (incl. the padding user 0)."""
    from news_recommendation_amd.dropin.model.LSTUR import LSTUR
    from news_recommendation_amd.optim import EngineAdam
    from news_recommendation_amd import ops
    torch.manual_seed(0)
    a = LSTUR(_Cfg).to(DEV).train()
    b = LSTUR(_Cfg).to(DEV).train()
    b.load_state_dict(a.state_dict())
    oa = torch.optim.Adam(a.parameters(), lr=1e-3)
    ob = EngineAdam(b, lr=1e-3, row_sparse=('user_embedding.weight',))
    assert b.user_embedding.weight.grad is None
    rng = np.random.default_rng(5)
    B = 6
    y = torch.zeros(B, dtype=torch.long, device=DEV)
    mk = lambda *s: torch.from_numpy(rng.integers(1, 40, size=s).astype(np.int64)).to(DEV)
    batches = []
    for _ in range(5):
        cand = {'title': torch.from_numpy(rng.integers(1, 3000, size=(B, 3, 20)).astype(np.int64)).to(DEV), 'category': mk(B, 3), 'subcategory': mk(B, 3)}
        click = {'title': torch.from_numpy(rng.integers(1, 3000, size=(B, 50, 20)).astype(np.int64)).to(DEV), 'category': mk(B, 50), 'subcategory': mk(B, 50)}
        user = torch.from_numpy(rng.integers(0, 12, size=B).astype(np.int64)).to(DEV)        # few users: repeats, idle gaps, the padding user 0
        length = torch.from_numpy(rng.integers(1, 51, size=B).astype(np.int64))
        batches.append((user, length, cand, click))
    it = {'a': iter(batches), 'b': iter(batches)}

    def losses(m):
        user, length, cand, click = next(it['a' if m is a else 'b'])
        if m is a:
            ops.invalidate_packed()
        return torch.nn.functional.cross_entropy(m.forward_ids(user, length.clone(), cand, click), y)
    _lockstep(a, b, oa, ob, losses, sparse_name='user_embedding.weight', steps=5)
    ua, ub = a.state_dict()['user_embedding.weight'], b.state_dict()['user_embedding.weight']
    assert torch.equal(ub[12:], ua[12:])                                   # users never drawn: untouched in both


# ---- MIND-large-shaped parity (BASELINE configs[3], configs[4]) ---------------------------------------------------------------------------
def _rel(got, ref, floor=0.0):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + floor + 1e-30)


def test_nrms_mind_large_vocabulary_vs_oracle():
    """NRMS with the MIND-large knob set of bench.py --shape large (vocabulary 1 + 130,000): logits and every gradient vs the CPU fp32
    oracle at B = 8, token ids spread over the WHOLE vocabulary (17-bit keys: two 9-bit passes of the id sort, top row included)."""
    from tests.test_model_gpu import build, as_lists, mind_batch, grad_floor
    from news_recommendation_amd import synth
    V = synth.SHAPES['large']['num_words']
    rng = np.random.default_rng(31)
    params = onp.random_nrms_params(rng, V, 300, 200, np.float32, emb_std=0.4)
    cand, click = mind_batch(rng, 8, V=V)
    cand[cand > 0] = rng.integers(1, V, size=int((cand > 0).sum()))       # uniform over the vocabulary, incl. the top rows
    click[click > 0] = rng.integers(1, V, size=int((click > 0).sum()))
    cand[0, 0, 0], click[0, -1, 0] = V - 1, V - 1
    ref = OracleNRMS(V, 300, 15, 200, 0.2)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ref.eval()
    lr = ref(as_lists(cand), as_lists(click))
    torch.nn.CrossEntropyLoss()(lr, torch.zeros(8, dtype=torch.long)).backward()
    m = build(V, 300, 15, 200, 50, 20, params).eval()
    lg = m(as_lists(cand), as_lists(click))
    torch.nn.CrossEntropyLoss()(lg, torch.zeros(8, dtype=torch.long, device=DEV)).backward()
    assert _rel(lg.detach().cpu().numpy(), lr.detach().numpy()) < 1.5e-2          # bf16 operands, fp32 accumulation
    gref = dict(ref.named_parameters())
    fl = grad_floor({k: v.grad.numpy() for k, v in gref.items()})
    for k, p in m.named_parameters():
        e = _rel(p.grad.cpu().numpy(), gref[k].grad.numpy(), fl)
        assert e < 5e-2, (k, e)
    gw = m.news_encoder.word_embedding.weight.grad
    assert torch.all(gw[0] == 0) and gw[V - 1].abs().sum() > 0


def test_lstur_mind_large_user_table_vs_oracle():
    """LSTUR with the 711,223-row user table of MIND-large (2.56 GB fp32): forward logits vs the CPU oracle with user ids from the whole
    range, then one EngineAdam step (row-sparse table) vs one torch.optim.Adam step of the oracle on the touched rows."""
    from news_recommendation_amd.dropin.model.LSTUR import LSTUR
    from news_recommendation_amd.optim import EngineAdam
    from news_recommendation_amd import synth
    from oracle.lstur_torch import OracleLSTUR

    class Cfg(_Cfg):
        num_users = synth.SHAPES['large']['num_users']
        num_words = 5000
    torch.manual_seed(2)
    m = LSTUR(Cfg).to(DEV).eval()
    with torch.no_grad():
        m.user_embedding.weight.normal_(0, 0.3)
        m.user_embedding.weight[0].zero_()
    ref = OracleLSTUR(Cfg.num_words, 300, Cfg.num_categories, Cfg.num_users, 300, 3, 200, 0.0, 0.0, 'ini')
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    ref.eval()
    rng = np.random.default_rng(9)
    B = 6
    mk = lambda *s: torch.from_numpy(rng.integers(1, 40, size=s).astype(np.int64))
    cand = {'title': torch.from_numpy(rng.integers(1, 5000, size=(B, 3, 20)).astype(np.int64)), 'category': mk(B, 3), 'subcategory': mk(B, 3)}
    click = {'title': torch.from_numpy(rng.integers(1, 5000, size=(B, 50, 20)).astype(np.int64)), 'category': mk(B, 50), 'subcategory': mk(B, 50)}
    user = torch.tensor([Cfg.num_users - 1, 1, 355_611, 700_000, 0, 2 ** 19 + 3], dtype=torch.int64)
    length = torch.from_numpy(rng.integers(1, 51, size=B).astype(np.int64))
    cl = [{k: cand[k][:, j] for k in cand} for j in range(3)]
    hl = [{k: click[k][:, j] for k in click} for j in range(50)]
    lr = ref(user, length.clone(), cl, hl)
    lg = m(user, length.clone(), cl, hl)
    assert _rel(lg.detach().cpu().numpy(), lr.detach().numpy()) < 2e-2
    # one optimiser step on both
    m.train()                                   # dropout / masking probabilities are 0 in this config: train mode only enables gradients
    ref.train()
    opt = EngineAdam(m, lr=1e-3, row_sparse=('user_embedding.weight',))
    oref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    y = torch.zeros(B, dtype=torch.long)
    torch.nn.functional.cross_entropy(ref(user, length.clone(), cl, hl), y).backward()
    oref.step()
    torch.nn.functional.cross_entropy(m(user, length.clone(), cl, hl), y.to(DEV)).backward()
    opt.step()
    opt.flush()
    rows = user[user > 0]
    got = m.user_embedding.weight.detach()[rows.to(DEV)].cpu().numpy()
    want = ref.user_embedding.weight.detach()[rows].numpy()
    # first Adam step moves every touched element by lr * sign(g): agreement means the row gradients have the oracle's signs
    assert np.mean(np.abs(got - want) < 1e-4) > 0.97
    untouched = torch.tensor([5, 123_456, 711_000])
    assert torch.equal(m.user_embedding.weight.detach()[untouched.to(DEV)].cpu(), ref.user_embedding.weight.detach()[untouched])
