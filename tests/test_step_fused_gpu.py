"""The training step without framework kernels (round 6): model.forward_stacked(..., loss=True) -- stacked batch layout, fused scorer +
cross entropy (nr_score_ce_*), gradient destinations of split_rows, one accumulate launch for the small weight gradients (nr_accum_many) --
against the reference-shaped form of the same step: forward_ids + torch.nn.CrossEntropyLoss (src/train.py:202-207), whose parity with the
oracle the model tests hold.  Same parameters, same batch, dropout off (the two forms draw the same masks only by accident of call order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _workload(name, B=48, vocab=3000):
    import bench
    cfg = bench.make_cfg(name, 'small', vocab)
    cfg.dropout_probability = 0.0
    if name == 'LSTUR':
        cfg.masking_probability = 0.0
    wl = bench.Workload(name, cfg)
    return wl, wl.batches(0, 1, B, torch.device(DEV))[0]


def _grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _criterion_step(wl, model, b, target):
    model.zero_grad(set_to_none=True)
    loss = torch.nn.CrossEntropyLoss()(wl.forward(model, b), target)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), _grads(model)


def _stacked_loss(wl, model, b, target=None):
    if wl.name == 'NRMS':
        return model.forward_stacked(b['ids']['title'], b['B'], b['C'], loss=True, target=target)
    if wl.name == 'NAML':
        return model.forward_stacked(b['ids'], b['B'], b['C'], loss=True, target=target)
    return model.forward_stacked(b['user'], b['length'].clone(), b['ids'], b['B'], b['C'], loss=True, target=target)


def _close(got, ref, what, rel=1e-2, floor=0.0):
    """rel: fraction of the tensor's largest magnitude; floor: absolute allowance (gradients that are sums with heavy cancellation -- the key
    bias of the attention, whose exact gradient is zero -- are rounding noise at the scale of the OTHER gradients).  Default rel = a couple of bf16
    steps: the two forms differ in the last bits of the logit gradient (expf / fixed-order mean vs torch's), which re-rounds a few of the bf16
    intermediates of the backward (dqkv, dX, dpre) by one step -- measured on MI355X: 2e-4 .. 7e-3 of the maximum."""
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= rel * max(scale, 1e-6) + floor + 1e-9, f'{what}: max |diff| {err:.3e} at scale {scale:.3e} (floor {floor:.1e})'


def _floor(ref_grads):
    """1e-3 of the median over the parameters of their largest gradient magnitude."""
    return 1e-3 * float(np.median([float(g.abs().max()) for g in ref_grads.values()]))


@pytest.mark.parametrize('name', ['NRMS', 'NAML', 'LSTUR'])
def test_fused_step_equals_criterion_step_plain_autograd(name):
    """Plain autograd (gradients returned to AccumulateGrad): loss and every parameter gradient of the two forms agree to fp32 rounding --
    they run the same encoder kernels; only the scorer / loss tail and the routing of the input gradients differ."""
    wl, b = _workload(name)
    torch.manual_seed(3)
    model = wl.make_model().to(DEV).train()
    target = torch.zeros(b['B'], dtype=torch.long, device=DEV)
    l_ref, g_ref = _criterion_step(wl, model, b, target)
    model.zero_grad(set_to_none=True)
    loss = _stacked_loss(wl, model, b)
    loss.backward()
    torch.cuda.synchronize()
    g = _grads(model)
    assert abs(float(loss.detach()) - l_ref) <= 2e-6 * max(1.0, abs(l_ref))
    assert set(g) == set(g_ref)
    for n in g_ref:
        _close(g[n], g_ref[n], f'{name} d {n}', floor=_floor(g_ref))


@pytest.mark.parametrize('name', ['NRMS', 'NAML', 'LSTUR'])
def test_fused_step_with_persistent_gradient_buffers(name):
    """EngineAdam's flat buffers (ops.inplace_grads): the backward functions queue their small weight gradients and return None; the queue is
    flushed by ONE nr_accum_many launch when autograd's engine finishes the pass -- p.grad holds the same values as plain autograd's, twice in a
    row (the buffers accumulate), and nothing is left in the queue."""
    from news_recommendation_amd import ops
    wl, b = _workload(name)
    torch.manual_seed(4)
    model = wl.make_model().to(DEV).train()
    target = torch.zeros(b['B'], dtype=torch.long, device=DEV)
    _, g_ref = _criterion_step(wl, model, b, target)
    model.zero_grad(set_to_none=True)
    opt = wl.make_optimizer(model)          # takes over .grad: views of one flat zeroed buffer
    sparse = {n for n, p in model.named_parameters() if p.grad is None}          # LSTUR's user table: row-sparse sink, no dense gradient
    for rep in (1, 2):
        loss = _stacked_loss(wl, model, b)
        loss.backward()
        torch.cuda.synchronize()
        assert not ops._gq and not ops._gq_armed
        for n, p in model.named_parameters():
            if n in sparse or n not in g_ref:
                continue
            _close(p.grad, rep * g_ref[n], f'{name} (buffers, pass {rep}) d {n}', floor=rep * _floor(g_ref))
    opt.discard_grads()


def test_fused_loss_with_targets_and_upstream_scale():
    """dot_score_ce against log_softmax + nll_loss of torch on the same vectors: arbitrary target classes, a scaled loss (gradient != 1 arrives at
    the scalar), vectors that are NOT parts of a split_rows (own gradient buffers)."""
    from news_recommendation_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, D = 77, 5, 300
    cand = (torch.randn(B, C, D, generator=g) * 0.2).to(DEV).requires_grad_()
    user = (torch.randn(B, D, generator=g) * 0.2).to(DEV).requires_grad_()
    target = torch.randint(0, C, (B,), generator=g).to(DEV)
    loss = ops.dot_score_ce(cand, user, target)
    (loss * 0.37).backward()
    c2, u2 = cand.detach().clone().requires_grad_(), user.detach().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(torch.einsum('bcd,bd->bc', c2.double(), u2.double()), target)
    (ref * 0.37).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6
    _close(cand.grad, c2.grad.float(), 'd cand', rel=2e-5)
    _close(user.grad, u2.grad.float(), 'd user', rel=2e-5)
    with pytest.raises(NotImplementedError):
        ops.dot_score_ce(torch.zeros(2, 65, 300, device=DEV), torch.zeros(2, 300, device=DEV))


def test_split_rows_falls_back_to_one_concatenation():
    """A consumer that does not ask for a gradient destination (plain torch ops downstream of one part): the backward concatenates as before."""
    from news_recommendation_amd import ops
    x = torch.randn(10, 300, device=DEV, requires_grad=True)
    a, b = ops.split_rows(x, 4)
    user = torch.randn(2, 300, device=DEV, requires_grad=True)
    loss = ops.dot_score_ce(a.view(2, 2, 300), user) + (b * b).sum()
    loss.backward()
    torch.cuda.synchronize()
    ref = x.detach().clone().requires_grad_()
    l2 = torch.nn.functional.cross_entropy(torch.einsum('bcd,bd->bc', ref[:4].view(2, 2, 300), user.detach()),
                                           torch.zeros(2, dtype=torch.long, device=DEV)) + (ref[4:] * ref[4:]).sum()
    l2.backward()
    _close(x.grad, ref.grad, 'd x through split_rows', rel=2e-5)
