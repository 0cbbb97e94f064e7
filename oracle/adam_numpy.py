"""TEST INFRASTRUCTURE (oracle): CPU restatement of the optimiser step of the reference's training loop.

The reference builds ``torch.optim.Adam(model.parameters(), lr=config.learning_rate)`` (src/train.py:127-128) and calls
``optimizer.zero_grad(); loss.backward(); optimizer.step()`` (src/train.py:229-233).  The arithmetic lives in a third-party
dependency, PyTorch (``requirements.txt:1`` pins no version; installed here: torch 2.10.0): ``torch/optim/adam.py``
``_single_tensor_adam`` with the defaults betas = (0.9, 0.999), eps = 1e-8, weight_decay = 0, amsgrad = False, maximize = False:

    exp_avg.lerp_(grad, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bias_correction1 = 1 - beta1 ** step ; bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1 ; bias_correction2_sqrt = bias_correction2 ** 0.5
    denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-step_size)

``adam_step`` restates it in numpy float32, one rounding per operation.  Pinned by tests/test_optim_cpu.py against torch.optim.Adam
itself (which runs here on CPU).  Only tests/ import this module; the product's update runs in csrc/k_optim.h.
"""
import numpy as np

F = np.float32


def scalars(lr, betas, step):
    """(step_size, bias_correction2_sqrt) of step `step` (1-based), in double then float32 like the engine's schedule table."""
    bc1 = 1.0 - betas[0] ** step
    bc2 = 1.0 - betas[1] ** step
    return F(lr / bc1), F(bc2 ** 0.5)


def adam_step(p, g, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """One Adam step on float32 arrays (returns new p, m, v; inputs untouched)."""
    p, g, m, v = (np.asarray(a, dtype=F) for a in (p, g, m, v))
    g = g * F(grad_scale)
    om_b1, b2, om_b2 = F(1.0 - betas[0]), F(betas[1]), F(1.0 - betas[1])
    step_size, bc2_sqrt = scalars(lr, betas, step)
    m = m + (g - m) * om_b1
    v = v * b2 + (om_b2 * g) * g
    denom = np.sqrt(v) / bc2_sqrt + F(eps)
    p = p - step_size * (m / denom)
    return p.astype(F), m.astype(F), v.astype(F)


def adam_run(p0, grads, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """Dense Adam over a list of gradients (None = zero gradient, what an untouched row of a table sees)."""
    p = np.asarray(p0, dtype=F).copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t, g in enumerate(grads, 1):
        p, m, v = adam_step(p, np.zeros_like(p) if g is None else g, m, v, t, lr, betas, eps, grad_scale)
    return p, m, v
