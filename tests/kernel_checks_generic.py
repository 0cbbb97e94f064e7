"""Backend-agnostic parity checks of the general-geometry kernels (csrc/k_generic.h) and of the GEMM forms they lean on (overlapping-row
nr_gemm_nt_rows, nr_gemm_tn with any number of taps) against float64 torch / numpy restatements of the reference's formulas:
ScaledDotProductAttention (multihead_self.py:15-23), AdditiveAttention (additive.py:27-53), Conv2d(1, F, (w, D)) + relu + dropout
(NAML news_encoder.py:21-32).  Run on the CPU wave emulator (tests/test_generic_emu.py) and on MI355X (tests/test_generic_gpu.py)."""
import numpy as np
import torch

from tests.backends import bf16_to_f32, f32_to_bf16, bf16_round
from tests import kernel_checks as kc

F32 = np.float32


def pad32(n):
    return (n + 1 + 31) // 32 * 32


def check_attn(be, n_seq=5, S=13, H=4, dk=10, seed=0, with_len=True):
    rng = np.random.default_rng(seed)
    D = H * dk
    ld = 3 * D + 4
    qkv = rng.normal(0, 0.8, size=(n_seq * S, ld)).astype(F32)
    dctx = rng.normal(0, 1.0, size=(n_seq * S, D)).astype(F32)
    kl = rng.integers(1, S + 1, size=n_seq).astype(np.int32) if with_len else None
    if kl is not None:
        kl[0] = S
    h_qkv, h_len = be.dev(qkv), (be.dev(kl) if kl is not None else None)
    ctx = be.poison((n_seq * S, D), F32)
    kc.ck(be, be.lib.nr_g_attn_fwd(be.ptr(h_qkv), ld, be.ptr(ctx), be.ptr(h_len), n_seq, S, H, dk, be.stream))
    dq = be.poison((n_seq * S, ld), F32)
    kc.ck(be, be.lib.nr_g_attn_bwd(be.ptr(h_qkv), ld, be.ptr(be.dev(dctx)), be.ptr(dq), be.ptr(h_len), n_seq, S, H, dk, be.stream))
    be.sync()
    t = torch.from_numpy(qkv[:, :3 * D].astype(np.float64)).requires_grad_(True)
    v = t.view(n_seq, S, 3, H, dk)
    q, k, vv = (v[:, :, i].transpose(1, 2) for i in range(3))                       # [n_seq, H, S, dk]
    e = torch.exp(q @ k.transpose(-1, -2) / np.sqrt(dk))                            # multihead_self.py:16-17
    if kl is not None:
        mask = (torch.arange(S)[None, :] < torch.from_numpy(kl.astype(np.int64))[:, None]).to(torch.float64)     # :60-70, applied to exp(scores) (:18-19)
        e = e * mask[:, None, None, :]
    a = e / (e.sum(-1, keepdim=True) + 1e-8)                                         # :20
    ref = (a @ vv).transpose(1, 2).reshape(n_seq * S, D)
    ref.backward(torch.from_numpy(dctx.astype(np.float64)))
    got = be.np(ctx)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    gd = be.np(dq)
    np.testing.assert_allclose(gd[:, :3 * D], t.grad.numpy(), rtol=2e-4, atol=2e-5)
    assert np.isnan(gd[:, 3 * D:]).all(), 'columns past 3 D of dqkv must not be written'
    assert be.lib.nr_g_attn_fwd(be.ptr(h_qkv), ld, be.ptr(ctx), None, n_seq, 65, H, dk, be.stream) != 0          # S > 64
    assert be.lib.nr_g_attn_fwd(be.ptr(h_qkv), ld, be.ptr(ctx), None, n_seq, S, H, 33, be.stream) != 0           # d_k > 32


def check_additive(be, n_seq=6, S=11, D=52, Q=24, valid=None, seed=1):
    rng = np.random.default_rng(seed)
    valid = S if valid is None else valid
    x = rng.normal(0, 0.7, size=(n_seq * S, D)).astype(F32)
    proj = rng.normal(0, 0.9, size=(n_seq * S, Q)).astype(F32)
    qv = rng.uniform(-0.5, 0.5, size=Q).astype(F32)
    g = rng.normal(0, 1.0, size=(n_seq, D)).astype(F32)
    hx, hp, hq = be.dev(x), be.dev(proj), be.dev(qv)
    out, aw = be.poison((n_seq, D), F32), be.poison((n_seq, S), F32)
    kc.ck(be, be.lib.nr_g_additive_fwd(be.ptr(hx), D, D, be.ptr(hp), Q, Q, be.ptr(hq), be.ptr(out), D, be.ptr(aw), n_seq, S, valid, be.stream))
    be.sync()
    tx = torch.from_numpy(x.astype(np.float64)).view(n_seq, S, D).requires_grad_(True)
    tp = torch.from_numpy(proj.astype(np.float64)).view(n_seq, S, Q).requires_grad_(True)
    tq = torch.from_numpy(qv.astype(np.float64)).requires_grad_(True)
    sc = torch.tanh(tp) @ tq                                                         # additive.py:35-38
    if valid < S:
        sc = sc.masked_fill(torch.arange(S)[None, :] >= valid, float('-inf'))
    w = torch.softmax(sc, dim=1)
    ref = torch.bmm(w.unsqueeze(1), tx).squeeze(1)                                   # :51-52
    np.testing.assert_allclose(be.np(out), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(be.np(aw), w.detach().numpy(), rtol=2e-5, atol=1e-7)
    if valid < S:
        return
    ref.backward(torch.from_numpy(g.astype(np.float64)))
    dpre, dqp = be.poison((n_seq * S, Q), F32), be.poison((n_seq, Q), F32)
    kc.ck(be, be.lib.nr_g_additive_bwd(be.ptr(hx), D, D, be.ptr(hp), Q, Q, be.ptr(hq), be.ptr(aw), be.ptr(be.dev(g)), D, be.ptr(dpre), Q, be.ptr(dqp),
                                       n_seq, S, be.stream))
    dx = be.poison((n_seq * S, D), F32)
    kc.ck(be, be.lib.nr_g_rows_axpy(be.ptr(dx), D, be.ptr(aw), be.ptr(be.dev(g)), D, S, D, n_seq * S, 0, be.stream))
    be.sync()
    np.testing.assert_allclose(be.np(dpre), tp.grad.numpy().reshape(n_seq * S, Q), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(be.np(dqp).sum(0), tq.grad.numpy(), rtol=2e-4, atol=2e-5)
    # the direct term of the input gradient (the path through proj is the linear layer's backward: a GEMM)
    np.testing.assert_allclose(be.np(dx), (w.detach().numpy()[:, :, None] * g[:, None, :]).reshape(n_seq * S, D), rtol=2e-5, atol=1e-7)


def check_dropout(be, n=4 * 333, p=0.3, seed=0x1234567, site=2, elem0=64):
    """nr_g_dropout applies exactly the mask nr_dropout_mask exports for the same (seed, site, element counter)."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=n).astype(F32)
    y = be.poison((n,), F32)
    kc.ck(be, be.lib.nr_g_dropout(be.ptr(be.dev(x)), be.ptr(y), n, elem0, p, seed, site, be.stream))
    be.sync()
    keep = kc.export_mask(be, elem0 + n, p, seed, site)[elem0:]
    np.testing.assert_allclose(be.np(y), x * keep / F32(1.0 - p), rtol=1e-6, atol=0)
    assert 0.5 < keep.mean() / (1 - p) < 1.5
    assert be.lib.nr_g_dropout(be.ptr(y), be.ptr(y), 6, 0, p, seed, site, be.stream) != 0                  # not a multiple of 4


def check_conv(be, n_seq=5, S=9, D=20, F=24, w=3, p=0.25, seed=4):
    """Conv2d(1, F, (w, D), padding = ((w - 1) / 2, 0)) + relu + dropout, its tap gradients and its data gradient through the seqpad GEMM forms."""
    rng = np.random.default_rng(seed)
    pad = (w - 1) // 2
    Dp, Fp = pad32(D), pad32(F)
    M = n_seq * (S + pad)
    x = rng.normal(0, 0.6, size=(n_seq * S, D)).astype(F32)
    W = rng.normal(0, 0.3, size=(F, w, D)).astype(F32)
    b = rng.normal(0, 0.2, size=F).astype(F32)
    dseed = 0xABCDEF12345
    # operands as the host packs them (ops_generic._pack_conv / _pack_conv_dgrad)
    Wc = np.zeros((F, w, Dp), dtype=F32); Wc[:, :, :D] = W; Wc[:, pad, D] = b
    Wd = np.zeros((D, w, Fp), dtype=F32); Wd[:, :, :F] = W[:, ::-1, :].transpose(2, 1, 0)
    xpad = be.empty((M + 2 * pad + 1, Dp), np.uint16)
    kc.ck(be, be.lib.nr_g_rows_to_seqpad(be.ptr(be.dev(x)), D, D, be.ptr(xpad), Dp, S, pad, n_seq * S, 1, be.stream))
    yv = be.poison((M, F), F32)
    kc.ck(be, be.lib.nr_gemm_nt_rows(be.ptr(xpad), Dp, be.ptr(be.dev(f32_to_bf16(Wc.reshape(F, w * Dp)))), w * Dp, be.ptr(yv), F, M, F, w * Dp, be.stream))
    act = be.poison((n_seq * S, F), F32)
    kc.ck(be, be.lib.nr_g_relu_drop(be.ptr(yv), F, be.ptr(act), F, S, pad, n_seq * S, p, dseed, 8 * F, be.stream))
    be.sync()
    xp = be.np(xpad)
    ref_pad = np.zeros((M + 2 * pad + 1, Dp), dtype=F32)
    for q in range(n_seq):
        ref_pad[pad + q * (S + pad):pad + q * (S + pad) + S, :D] = bf16_round(x[q * S:(q + 1) * S])
        ref_pad[pad + q * (S + pad):pad + q * (S + pad) + S, D] = 1.0
    assert np.array_equal(bf16_to_f32(xp), ref_pad), 'seqpad layout'
    xq = torch.from_numpy(bf16_round(x).astype(np.float64)).view(n_seq, 1, S, D).requires_grad_(True)
    Wq = torch.from_numpy(bf16_round(W).astype(np.float64)).view(F, 1, w, D).requires_grad_(True)
    bq = torch.from_numpy(bf16_round(b).astype(np.float64)).requires_grad_(True)
    y = torch.nn.functional.conv2d(xq, Wq, bq, padding=(pad, 0)).squeeze(3).transpose(1, 2).reshape(n_seq * S, F)      # NAML news_encoder.py:27-28
    keep = kc.export_mask(be, (8 + n_seq * S) * F, p, dseed, 2)[8 * F:].reshape(n_seq * S, F).astype(np.float64)
    ref_act = torch.relu(y) * torch.from_numpy(keep) / (1.0 - p)
    np.testing.assert_allclose(be.np(act), ref_act.detach().numpy(), rtol=0, atol=3e-5 * np.sqrt(w * D))
    # ---- backward: dact -> dy (seqpad, bf16) -> tap gradients (one GEMM, taps = w) and data gradient (overlapping-row GEMM, flipped taps) ----
    dact = rng.normal(0, 1.0, size=(n_seq * S, F)).astype(F32)
    dypad = be.empty((M + 2 * pad + 1, Fp), np.uint16)
    kc.ck(be, be.lib.nr_g_relu_drop_bwd(be.ptr(be.dev(dact)), be.ptr(act), be.ptr(dypad), F, Fp, S, pad, n_seq * S, p, be.stream))
    be.sync()
    dy_tok = np.where(be.np(act) != 0, dact * (F32(1.0) / (F32(1.0) - F32(p))), 0).astype(F32)        # (the kernel multiplies by the reciprocal)
    dyp = bf16_to_f32(be.np(dypad))
    for q in range(n_seq):
        assert np.array_equal(dyp[pad + q * (S + pad):pad + q * (S + pad) + S, :F], bf16_round(dy_tok[q * S:(q + 1) * S]))
    assert not dyp[:, F:].any()
    y.backward(torch.from_numpy(bf16_round(dy_tok).astype(np.float64)))
    N = w * Dp
    P = be.lib.nr_gemm_tn_parts(F, N, M)
    parts = be.poison((P, F, N), F32)
    zeros = be.empty((64,), np.uint16)
    g_ptr = be.ptr(dypad) + pad * Fp * 2                                             # G starts at the first token row
    kc.ck(be, be.lib.nr_gemm_tn(g_ptr, Fp, F, be.ptr(xpad), Dp, Dp, w, be.ptr(zeros), be.ptr(parts), N, M, P, be.stream))
    dxv = be.poison((M, D), F32)
    kc.ck(be, be.lib.nr_gemm_nt_rows(be.ptr(dypad), Fp, be.ptr(be.dev(f32_to_bf16(Wd.reshape(D, w * Fp)))), w * Fp, be.ptr(dxv), D, M, D, w * Fp, be.stream))
    dx = be.poison((n_seq * S, D), F32)
    kc.ck(be, be.lib.nr_g_unpad_rows(be.ptr(dxv), D, be.ptr(dx), D, S, pad, n_seq * S, be.stream))
    be.sync()
    ext = be.np(parts).astype(np.float64).sum(0).reshape(F, w, Dp)
    tol = 3e-5 * np.sqrt(n_seq * S)
    np.testing.assert_allclose(ext[:, :, :D], Wq.grad.numpy()[:, 0], rtol=0, atol=tol * 4)
    np.testing.assert_allclose(ext[:, pad, D], bq.grad.numpy(), rtol=0, atol=tol * 4)
    np.testing.assert_allclose(be.np(dx), xq.grad.numpy().reshape(n_seq * S, D), rtol=0, atol=3e-5 * np.sqrt(w * F) * 4)


def check_relu(be, n=1000):
    rng = np.random.default_rng(5)
    x, gate = rng.normal(size=n).astype(F32), rng.normal(size=n).astype(F32)
    y = be.poison((n,), F32)
    kc.ck(be, be.lib.nr_g_relu(be.ptr(be.dev(x)), None, be.ptr(y), n, 1.0, be.stream))
    be.sync()
    assert np.array_equal(be.np(y), np.maximum(x, 0))
    kc.ck(be, be.lib.nr_g_relu(be.ptr(be.dev(x)), be.ptr(be.dev(gate)), be.ptr(y), n, 1.25, be.stream))
    be.sync()
    assert np.array_equal(be.np(y), np.where(gate > 0, x * F32(1.25), 0).astype(F32))


def check_split_linear(be, n=37, D=100, N=24, seed=6):
    """[hi | hi | lo] rows x [Wh | Wl | Wh] weights through ONE bf16 GEMM = x W^T + b to ~2^-16 relative."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.7, size=(n, D)).astype(F32)
    W = rng.normal(0, 0.3, size=(N, D)).astype(F32)
    b = rng.normal(0, 0.2, size=N).astype(F32)
    Dp = pad32(D)
    x3 = be.poison((n, 3 * Dp), np.uint16)
    kc.ck(be, be.lib.nr_g_rows_split_bf16(be.ptr(be.dev(x)), D, D, be.ptr(x3), Dp, n, be.stream))
    full = np.zeros((N, Dp), dtype=F32); full[:, :D] = W; full[:, D] = b
    hi = bf16_round(full); lo = bf16_round(full - hi)
    W3 = f32_to_bf16(np.concatenate([hi, lo, hi], axis=1))
    y = be.poison((n, N), F32)
    kc.ck(be, be.lib.nr_gemm_nt(be.ptr(x3), 3 * Dp, be.ptr(be.dev(W3)), 3 * Dp, be.ptr(y), N, n, N, 3 * Dp, be.stream))
    be.sync()
    ref = x.astype(np.float64) @ W.astype(np.float64).T + b
    np.testing.assert_allclose(be.np(y), ref, rtol=0, atol=6e-5 * np.abs(ref).max())
    plain = bf16_round(x).astype(np.float64) @ bf16_round(W).astype(np.float64).T + bf16_round(b)
    assert np.abs(be.np(y) - ref).max() < 0.05 * np.abs(plain - ref).max()          # ... i.e. far below the plain bf16 product's error
