"""Checks of the flat pooling backward (csrc/k_pool3.h, nr_additive_bwd_flat) against the numpy restatement of AdditiveAttention's autograd
(src/model/general/attention/additive.py:27-53) -- shared by the emulator tests (CPU) and the GPU tests, like tests/kernel_checks.py."""
import numpy as np

from oracle import nrms_numpy as onp
from tests.backends import bf16_round, bf16_to_f32, f32_to_bf16
from tests.kernel_checks import NR_D, NR_KP, NR_QP, ck, close_bf16, make_params, pack_additive
from tests.kernel_checks_conv import seqpad_rows, to_seqpad

A_ = 'news_encoder.additive_attention.'


def _setup(be, S, n_seq, valid, seed, relu=False, y_stride=NR_D):
    params = make_params(14)
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    if relu:                                        # activations of a conv text encoder: exact zeros where relu / dropout cut
        x = np.maximum(x, 0)
        x[rng.random(size=x.shape) < 0.2] = 0
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = x
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    hctx = be.dev(ctx_u)
    out = be.poison((n_seq, y_stride), np.float32)
    aw = be.poison((n_seq, S), np.float32)
    if S in (4, 20, 50):
        ck(be, be.lib.nr_additive_fwd_v(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), y_stride, None, 0, be.ptr(aw), n_seq, S,
                                        S if valid is None else valid, be.stream))
    else:                                           # a length the forward kernels are not instantiated for: the oracle's forward supplies y and w
        assert valid is None
        xs = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)
        y64, w64, _ = onp.additive(xs, bf16_round(params[A_ + 'linear.weight']).astype(np.float64), params[A_ + 'linear.bias'].astype(np.float64),
                                   params[A_ + 'attention_query_vector'].astype(np.float64))
        yh = np.zeros((n_seq, y_stride), dtype=np.float32); yh[:, :NR_D] = y64
        out, aw = be.dev(yh), be.dev(w64.astype(np.float32))
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    return params, ctx_u, hctx, (Wap, bap, qvp), out, aw, go


def _reference(params, ctx_u, go, S, n_seq, V_):
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)[:, :V_]
    W = bf16_round(params[A_ + 'linear.weight']).astype(np.float64)
    b = params[A_ + 'linear.bias'].astype(np.float64)
    qv = params[A_ + 'attention_query_vector'].astype(np.float64)
    _, w, temp = onp.additive(x, W, b, qv)
    g = go.astype(np.float64)
    dw = np.einsum('bd,bsd->bs', g, x)
    ds = w * (dw - (w * dw).sum(1, keepdims=True))
    dpre_ref = ds[:, :, None] * qv[None, None, :] * (1 - temp * temp)
    dq_ref = np.einsum('bs,bsq->q', ds, temp)
    # the restatement agrees with the oracle's own backward
    dx_ref, dW_ref, db_ref, dqv_ref = onp.additive_bwd(g, x, w, temp, W, qv)
    np.testing.assert_allclose(dq_ref, dqv_ref, rtol=1e-9)
    np.testing.assert_allclose(w[:, :, None] * g[:, None, :] + dpre_ref @ W, dx_ref, rtol=1e-9, atol=1e-12)
    return W, w, dpre_ref, dq_ref


def check_flat(be, S=20, n_seq=6, valid=None, seed=15, y_stride=NR_D, with_dctx=True, g_stride=NR_D):
    """dpre, dq and the fused dctx = dpre @ Wa of nr_additive_bwd_flat: sequences that straddle the 48-row groups of the waves, padded tails
    (valid < S: exact zeros there), a strided y, and -- with_dctx=False -- the form that stops at dpre / dq."""
    params, ctx_u, hctx, (Wap, bap, qvp), out, aw, go = _setup(be, S, n_seq, valid, seed, y_stride=y_stride)
    V_ = S if valid is None else valid
    nwg = be.lib.nr_additive_bwd_flat_grid(n_seq * S)
    assert nwg >= 1
    dpre = be.poison((n_seq * S, NR_QP), np.uint16)
    dqp = be.poison((nwg, NR_QP), np.float32)
    tot = be.poison((n_seq,), np.float32)
    dctx = be.poison((n_seq * S, NR_KP), np.uint16) if with_dctx else None
    hgo, g_off = _wide_rows(be, go, g_stride)
    ck(be, be.lib.nr_additive_bwd_flat_gs(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(hgo) + g_off, g_stride, be.ptr(out),
                                          y_stride, be.ptr(tot), be.ptr(dpre), be.ptr(dqp), be.ptr(dctx) if with_dctx else None, None, 0.0, n_seq, S,
                                          200, be.stream))
    be.sync()
    W, w, dpre_ref, dq_ref = _reference(params, ctx_u, go, S, n_seq, V_)
    # the per-sequence scalar: g . y == sum_s w[s] (g . x[s])
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)[:, :V_]
    tot_ref = (w * np.einsum('bd,bsd->bs', go.astype(np.float64), x)).sum(1)
    np.testing.assert_allclose(be.np(tot), tot_ref, rtol=2e-4, atol=2e-4)
    got = bf16_to_f32(be.np(dpre)).reshape(n_seq, S, NR_QP)
    assert not got[:, V_:].any(), 'padded positions must get exact zeros'
    got = got[:, :V_].reshape(-1, NR_QP)
    close_bf16(got[:, :200], dpre_ref.reshape(-1, 200), f'flat pooling bwd dpre S={S}', rel=2.0 ** -7, floor=2e-3)
    assert not got[:, 200:].any()
    dq = be.np(dqp).astype(np.float64).sum(0)
    np.testing.assert_allclose(dq[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())
    assert not dq[200:].any()
    if with_dctx:
        dref = bf16_to_f32(be.np(dpre)).astype(np.float64)[:, :200] @ W                # bit-level operands as the kernel sees them
        dc = bf16_to_f32(be.np(dctx)).reshape(n_seq, S, NR_KP)
        assert not dc[:, V_:, :NR_D].any()
        close_bf16(dc.reshape(-1, NR_KP)[:, :NR_D], dref, f'flat pooling bwd fused dctx S={S}', rel=2.0 ** -7, floor=1e-3)


def _wide_rows(be, go, g_stride):
    """The sequence gradients as the LAST NR_D columns of rows of g_stride floats (everything else NaN: the kernel must not read it);
    returns (device handle, byte offset of row 0's block).  g_stride == NR_D: the plain contiguous matrix."""
    if g_stride == NR_D:
        return be.dev(go), 0
    wide = np.full((go.shape[0], g_stride), np.nan, dtype=np.float32)
    wide[:, g_stride - NR_D:] = go
    return be.dev(wide), (g_stride - NR_D) * 4


def check_flat_act(be, S=20, n_seq=7, p_drop=0.2, seed=22, g_stride=NR_D):
    """The fused activation gradient (dy_pad) of nr_additive_bwd_flat against the float64 formula and, where the sequence-shaped kernels
    exist, against nr_additive_bwd_act (same dpre / dy_pad up to bf16 rounding of slightly different fp32 sums)."""
    params, ctx_u, hctx, (Wap, bap, qvp), out, aw, go = _setup(be, S, n_seq, None, seed, relu=True)
    hgo, g_off = _wide_rows(be, go, g_stride)
    nwg = be.lib.nr_additive_bwd_flat_grid(n_seq * S)
    dpre = be.poison((n_seq * S, NR_QP), np.uint16)
    dqp = be.poison((nwg, NR_QP), np.float32)
    tot = be.poison((n_seq,), np.float32)
    dy = be.empty((seqpad_rows(n_seq, S), NR_KP), np.uint16)
    ck(be, be.lib.nr_additive_bwd_flat_gs(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(hgo) + g_off, g_stride, be.ptr(out), NR_D,
                                          be.ptr(tot), be.ptr(dpre), be.ptr(dqp), None, be.ptr(dy), p_drop, n_seq, S, 200, be.stream))
    be.sync()
    W, w, dpre_ref, dq_ref = _reference(params, ctx_u, go, S, n_seq, S)
    got = bf16_to_f32(be.np(dpre))
    close_bf16(got[:, :200], dpre_ref.reshape(-1, 200), f'flat pooling bwd (act) dpre S={S}', rel=2.0 ** -7, floor=2e-3)
    dq = be.np(dqp).astype(np.float64).sum(0)
    np.testing.assert_allclose(dq[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())
    dref = got.astype(np.float64)[:, :200] @ W
    full = (dref.reshape(n_seq, S, NR_D) + be.np(aw).astype(np.float64)[:, :, None] * go.astype(np.float64)[:, None, :]) / (1.0 - p_drop)
    full = np.where(bf16_to_f32(ctx_u[:, :NR_D]).reshape(n_seq, S, NR_D) != 0, full, 0.0)
    want = np.zeros((n_seq * S, NR_KP)); want[:, :NR_D] = full.reshape(-1, NR_D)
    want = to_seqpad(want, n_seq, S)
    gdy = bf16_to_f32(be.np(dy)).astype(np.float64)
    assert not gdy[want == 0].any(), 'masked / separator / padding positions must be exact zeros'
    close_bf16(gdy, want, f'flat pooling bwd dy_pad S={S}', rel=2.0 ** -6, floor=2e-3)


def check_flat_bad_args(be):
    pa = be.ptr(be.empty((64,), np.float32))
    lib = be.lib
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, None, NR_D, pa, pa, pa, None, None, 0.0, 4, 20, 200, be.stream) != 0 and b'nr_additive_bwd_flat' in lib.nr_last_error()
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, pa, pa, 0.0, 4, 20, 200, be.stream) != 0          # dctx AND dy_pad
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D - 4, pa, pa, pa, None, None, 0.0, 4, 20, 200, be.stream) != 0   # y stride
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, None, 1.0, 4, 20, 200, be.stream) != 0       # p_drop
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, None, 0.0, 4, 3, 200, be.stream) != 0         # S < 4: more than 16 sequences in 48 tokens
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, pa, 0.0, 4, 15, 200, be.stream) != 0       # S < 16 with dy_pad: more than 4
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, None, 0.0, 0, 20, 200, be.stream) == 0        # nothing to do
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, None, 0.0, 4, 20, 201, be.stream) != 0 and b'200 rows' in lib.nr_last_error()      # query_vector_dim 201 .. 208: the sequence-shaped kernels
    assert lib.nr_additive_bwd_flat(pa, pa, pa, pa, pa, pa, pa, NR_D, pa, pa, pa, None, None, 0.0, 4, 20, 0, be.stream) != 0
    assert lib.nr_additive_bwd_flat_grid(0) == 0 and lib.nr_additive_bwd_flat_grid(1) == 1


def check_flat_scale(be, S=20, n_seq=27136, chunk=2048, act=False, p_drop=0.2):
    """The flat kernel at the bench's launch sizes (every CU busy, tens of iterations per wave) against the float64 restatement evaluated over
    chunks of sequences: dpre, the fused dctx (or the activation gradient dy_pad), and dq summed over all workgroups' partial rows."""
    params, ctx_u, hctx, (Wap, bap, qvp), out, aw, go = _setup(be, S, n_seq, None, 151, relu=act)
    nwg = be.lib.nr_additive_bwd_flat_grid(n_seq * S)
    dpre = be.poison((n_seq * S, NR_QP), np.uint16)
    dqp = be.poison((nwg, NR_QP), np.float32)
    tot = be.poison((n_seq,), np.float32)
    dctx = None if act else be.poison((n_seq * S, NR_KP), np.uint16)
    dy = be.empty((seqpad_rows(n_seq, S), NR_KP), np.uint16) if act else None
    ck(be, be.lib.nr_additive_bwd_flat(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(be.dev(go)), be.ptr(out), NR_D,
                                       be.ptr(tot), be.ptr(dpre), be.ptr(dqp), None if act else be.ptr(dctx), be.ptr(dy) if act else None,
                                       p_drop if act else 0.0, n_seq, S, 200, be.stream))
    be.sync()
    dpre_n, aw_n = be.np(dpre), be.np(aw)
    dctx_n = None if act else be.np(dctx)
    dy_n = bf16_to_f32(be.np(dy)) if act else None
    W = bf16_round(params[A_ + 'linear.weight']).astype(np.float64)
    b = params[A_ + 'linear.bias'].astype(np.float64)
    qv = params[A_ + 'attention_query_vector'].astype(np.float64)
    dq_ref = np.zeros(200)
    for lo in range(0, n_seq, chunk):
        hi = min(lo + chunk, n_seq)
        x = bf16_to_f32(ctx_u[lo * S:hi * S, :NR_D]).astype(np.float64).reshape(hi - lo, S, NR_D)
        _, w, temp = onp.additive(x, W, b, qv)
        g = go[lo:hi].astype(np.float64)
        dw = np.einsum('bd,bsd->bs', g, x)
        ds = w * (dw - (w * dw).sum(1, keepdims=True))
        dpre_ref = ds[:, :, None] * qv[None, None, :] * (1 - temp * temp)
        dq_ref += np.einsum('bs,bsq->q', ds, temp)
        got = bf16_to_f32(dpre_n[lo * S:hi * S])
        close_bf16(got[:, :200], dpre_ref.reshape(-1, 200), f'flat pooling bwd dpre, seqs {lo}..{hi}', rel=2.0 ** -7, floor=2e-3)
        assert not got[:, 200:].any()
        dref = got[:, :200].astype(np.float64) @ W
        if not act:
            close_bf16(bf16_to_f32(dctx_n[lo * S:hi * S, :NR_D]), dref, f'flat pooling bwd dctx, seqs {lo}..{hi}', rel=2.0 ** -7, floor=1e-3)
        else:
            full = (dref.reshape(hi - lo, S, NR_D) + aw_n[lo:hi].astype(np.float64)[:, :, None] * g[:, None, :]) / (1.0 - p_drop)
            full = np.where(x != 0, full, 0.0)
            rows = dy_n[lo * (S + 1) + 1:hi * (S + 1) + 1].reshape(hi - lo, S + 1, NR_KP)      # seqpad: row of token (q, s) = q (S + 1) + s + 1
            gotd = rows[:, :S, :NR_D].astype(np.float64)
            assert not gotd[full == 0].any() and not rows[:, S].any(), 'masked positions and separator rows must be exact zeros'
            close_bf16(gotd.reshape(-1, NR_D), full.reshape(-1, NR_D), f'flat pooling bwd dy_pad, seqs {lo}..{hi}', rel=2.0 ** -6, floor=2e-3)
    dq = be.np(dqp).astype(np.float64).sum(0)
    np.testing.assert_allclose(dq[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())
