"""Backend-agnostic parity checks of the NAML / LSTUR kernels (convolution, pooling variants, element tables, scatters)
against numpy restatements with the engine's rounding points.  Used with the CPU wave emulator and on cuda:0."""
import numpy as np

from news_recommendation_amd._capi import NR_D, NR_KP, NR_QP
from oracle import nrms_numpy as onp
from tests.backends import bf16_to_f32, f32_to_bf16, bf16_round
from tests.kernel_checks import ck, export_mask, close_bf16, untile

F = NR_D


def conv_params(seed, D=NR_D, Fn=NR_D):
    rng = np.random.default_rng(seed)
    W = (rng.normal(size=(Fn, 1, 3, D)) * (1.0 / (3 * D)) ** 0.5).astype(np.float32)
    b = rng.uniform(-0.05, 0.05, size=Fn).astype(np.float32)
    return W, b


def pack_conv(be, W, b, want_wd=True):
    Wc = be.poison((3, NR_KP, NR_KP), np.uint16)
    Wd = be.poison((3, NR_KP, NR_KP), np.uint16) if want_wd else None
    bc = be.poison((NR_KP,), np.float32)
    ck(be, be.lib.nr_pack_conv(be.ptr(be.dev(W)), be.ptr(be.dev(b)), W.shape[0], W.shape[3], be.ptr(Wc), be.ptr(Wd), be.ptr(bc), be.stream))
    return Wc, Wd, bc


def check_pack_conv(be, D=NR_D, Fn=NR_D):
    W, b = conv_params(1, D, Fn)
    Wc, Wd, bc = pack_conv(be, W, b)
    be.sync()
    Wc, Wd = (np.stack([untile(m[t], NR_KP, NR_KP) for t in range(3)]) for m in (be.np(Wc), be.np(Wd)))
    bc = be.np(bc)
    for t in range(3):
        assert np.array_equal(Wc[t, :Fn, :D], f32_to_bf16(W[:, 0, t, :]))
        assert np.array_equal(Wd[t, :D, :Fn], f32_to_bf16(W[:, 0, 2 - t, :].T))
        assert not Wc[t, Fn:].any() and not Wc[t, :, D:].any() and not Wd[t, D:].any() and not Wd[t, :, Fn:].any()
    assert np.array_equal(bc[:Fn], b) and not bc[Fn:].any()


def conv_ref(x, W, b):
    """y[n,s,f] = b[f] + sum_w sum_d x[n,s+w-1,d] W[f,0,w,d] in float64 (x, W already at their rounding points)."""
    n, S, D = x.shape
    xp = np.zeros((n, S + 2, D))
    xp[:, 1:S + 1] = x
    y = np.zeros((n, S, W.shape[0])) + b
    for w in range(3):
        y += xp[:, w:w + S] @ W[:, 0, w, :].T
    return y


def seqpad_rows(n_seq, S):
    return n_seq * (S + 1) + 1


def to_seqpad(x_tok, n_seq, S):
    """[n_seq*S, C] -> seqpad [n_seq*(S+1)+1, C] with zero separator rows."""
    out = np.zeros((seqpad_rows(n_seq, S), x_tok.shape[1]), dtype=x_tok.dtype)
    idx = np.arange(n_seq * S)
    out[idx + idx // S + 1] = x_tok
    return out


def _fwd_gemm_operand(be, W):
    Wf2 = be.poison((NR_KP, 3 * NR_KP), np.uint16)
    ck(be, be.lib.nr_pack_conv_fwd2(be.ptr(be.dev(W)), W.shape[0], W.shape[3], be.ptr(Wf2), be.stream))
    return Wf2


def check_conv_fwd(be, S=20, n_seq=6, V=300, p_drop=0.0, seed=4321, tok_offset=0, gemm=False):
    """gemm=True: the training forward as gather pass + persistent ring GEMM (nr_conv3_fwd_gemm, csrc/k_convgemm.h EPI) against the same oracle,
    and its x_save bit for bit against the LDS-tile kernel's contract."""
    W, b = conv_params(2)
    rng = np.random.default_rng(3)
    table = rng.normal(0, 0.5, size=(V, NR_D)).astype(np.float32)
    ids = rng.integers(1, V, size=(n_seq, S))
    ids[:, S * 2 // 3:] = 0
    ids[1] = 0
    Wc, _, bc = pack_conv(be, W, b, False)
    act = be.poison((n_seq * S, NR_KP), np.uint16)
    xs = be.poison((seqpad_rows(n_seq, S), NR_KP), np.uint16)      # the kernel writes the separator rows too
    if gemm:
        Wf2 = _fwd_gemm_operand(be, W)
        ck(be, be.lib.nr_conv3_fwd_gemm(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wf2), be.ptr(bc),
                                        be.ptr(act), be.ptr(xs), n_seq, S, S, p_drop, seed, tok_offset, be.stream))
    else:
        ck(be, be.lib.nr_conv3_fwd(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wc), be.ptr(bc), be.ptr(act), be.ptr(xs),
                                   n_seq, S, p_drop, seed, tok_offset, be.stream))
    be.sync()
    x = table[ids].astype(np.float64)
    m2, scale = None, 1.0
    if p_drop > 0:
        scale = np.float32(1.0 / (1.0 - p_drop))
        ntot = (tok_offset + n_seq * S) * NR_D
        m1 = export_mask(be, ntot, p_drop, seed, 1)[tok_offset * NR_D:].reshape(n_seq, S, NR_D)
        m2 = export_mask(be, ntot, p_drop, seed, 2)[tok_offset * NR_D:].reshape(n_seq, S, NR_D)
        assert abs(m1.mean() - (1 - p_drop)) < 0.03
        x = x * m1 * scale
    xq = bf16_round(x.astype(np.float32)).astype(np.float64)
    y = conv_ref(xq, bf16_round(W).astype(np.float64), b.astype(np.float64))
    ref = np.maximum(y, 0.0)
    if m2 is not None:
        ref = ref * m2 * scale
    got_u = be.np(act)
    assert (got_u[:, NR_D] == 0x3F80).all() and not got_u[:, NR_D + 1:].any(), 'activation K-padding wrong'
    got = bf16_to_f32(got_u[:, :NR_D]).astype(np.float64).reshape(ref.shape)
    # bf16 output rounding (2^-8 relative) + fp32 accumulation noise
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all(), f'conv fwd S={S}: max err {err.max():.3g}'
    # saved token matrix: seqpad layout, masked bf16 tokens, col D = 1.0 on token rows, separators zero
    xs_np = be.np(xs)
    want = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    want[:, :NR_D] = xq.reshape(-1, NR_D)
    want[:, NR_D] = 1.0
    assert np.array_equal(bf16_to_f32(xs_np), to_seqpad(want, n_seq, S)), 'x_save mismatch'
    return err.max() / np.abs(ref).max()


def check_conv_fwd_valid(be, S=20, n_seq=6, V=300, valid=13, gemm=False):
    """nr_conv3_fwd_v: texts of `valid` tokens zero-padded to S (padding ids are NOT zero here, to prove they are ignored): the first
    `valid` outputs of every sequence equal the convolution of the truncated text (Conv2d zero padding right after its last token)."""
    W, b = conv_params(2)
    rng = np.random.default_rng(31)
    table = rng.normal(0, 0.5, size=(V, NR_D)).astype(np.float32)
    ids = rng.integers(1, V, size=(n_seq, S))
    Wc, _, bc = pack_conv(be, W, b, False)
    act = be.poison((n_seq * S, NR_KP), np.uint16)
    xs = be.poison((seqpad_rows(n_seq, S), NR_KP), np.uint16)
    if gemm:
        Wf2 = _fwd_gemm_operand(be, W)
        ck(be, be.lib.nr_conv3_fwd_gemm(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wf2), be.ptr(bc),
                                        be.ptr(act), be.ptr(xs), n_seq, S, valid, 0.0, 0, 0, be.stream))
        assert be.lib.nr_conv3_fwd_gemm(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wf2), be.ptr(bc),
                                        be.ptr(act), None, n_seq, S, valid, 0.0, 0, 0, be.stream) != 0 and b'nr_conv3_fwd_gemm' in be.lib.nr_last_error()
    else:
        ck(be, be.lib.nr_conv3_fwd_v(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wc), be.ptr(bc), be.ptr(act), be.ptr(xs),
                                     n_seq, S, valid, 0.0, 0, 0, be.stream))
    be.sync()
    xq = bf16_round(table[ids[:, :valid]]).astype(np.float64)
    ref = np.maximum(conv_ref(xq, bf16_round(W).astype(np.float64), b.astype(np.float64)), 0.0)
    got = bf16_to_f32(be.np(act)[:, :NR_D]).astype(np.float64).reshape(n_seq, S, NR_D)[:, :valid]
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all(), f'conv fwd valid={valid}: max err {err.max():.3g}'
    # the saved token matrix holds zero vectors at the padded positions
    xs_np = bf16_to_f32(be.np(xs))[1:].reshape(-1)[:n_seq * (S + 1) * NR_KP].reshape(n_seq, S + 1, NR_KP)[:, :S, :NR_D]
    assert not xs_np[:, valid:].any() and np.array_equal(xs_np[:, :valid], xq.astype(np.float32))
    assert be.lib.nr_conv3_fwd_v(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), V, be.ptr(Wc), be.ptr(bc), be.ptr(act), be.ptr(xs),
                                 n_seq, S, 0, 0.0, 0, 0, be.stream) != 0


def check_conv_dgrad(be, S=20, n_seq=5):
    W, b = conv_params(5)
    rng = np.random.default_rng(6)
    dy = rng.normal(0, 0.3, size=(n_seq * S, NR_D)).astype(np.float32)
    dy[rng.random(size=dy.shape) < 0.4] = 0.0
    dyu = np.zeros((n_seq * S, NR_KP), dtype=np.uint16)
    dyu[:, :NR_D] = f32_to_bf16(dy)
    dyu[:, NR_D] = 0x3F80            # junk in the padding columns must be ignored
    dy_pad = to_seqpad(dyu, n_seq, S)
    _, Wd, _ = pack_conv(be, W, b)
    dx = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_conv3_dgrad(be.ptr(be.dev(dy_pad)), be.ptr(Wd), be.ptr(dx), n_seq, S, be.stream))
    be.sync()
    dyq = bf16_to_f32(dyu[:, :NR_D]).astype(np.float64).reshape(n_seq, S, NR_D)
    Wq = bf16_round(W).astype(np.float64)
    # dX[s][d] = sum_w sum_f dY[s - w + 1][f] W[f][w][d]
    dyp = np.zeros((n_seq, S + 2, NR_D))
    dyp[:, 1:S + 1] = dyq
    ref = np.zeros((n_seq, S, NR_D))
    for w in range(3):
        ref += dyp[:, 2 - w:2 - w + S] @ Wq[:, 0, w, :]
    got = bf16_to_f32(be.np(dx)[:, :NR_D]).astype(np.float64).reshape(ref.shape)
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all(), f'conv dgrad S={S}: max err {err.max():.3g}'


def check_conv_dgrad_gemm(be, S=20, n_seq=5):
    """nr_conv3_dgrad_gemm (the data gradient as ONE GEMM over virtual 3-tap rows, csrc/k_gemm.h) against numpy; columns >= D exact zeros; rows
    of the token layout complete (every token written once, separator rows of the seqpad input dropped)."""
    W, b = conv_params(5)
    rng = np.random.default_rng(6)
    dy = rng.normal(0, 0.3, size=(n_seq * S, NR_D)).astype(np.float32)
    dy[rng.random(size=dy.shape) < 0.4] = 0.0
    dyu = np.zeros((n_seq * S, NR_KP), dtype=np.uint16)
    dyu[:, :NR_D] = f32_to_bf16(dy)
    dyu[:, NR_D] = 0x3F80            # junk in the padding columns must be ignored (the packed filters are zero there)
    dy_pad = to_seqpad(dyu, n_seq, S)
    Wd2 = be.poison((NR_KP, 3 * NR_KP), np.uint16)
    ck(be, be.lib.nr_pack_conv_dgrad(be.ptr(be.dev(W)), W.shape[0], W.shape[3], be.ptr(Wd2), be.stream))
    dx = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_conv3_dgrad_gemm(be.ptr(be.dev(dy_pad)), be.ptr(Wd2), be.ptr(dx), n_seq, S, be.stream))
    be.sync()
    w2 = be.np(Wd2).reshape(NR_KP, 3, NR_KP)
    for t in range(3):
        assert np.array_equal(w2[:NR_D, t, :NR_D], f32_to_bf16(W[:, 0, 2 - t, :]).T) and not w2[NR_D:, t].any() and not w2[:, t, NR_D:].any()
    dyq = bf16_to_f32(dyu[:, :NR_D]).astype(np.float64).reshape(n_seq, S, NR_D)
    Wq = bf16_round(W).astype(np.float64)
    dyp = np.zeros((n_seq, S + 2, NR_D))
    dyp[:, 1:S + 1] = dyq
    ref = np.zeros((n_seq, S, NR_D))
    for w in range(3):
        ref += dyp[:, 2 - w:2 - w + S] @ Wq[:, 0, w, :]
    out = be.np(dx)
    assert not out[:, NR_D:].any(), 'padding columns of dx must be exact zeros'
    got = bf16_to_f32(out[:, :NR_D]).astype(np.float64).reshape(ref.shape)
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all(), f'conv dgrad (GEMM form) S={S}: max err {err.max():.3g}'
    assert be.lib.nr_conv3_dgrad_gemm(None, be.ptr(Wd2), be.ptr(dx), n_seq, S, be.stream) != 0 and b'nr_conv3_dgrad_gemm' in be.lib.nr_last_error()


def check_conv_dgrad_gemm_scale(be, S=50, n_seq=27136, chunk=1024):
    """nr_conv3_dgrad_gemm at the bench's launch size (NAML abstracts: 27,136 x 50 tokens, 5,300+ output tiles) against the float64 convolution
    evaluated over chunks of sequences: every token row, padding columns exact zeros."""
    W, b = conv_params(5)
    rng = np.random.default_rng(16)
    dyu = np.zeros((n_seq * S, NR_KP), dtype=np.uint16)
    dy = rng.normal(0, 0.3, size=(n_seq * S, NR_D)).astype(np.float32)
    dy[rng.random(size=dy.shape) < 0.4] = 0.0
    dyu[:, :NR_D] = f32_to_bf16(dy)
    del dy
    dy_pad = to_seqpad(dyu, n_seq, S)
    Wd2 = be.poison((NR_KP, 3 * NR_KP), np.uint16)
    ck(be, be.lib.nr_pack_conv_dgrad(be.ptr(be.dev(W)), W.shape[0], W.shape[3], be.ptr(Wd2), be.stream))
    dx = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_conv3_dgrad_gemm(be.ptr(be.dev(dy_pad)), be.ptr(Wd2), be.ptr(dx), n_seq, S, be.stream))
    be.sync()
    out = be.np(dx)
    assert not out[:, NR_D:].any(), 'padding columns of dx must be exact zeros'
    Wq = bf16_round(W).astype(np.float64)
    worst = 0.0
    for lo in range(0, n_seq, chunk):
        hi = min(lo + chunk, n_seq)
        n = hi - lo
        dyp = np.zeros((n, S + 2, NR_D))
        dyp[:, 1:S + 1] = bf16_to_f32(dyu[lo * S:hi * S, :NR_D]).astype(np.float64).reshape(n, S, NR_D)
        ref = np.zeros((n, S, NR_D))
        for w in range(3):
            ref += dyp[:, 2 - w:2 - w + S] @ Wq[:, 0, w, :]
        got = bf16_to_f32(out[lo * S:hi * S, :NR_D]).astype(np.float64).reshape(ref.shape)
        err = np.abs(got - ref)
        bound = 2.0 ** -7 * np.abs(ref) + 2e-4 * np.abs(ref).max()
        assert (err <= bound).all(), f'conv dgrad (GEMM form) S={S} sequences {lo}..{hi}: max err {err.max():.3g}'
        worst = max(worst, float((err / bound).max()))
    return worst


def check_conv_act_bwd(be, S=20, n_seq=7, p_drop=0.2):
    rng = np.random.default_rng(7)
    act = np.maximum(rng.normal(size=(n_seq * S, NR_D)), 0).astype(np.float32)
    act_u = np.zeros((n_seq * S, NR_KP), dtype=np.uint16)
    act_u[:, :NR_D] = f32_to_bf16(act)
    act_u[:, NR_D] = 0x3F80
    dg = f32_to_bf16(rng.normal(0, 0.1, size=(n_seq * S, NR_D)).astype(np.float32))
    aw = rng.random(size=(n_seq, S)).astype(np.float32)
    go = rng.normal(size=(n_seq, NR_D)).astype(np.float32)
    dy = be.empty((seqpad_rows(n_seq, S), NR_KP), np.uint16)
    ck(be, be.lib.nr_conv_act_bwd(be.ptr(be.dev(act_u)), be.ptr(be.dev(dg)), NR_D, be.ptr(be.dev(aw)), be.ptr(be.dev(go)), NR_D, be.ptr(dy),
                                  n_seq, S, p_drop, be.stream))
    be.sync()
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p_drop))
    ref = (bf16_to_f32(dg).reshape(n_seq, S, NR_D) + aw[:, :, None] * go[:, None, :]) * scale
    ref = np.where(bf16_to_f32(act_u[:, :NR_D]).reshape(n_seq, S, NR_D) != 0, ref, 0.0).astype(np.float32)
    want = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    want[:, :NR_D] = bf16_round(ref.reshape(-1, NR_D))
    got = bf16_to_f32(be.np(dy))
    np.testing.assert_allclose(got, to_seqpad(want, n_seq, S), rtol=2.0 ** -7, atol=1e-6)


def additive_params(seed, qdim=200):
    rng = np.random.default_rng(seed)
    return ((rng.normal(size=(qdim, NR_D)) * (1.0 / NR_D) ** 0.5).astype(np.float32), rng.uniform(-0.05, 0.05, size=qdim).astype(np.float32),
            rng.uniform(-0.1, 0.1, size=qdim).astype(np.float32))


def pack_add(be, W, b, q):
    Wap = be.poison((NR_QP, NR_KP), np.uint16); bap = be.poison((NR_QP,), np.float32); qvp = be.poison((NR_QP,), np.float32)
    ck(be, be.lib.nr_pack_additive(be.ptr(be.dev(W)), be.ptr(be.dev(b)), be.ptr(be.dev(q)), W.shape[0], be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.stream))
    return Wap, bap, qvp


def check_additive_ex(be, S=4, n_seq=23, out_stride=900, col0=600):
    """strided f32 output + bf16 ctx-layout copy with a row stride (views buffer) + the S=4 instantiation"""
    W, b, q = additive_params(8)
    rng = np.random.default_rng(9)
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_add(be, W, b, q)
    out = be.dev(np.full((n_seq, out_stride), 7.0, dtype=np.float32))
    ob_stride = 4 * NR_KP
    outb = be.dev(np.full((n_seq * 4, NR_KP), 0x1234, dtype=np.uint16))
    aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd_ex(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out) + col0 * 4, out_stride,
                                     be.ptr(outb) + NR_KP * 2, ob_stride, be.ptr(aw), n_seq, S, be.stream))
    be.sync()
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)
    ref, w, _ = onp.additive(x, bf16_round(W).astype(np.float64), b.astype(np.float64), q.astype(np.float64))
    o = be.np(out)
    np.testing.assert_allclose(be.np(aw), w, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(o[:, col0:col0 + NR_D], ref, rtol=2e-4, atol=2e-5)
    assert (o[:, :col0] == 7.0).all() and (o[:, col0 + NR_D:] == 7.0).all()
    ob = be.np(outb).reshape(n_seq, 4, NR_KP)
    assert (ob[:, [0, 2, 3]] == 0x1234).all()
    assert np.array_equal(ob[:, 1, :NR_D], f32_to_bf16(o[:, col0:col0 + NR_D]))
    assert (ob[:, 1, NR_D] == 0x3F80).all() and not ob[:, 1, NR_D + 1:].any()


def check_additive_bwd_act(be, S=20, n_seq=7, p_drop=0.2):
    """nr_additive_bwd_act == nr_additive_bwd_ex followed by nr_conv_act_bwd (to bf16 rounding: the fused epilogue rounds the sum once, the
    two-kernel form rounds dctx and the result), on activations with exact zeros (relu / dropout) and against the float64 formula."""
    W, b, q = additive_params(21)
    rng = np.random.default_rng(22)
    act = np.maximum(rng.normal(0, 0.6, size=(n_seq * S, NR_D)), 0)                 # ~half of the entries are exact zeros
    act[rng.random(size=act.shape) < 0.2] = 0
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = act
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_add(be, W, b, q)
    hctx = be.dev(ctx_u)
    out = be.poison((n_seq, NR_D), np.float32); aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), be.ptr(aw), n_seq, S, be.stream))
    go = rng.normal(size=(n_seq, NR_D)).astype(np.float32)
    hgo = be.dev(go)
    nwg = be.lib.nr_additive_bwd_grid(n_seq, S)
    WaT = be.poison((NR_KP, 224), np.uint16)
    ck(be, be.lib.nr_pack_additive_t(be.ptr(be.dev(W)), W.shape[0], be.ptr(WaT), be.stream))
    # two kernels
    dpre1 = be.empty((n_seq * S, NR_QP), np.uint16); dqp1 = be.poison((nwg, NR_QP), np.float32)
    dctx1 = be.empty((n_seq * S, NR_KP), np.uint16); dy1 = be.empty((seqpad_rows(n_seq, S), NR_KP), np.uint16)
    ck(be, be.lib.nr_additive_bwd_ex(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(hgo), be.ptr(dpre1), be.ptr(dqp1),
                                     be.ptr(WaT), be.ptr(dctx1), n_seq, S, be.stream))
    ck(be, be.lib.nr_conv_act_bwd(be.ptr(hctx), be.ptr(dctx1), NR_KP, be.ptr(aw), be.ptr(hgo), NR_D, be.ptr(dy1), n_seq, S, p_drop, be.stream))
    # one call
    dpre2 = be.empty((n_seq * S, NR_QP), np.uint16); dqp2 = be.poison((nwg, NR_QP), np.float32)
    scratch = be.empty((n_seq * S, NR_KP), np.uint16); dy2 = be.empty((seqpad_rows(n_seq, S), NR_KP), np.uint16)
    ck(be, be.lib.nr_additive_bwd_act(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(hgo), be.ptr(dpre2), be.ptr(dqp2),
                                      be.ptr(WaT), be.ptr(scratch), be.ptr(dy2), p_drop, n_seq, S, be.stream))
    be.sync()
    assert np.array_equal(be.np(dpre1), be.np(dpre2)) and np.array_equal(be.np(dqp1), be.np(dqp2))
    # float64 reference from the kernel's own dpre (bit-level operands) and forward weights
    dref = bf16_to_f32(be.np(dpre2)).astype(np.float64)[:, :W.shape[0]] @ bf16_round(W).astype(np.float64)
    full = (dref.reshape(n_seq, S, NR_D) + be.np(aw).astype(np.float64)[:, :, None] * go.astype(np.float64)[:, None, :]) / (1.0 - p_drop)
    full = np.where(bf16_to_f32(ctx_u[:, :NR_D]).reshape(n_seq, S, NR_D) != 0, full, 0.0)
    want = np.zeros((n_seq * S, NR_KP)); want[:, :NR_D] = full.reshape(-1, NR_D)
    want = to_seqpad(want, n_seq, S)
    for name, dy in (('two kernels', dy1), ('one call', dy2)):
        got = bf16_to_f32(be.np(dy)).astype(np.float64)
        assert not got[want == 0].any(), f'{name}: masked / separator / padding positions must be exact zeros'
        close_bf16(got, want, f'additive_bwd_act {name} S={S}', rel=2.0 ** -6, floor=2e-3)


def check_additive_bwd_s4(be, n_seq=45):
    S = 4
    W, b, q = additive_params(10)
    rng = np.random.default_rng(11)
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_add(be, W, b, q)
    hctx = be.dev(ctx_u)
    out = be.poison((n_seq, NR_D), np.float32); aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), be.ptr(aw), n_seq, S, be.stream))
    go = rng.normal(size=(n_seq, NR_D)).astype(np.float32)
    nwg = be.lib.nr_additive_bwd_grid(n_seq, S)
    assert nwg == (n_seq + 19) // 20
    dpre = be.empty((n_seq * S, NR_QP), np.uint16); dqp = be.poison((nwg, NR_QP), np.float32)
    hgo = be.dev(go)
    ck(be, be.lib.nr_additive_bwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(hgo), be.ptr(dpre), be.ptr(dqp),
                                  n_seq, S, be.stream))
    be.sync()
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)
    Wq = bf16_round(W).astype(np.float64)
    _, w, temp = onp.additive(x, Wq, b.astype(np.float64), q.astype(np.float64))
    g = go.astype(np.float64)
    dw = np.einsum('bd,bsd->bs', g, x)
    ds = w * (dw - (w * dw).sum(1, keepdims=True))
    dpre_ref = ds[:, :, None] * q[None, None, :].astype(np.float64) * (1 - temp * temp)
    close_bf16(bf16_to_f32(be.np(dpre))[:, :200], dpre_ref.reshape(-1, 200), 'additive_bwd S=4 dpre', rel=2.0 ** -7, floor=2e-3)
    dq_ref = np.einsum('bs,bsq->q', ds, temp)
    np.testing.assert_allclose(be.np(dqp).astype(np.float64).sum(0)[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())
    # additive_dx: dx = dgemm + aw (x) g, token-major and view-major
    dgemm = f32_to_bf16(rng.normal(0, 0.2, size=(n_seq * S, NR_D)).astype(np.float32))
    hdg = be.dev(dgemm)
    ref = bf16_to_f32(dgemm).reshape(n_seq, S, NR_D) + be.np(aw)[:, :, None] * go[:, None, :]
    for vm in (0, 1):
        dx = be.poison((n_seq * S, NR_D), np.float32)
        ck(be, be.lib.nr_additive_dx(be.ptr(hdg), NR_D, be.ptr(aw), be.ptr(hgo), be.ptr(dx), n_seq, S, vm, be.stream))
        be.sync()
        got = be.np(dx)
        got = got.reshape(S, n_seq, NR_D).transpose(1, 0, 2) if vm else got.reshape(n_seq, S, NR_D)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)


def check_element_tables(be, ncat=275, dcat=100, T=333):
    rng = np.random.default_rng(12)
    emb = rng.normal(0, 0.5, size=(ncat, dcat)).astype(np.float32)
    emb[0] = 0
    Ws = [(rng.normal(size=(NR_D, dcat)) * (1.0 / dcat) ** 0.5).astype(np.float32) for _ in range(2)]
    bs = [rng.uniform(-0.05, 0.05, size=NR_D).astype(np.float32) for _ in range(2)]
    E = be.poison((2, ncat, NR_D), np.float32)
    hemb, hW, hb = be.dev(emb), [be.dev(w) for w in Ws], [be.dev(b) for b in bs]
    ck(be, be.lib.nr_element_table_fwd(be.ptr(hemb), ncat, dcat, be.ptr(hW[0]), be.ptr(hb[0]), be.ptr(hW[1]), be.ptr(hb[1]), be.ptr(E), be.stream))
    be.sync()
    Eref = np.stack([np.maximum(emb.astype(np.float64) @ Ws[i].T.astype(np.float64) + bs[i], 0) for i in range(2)])
    np.testing.assert_allclose(be.np(E), Eref, rtol=1e-5, atol=1e-5)
    # views_fill
    cat = rng.integers(0, ncat, size=T).astype(np.int64); sub = rng.integers(0, ncat, size=T).astype(np.int64)
    views = be.dev(np.full((T * 4, NR_KP), 0x1234, dtype=np.uint16))
    ck(be, be.lib.nr_views_fill(be.ptr(be.dev(cat)), be.ptr(be.dev(sub)), be.ptr(E), ncat, be.ptr(views), T, be.stream))
    be.sync()
    v = be.np(views).reshape(T, 4, NR_KP)
    En = be.np(E)
    assert (v[:, :2] == 0x1234).all()
    assert np.array_equal(v[:, 2, :NR_D], f32_to_bf16(En[0][cat])) and np.array_equal(v[:, 3, :NR_D], f32_to_bf16(En[1][sub]))
    assert (v[:, 2:, NR_D] == 0x3F80).all() and not v[:, 2:, NR_D + 1:].any()
    # backward
    dE = rng.normal(size=(2, ncat, NR_D)).astype(np.float32)
    dW = be.poison((2, NR_D, dcat), np.float32); db = be.poison((2, NR_D), np.float32); demb = be.poison((ncat, dcat), np.float32)
    ck(be, be.lib.nr_element_table_bwd(be.ptr(hemb), ncat, dcat, be.ptr(hW[0]), be.ptr(hW[1]), be.ptr(E), be.ptr(be.dev(dE)), be.ptr(dW), be.ptr(db),
                                       be.ptr(demb), be.stream))
    be.sync()
    dpre = np.where(En > 0, dE, 0).astype(np.float64)
    np.testing.assert_allclose(be.np(dW), np.einsum('wcf,ck->wfk', dpre, emb.astype(np.float64)), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(be.np(db), dpre.sum(1), rtol=1e-4, atol=1e-4)
    de = sum(dpre[i] @ Ws[i].astype(np.float64) for i in range(2))
    de[0] = 0
    np.testing.assert_allclose(be.np(demb), de, rtol=1e-4, atol=1e-4)


def check_row_scatters(be, n=777, rows=40):
    rng = np.random.default_rng(13)
    ids = rng.integers(0, rows, size=n).astype(np.int64)
    ld = 900
    src = rng.normal(size=(n, ld)).astype(np.float32)
    perm = np.argsort(ids, kind='stable').astype(np.int64)
    hsrc = be.dev(src)
    for pad_row, col0 in ((0, 300), (-1, 0)):
        dst = be.dev(np.zeros((rows, NR_D), dtype=np.float32))
        ck(be, be.lib.nr_scatter_sorted_f32(be.ptr(be.dev(ids[perm])), be.ptr(be.dev(perm)), be.ptr(hsrc) + col0 * 4, ld, be.ptr(dst), rows, n,
                                            pad_row, be.stream))
        be.sync()
        ref = np.zeros((rows, NR_D))
        sel = ids > pad_row
        np.add.at(ref, ids[sel], src[sel, col0:col0 + NR_D].astype(np.float64))
        np.testing.assert_allclose(be.np(dst), ref, rtol=1e-5, atol=1e-4)
    # generic atomic scatter with a per-row factor, d = 900
    scale = np.where(rng.random(size=n) < 0.5, 0.0, 2.0).astype(np.float32)
    dst = be.dev(np.zeros((rows, ld), dtype=np.float32))
    ck(be, be.lib.nr_rows_scatter_add(be.ptr(be.dev(ids)), be.ptr(hsrc), ld, be.ptr(be.dev(scale)), be.ptr(dst), rows, ld, n, 0, be.stream))
    be.sync()
    ref = np.zeros((rows, ld))
    sel = ids > 0
    np.add.at(ref, ids[sel], (src[sel] * scale[sel, None]).astype(np.float64))
    np.testing.assert_allclose(be.np(dst), ref, rtol=1e-5, atol=1e-4)
    # strided gather with a per-row factor
    table = rng.normal(size=(rows, NR_D)).astype(np.float32)
    out = be.dev(np.full((n, ld), 3.0, dtype=np.float32))
    ck(be, be.lib.nr_gather_rows_strided(be.ptr(be.dev(ids)), be.ptr(be.dev(table)), rows, NR_D, be.ptr(be.dev(scale)), be.ptr(out) + 300 * 4, ld, n, be.stream))
    be.sync()
    o = be.np(out)
    assert np.array_equal(o[:, 300:600], table[ids] * scale[:, None]) and (o[:, :300] == 3.0).all() and (o[:, 600:] == 3.0).all()
