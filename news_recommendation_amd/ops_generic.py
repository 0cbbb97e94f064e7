"""GENERAL-GEOMETRY host path: the model-geometry knobs of src/config.py away from the tuned instantiation (word_embedding_dim 300,
num_attention_heads 15, num_filters 300, window_size 3, query_vector_dim <= 208), plus the cross-attention form of MultiHeadSelfAttention.

The reference builds ANY word_embedding_dim, any num_attention_heads dividing it, any num_filters, any odd window_size (src/config.py:34,45,54,
55; model/general/attention/multihead_self.py:27-38; model/NAML/news_encoder.py:10-19; model/LSTUR/news_encoder.py:23).  ops.py / ops_conv.py
send those here.  Same rules as there: every number is produced by a kernel of libnr_engine.so -- the dense contractions by the general ring
GEMMs (nr_gemm_nt / nr_gemm_tn: bf16 operands, fp32 accumulation, the numerics of the tuned path), the rest by csrc/k_generic.h -- PyTorch
owns memory, streams, autograd bookkeeping and the LAYOUT of packed weights (concatenate / pad / cast of parameters, cached per parameter
state).  No CPU path.  The path is built from small autograd Functions (linear layer, attention core, pooling core, convolution, relu+dropout,
gather / dropout) composed like the reference's modules; it is correct at any geometry within the limits below, and NOT tuned: the tuned
geometry never comes here.

Limits: sequence lengths <= 64, d_k <= 32, word_embedding_dim / num_filters multiples of 4 (dropout draws four elements per counter), odd
window_size <= 9."""
import torch

from . import ops
from .ops import _lib, _stream, _call, _ptr, _f32c, _BF16_AS_I16

S_MAX, DK_MAX, WINDOW_MAX = 64, 32, 9


def pad32(n):
    """Row length of a bf16 GEMM operand with n data columns + the 1.0 / bias column, padded to the MFMA k-step."""
    return (n + 1 + 31) // 32 * 32


def _rows_bf16(x2d, d):
    """f32 [n, >= d] -> bf16 [n, pad32(d)], column d = 1.0 (the bias rides the contraction), rest 0."""
    n, dp = x2d.shape[0], pad32(d)
    dst = torch.empty(n, dp, dtype=_BF16_AS_I16, device=x2d.device)
    _call('nr_rows_to_bf16', _lib().nr_rows_to_bf16, _ptr(x2d), x2d.stride(0), d, _ptr(dst), dp, n, _stream())
    return dst


def _pack_rows(W, b):
    """nn.Linear parameters -> the B operand of x W^T + b: bf16 [N, pad32(D)] = [W | b | 0] (layout only; cached per parameter state)."""
    def build():
        N, D = W.shape
        o = torch.zeros(N, pad32(D), dtype=torch.bfloat16, device=W.device)
        o[:, :D] = W.detach()
        if b is not None:
            o[:, D] = b.detach()
        return (o.view(_BF16_AS_I16),)
    return ops._packed('g_rows', (W,) if b is None else (W, b), build)[0]


def _pack_cols(W):
    """W [N, D] -> the B operand of dy W (an NN product run as NT against W^T): bf16 [D, pad32(N)], padding zero."""
    def build():
        N, D = W.shape
        o = torch.zeros(D, pad32(N), dtype=torch.bfloat16, device=W.device)
        o[:, :N] = W.detach().t()
        return (o.view(_BF16_AS_I16),)
    return ops._packed('g_cols', (W,), build)[0]


def _pack_rows_split(W, b):
    """[Wh | Wl | Wh] with W = Wh + Wl split into two bf16 numbers (bias likewise, in the 1.0 columns of the first two blocks): the B operand of
    the 3-term product that evaluates a linear layer to ~2^-16 relative on the bf16 GEMM (nr_g_rows_split_bf16 makes the matching rows)."""
    def build():
        N, D = W.shape
        Dp = pad32(D)
        full = torch.zeros(N, Dp, dtype=torch.float32, device=W.device)
        full[:, :D] = W.detach()
        if b is not None:
            full[:, D] = b.detach()
        hi = full.to(torch.bfloat16)
        lo = (full - hi.float()).to(torch.bfloat16)
        return (torch.cat([hi, lo, hi], dim=1).contiguous().view(_BF16_AS_I16),)
    return ops._packed('g_rows_split', (W,) if b is None else (W, b), build)[0]


def _sum0(parts):
    return ops.sum_parts(parts) if parts[0].numel() % 4 == 0 else parts.sum(dim=0)


class _LinearFn(torch.autograd.Function):
    """y f32 [n, N] = x f32 [n, D] W^T + b (nn.Linear: multihead_self.py:53-55, additive.py:35, NAML ElementEncoder news_encoder.py:46) as one
    nr_gemm_nt on bf16 operands; backward: dW | db = dy^T [x | 1] (nr_gemm_tn, split K), dx = dy W (nr_gemm_nt against W^T)."""

    @staticmethod
    def forward(ctx, x, W, b, precise=False):
        """precise: the forward product from split operands (x = xh + xl, W = Wh + Wl: three bf16 products in one GEMM, ~2^-16 relative) -- for
        a layer whose output feeds a relu that must not flip against the fp32 reference; the backward products stay plain bf16."""
        n, D = x.shape
        N = W.shape[0]
        x = _f32c(x)
        xb = _rows_bf16(x, D)
        if precise:
            Dp = pad32(D)
            x3 = torch.empty(n, 3 * Dp, dtype=_BF16_AS_I16, device=x.device)
            _call('nr_g_rows_split_bf16', _lib().nr_g_rows_split_bf16, _ptr(x), x.stride(0), D, _ptr(x3), Dp, n, _stream())
            y = ops.gemm_nt(x3, _pack_rows_split(W, b), n, N, 3 * Dp, 'nr_gemm_nt[g_linear3]')
        else:
            y = ops.gemm_nt(xb, _pack_rows(W, b), n, N, pad32(D), 'nr_gemm_nt[g_linear]')
        ctx.save_for_backward(xb, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, W = ctx.saved_tensors
        N, D = W.shape
        n = xb.shape[0]
        dyb = _rows_bf16(_f32c(dy), N)
        dW = db = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            ext = _sum0(ops.gemm_tn_parts(dyb, N, xb, pad32(D), 'nr_gemm_tn[g_linear]'))          # [N, pad32(D)]: column D = bias gradient
            dW, db = ext[:, :D], (ext[:, D] if ctx.has_bias else None)
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dyb, _pack_cols(W), n, D, pad32(N), 'nr_gemm_nt[g_linear_dx]')
        return dx, dW, db, None


class _DropoutFn(torch.autograd.Function):
    """F.dropout on f32 elements with the engine's counter-based masks (site, element counter from elem0); the backward applies the same mask."""

    @staticmethod
    def forward(ctx, x, p, seed, site, elem0):
        x = _f32c(x)
        y = torch.empty_like(x)
        _call('nr_g_dropout', _lib().nr_g_dropout, _ptr(x), _ptr(y), x.numel(), elem0, p, seed, site, _stream())
        ctx.meta = (p, seed, site, elem0)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed, site, elem0 = ctx.meta
        g = _f32c(g)
        out = torch.empty_like(g)
        _call('nr_g_dropout', _lib().nr_g_dropout, _ptr(g), _ptr(out), g.numel(), elem0, p, seed, site, _stream())
        return out, None, None, None, None


def dropout(x, p, seed, site, elem0=0):
    if p <= 0.0:
        return x
    if x.numel() % 4 or elem0 % 4:
        raise NotImplementedError("general-geometry dropout: element counts must be multiples of 4 (word_embedding_dim / num_filters % 4 == 0)")
    return _DropoutFn.apply(x, float(p), int(seed), int(site), int(elem0))


class _GatherFn(torch.autograd.Function):
    """nn.Embedding(padding_idx=0) forward (news_encoder.py:38 of every model) on any row width; backward: atomic row scatter, row 0 skipped."""

    @staticmethod
    def forward(ctx, ids, table):
        flat = ids.reshape(-1).contiguous()
        tab = table.detach()
        assert tab.dtype == torch.float32 and tab.is_contiguous()
        d = tab.shape[1]
        out = torch.empty(flat.numel(), d, dtype=torch.float32, device=tab.device)
        _call('nr_gather_rows_strided', _lib().nr_gather_rows_strided, _ptr(flat), _ptr(tab), tab.shape[0], d, None, _ptr(out), d, flat.numel(), _stream())
        ctx.save_for_backward(flat)
        ctx.table_param = table
        return out

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        table = ctx.table_param
        dst, ret = ops.grad_target(table)
        g = _f32c(g)
        _call('nr_rows_scatter_add', _lib().nr_rows_scatter_add, _ptr(flat), _ptr(g), g.shape[1], None, _ptr(dst), table.shape[0], table.shape[1], flat.numel(),
              0, _stream())
        ops.table_grad_ready(table)
        return None, ret


class _AttnCoreFn(torch.autograd.Function):
    """ScaledDotProductAttention (multihead_self.py:15-23; key lengths :60-70) from qkv f32 [n_seq * S, 3 D] -> ctx f32 [n_seq * S, D]."""

    @staticmethod
    def forward(ctx, qkv, key_len, n_seq, S, H, dk):
        qkv = _f32c(qkv)
        out = torch.empty(n_seq * S, H * dk, dtype=torch.float32, device=qkv.device)
        _call('nr_g_attn_fwd', _lib().nr_g_attn_fwd, _ptr(qkv), qkv.stride(0), _ptr(out), _ptr(key_len), n_seq, S, H, dk, _stream())
        ctx.save_for_backward(qkv, key_len)
        ctx.meta = (n_seq, S, H, dk)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, key_len = ctx.saved_tensors
        n_seq, S, H, dk = ctx.meta
        g = _f32c(g)
        dqkv = torch.empty_like(qkv)
        _call('nr_g_attn_bwd', _lib().nr_g_attn_bwd, _ptr(qkv), qkv.stride(0), _ptr(g), _ptr(dqkv), _ptr(key_len), n_seq, S, H, dk, _stream())
        return dqkv, None, None, None, None, None


class _PoolCoreFn(torch.autograd.Function):
    """AdditiveAttention after its linear layer (additive.py:35-52): proj = x Wa^T + ba (from _LinearFn) -> softmax(q . tanh(proj)) pooling of x.
    Returns (out [n_seq, D], attention weights [n_seq, S] -- not differentiable)."""

    @staticmethod
    def forward(ctx, x, proj, qv, n_seq, S, valid):
        x, proj, q = _f32c(x), _f32c(proj), _f32c(qv)
        D, Q = x.shape[1], proj.shape[1]
        out = torch.empty(n_seq, D, dtype=torch.float32, device=x.device)
        aw = torch.empty(n_seq, S, dtype=torch.float32, device=x.device)
        _call('nr_g_additive_fwd', _lib().nr_g_additive_fwd, _ptr(x), x.stride(0), D, _ptr(proj), proj.stride(0), Q, _ptr(q), _ptr(out), D, _ptr(aw),
              n_seq, S, valid, _stream())
        ctx.save_for_backward(x, proj, q, aw)
        ctx.meta = (n_seq, S)
        ctx.mark_non_differentiable(aw)
        return out, aw

    @staticmethod
    def backward(ctx, g, _gaw):
        x, proj, q, aw = ctx.saved_tensors
        n_seq, S = ctx.meta
        D, Q = x.shape[1], proj.shape[1]
        g = _f32c(g)
        dpre = torch.empty_like(proj)
        dq_part = torch.empty(n_seq, Q, dtype=torch.float32, device=x.device)
        _call('nr_g_additive_bwd', _lib().nr_g_additive_bwd, _ptr(x), x.stride(0), D, _ptr(proj), proj.stride(0), Q, _ptr(q), _ptr(aw), _ptr(g), g.stride(0),
              _ptr(dpre), dpre.stride(0), _ptr(dq_part), n_seq, S, _stream())
        dx = torch.empty_like(x)                                     # the direct term w_t g; the path through proj is _LinearFn's backward
        _call('nr_g_rows_axpy', _lib().nr_g_rows_axpy, _ptr(dx), D, _ptr(aw), _ptr(g), g.stride(0), S, D, n_seq * S, 0, _stream())
        return dx, dpre, _sum0(dq_part), None, None, None


def check_geometry(d_model=None, heads=None, S=None, what="sequence length"):
    if S is not None and not (1 <= S <= S_MAX):
        raise NotImplementedError(f"general-geometry path: {what} must be in [1, {S_MAX}] (got {S})")
    if d_model is not None and heads is not None:
        if d_model % heads or d_model // heads > DK_MAX:
            raise NotImplementedError(f"general-geometry path: num_attention_heads must divide word_embedding_dim with d_k <= {DK_MAX} "
                                      f"(got {d_model} / {heads})")


def mhsa(Q, K, V, m, length=None):
    """MultiHeadSelfAttention.forward(Q, K, V, length) (multihead_self.py:46-75) on f32 [n_seq, S, D] inputs (K, V: the same shape)."""
    n_seq, S, D = Q.shape
    H = m.num_attention_heads
    check_geometry(D, H, S)
    dk = D // H
    lin = lambda x, L: _LinearFn.apply(x.reshape(n_seq * S, D), L.weight, L.bias)
    if K is Q and V is Q:
        # one GEMM for the three projections: [Wq; Wk; Wv] stacked (layout only; the gradient flows back through the concatenation)
        qkv = _LinearFn.apply(Q.reshape(n_seq * S, D), torch.cat([m.W_Q.weight, m.W_K.weight, m.W_V.weight], dim=0),
                              torch.cat([m.W_Q.bias, m.W_K.bias, m.W_V.bias], dim=0))
    else:
        qkv = torch.cat([lin(Q, m.W_Q), lin(K, m.W_K), lin(V, m.W_V)], dim=1)
    key_len = None
    if length is not None:
        key_len = length.to(device=Q.device, dtype=torch.int32).reshape(-1).clamp(max=S).contiguous()
        if key_len.numel() != n_seq:
            raise ValueError(f"length must have one entry per sequence ({n_seq}), got {key_len.numel()}")
    return _AttnCoreFn.apply(qkv, key_len, n_seq, S, H, dk).view(n_seq, S, D)


def additive(x, a, return_weights=False):
    """AdditiveAttention.forward (additive.py:27-53) on f32 [n_seq, S, D]."""
    n_seq, S, D = x.shape
    check_geometry(S=S)
    x2 = x.reshape(n_seq * S, D)
    proj = _LinearFn.apply(x2, a.linear.weight, a.linear.bias)
    out, aw = _PoolCoreFn.apply(x2, proj, a.attention_query_vector, n_seq, S, S)
    return (out, aw) if return_weights else out


def embed(ids, table, p, seed, elem0=0):
    """dropout(table[ids]) -> f32 [n, L, D] (news_encoder.py:38-40)."""
    n, L = ids.shape
    x = _GatherFn.apply(ids, table)
    return dropout(x, p, seed, 1, elem0).view(n, L, table.shape[1])


def encode_titles(ids, table, mhsa_mod, additive_mod, p_drop, training):
    """NRMS news encoder (src/model/NRMS/news_encoder.py:27-48) at any geometry: gather -> dropout -> MHSA -> dropout -> additive."""
    p = float(p_drop) if training else 0.0
    seed = ops.new_seed() if p > 0 else 0
    x = embed(ids, table, p, seed)
    y = mhsa(x, x, x, mhsa_mod)
    y = dropout(y, p, seed, 2)
    return additive(y, additive_mod)


def encode_dense(x, mhsa_mod, additive_mod):
    """NRMS user encoder (src/model/NRMS/user_encoder.py:15-26)."""
    x = x.to(torch.float32)
    return additive(mhsa(x, x, x, mhsa_mod), additive_mod)


# ---- convolutional text encoder (NAML news_encoder.py:9-37, LSTUR news_encoder.py:23-30,58-67) ------------------------------------------------
def _pack_conv(W, b):
    """Conv2d(1, F, (w, D)) parameters -> the B operand of the conv-as-GEMM: bf16 [F, w * pad32(D)], tap t at columns t Dp .. t Dp + D, the bias
    in the 1.0 column of the CENTRE tap (the only tap whose row is a real token for every output)."""
    def build():
        F, _, w, D = W.shape
        Dp = pad32(D)
        o = torch.zeros(F, w, Dp, dtype=torch.bfloat16, device=W.device)
        o[:, :, :D] = W.detach()[:, 0]
        o[:, (w - 1) // 2, D] = b.detach()
        return (o.view(F, w * Dp).view(_BF16_AS_I16),)
    return ops._packed('g_conv', (W, b), build)[0]


def _pack_conv_dgrad(W):
    """... -> the B operand of the data gradient dX[m] = sum_u dY[m + u] W[:, w - 1 - u]: bf16 [D, w * pad32(F)] (taps flipped, filters transposed)."""
    def build():
        F, _, w, D = W.shape
        Fp = pad32(F)
        o = torch.zeros(D, w, Fp, dtype=torch.bfloat16, device=W.device)
        o[:, :, :F] = W.detach()[:, 0].flip(1).permute(2, 1, 0)
        return (o.view(D, w * Fp).view(_BF16_AS_I16),)
    return ops._packed('g_conv_d', (W,), build)[0]


def _gemm_rows(A, lda, B, M, N, K, name):
    """nr_gemm_nt whose A rows OVERLAP (row stride lda < K): row m of the product reads the K contiguous elements from A + m * lda -- the w rows of
    a convolution window in the seqpad buffer."""
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _call(name, _lib().nr_gemm_nt_rows, _ptr(A), lda, _ptr(B), B.stride(0), _ptr(C), N, M, N, K, _stream())
    return C


class _ConvFn(torch.autograd.Function):
    """Pre-activation of Conv2d(1, F, (w, D), padding = ((w - 1) / 2, 0)) over each sequence: x f32 [n_seq * S, D] -> y f32 [n_seq * S, F], as ONE
    GEMM over a seqpad copy of x ((w - 1) / 2 zero rows between sequences: a window is w contiguous rows); backward: the w tap gradients as one
    split-K GEMM with taps = w, the data gradient as the same overlapping-row GEMM with flipped taps."""

    @staticmethod
    def forward(ctx, x, W, b, n_seq, S):
        F, _, w, D = W.shape
        pad = (w - 1) // 2
        Dp = pad32(D)
        M = n_seq * (S + pad)
        x = _f32c(x)
        xpad = torch.zeros(M + 2 * pad + 1, Dp, dtype=_BF16_AS_I16, device=x.device)
        _call('nr_g_rows_to_seqpad', _lib().nr_g_rows_to_seqpad, _ptr(x), x.stride(0), D, _ptr(xpad), Dp, S, pad, n_seq * S, 1, _stream())
        yv = _gemm_rows(xpad, Dp, _pack_conv(W, b), M, F, w * Dp, 'nr_gemm_nt[g_conv]')
        y = torch.empty(n_seq * S, F, dtype=torch.float32, device=x.device)
        _call('nr_g_unpad_rows', _lib().nr_g_unpad_rows, _ptr(yv), F, _ptr(y), F, S, pad, n_seq * S, _stream())
        ctx.save_for_backward(xpad, W)
        ctx.meta = (n_seq, S)
        return y

    @staticmethod
    def backward(ctx, dy):
        xpad, W = ctx.saved_tensors
        n_seq, S = ctx.meta
        F, _, w, D = W.shape
        pad = (w - 1) // 2
        Dp, Fp = pad32(D), pad32(F)
        M = n_seq * (S + pad)
        dy = _f32c(dy)
        dypad = torch.zeros(M + 2 * pad + 1, Fp, dtype=_BF16_AS_I16, device=dy.device)
        _call('nr_g_rows_to_seqpad', _lib().nr_g_rows_to_seqpad, _ptr(dy), dy.stride(0), F, _ptr(dypad), Fp, S, pad, n_seq * S, 0, _stream())
        dW = db = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # out[f][t Dp + d] = sum_m dY[m + pad][f] X[m + t][d]: G starts at the first token row of dypad, X at the first row of xpad
            ext = _sum0(ops.gemm_tn_parts(dypad[pad:], F, xpad, Dp, 'nr_gemm_tn[g_conv]', taps=w, n_tok=M)).view(F, w, Dp)
            dW, db = ext[:, :, :D].unsqueeze(1), ext[:, pad, D]
        if ctx.needs_input_grad[0]:
            dxv = _gemm_rows(dypad, Fp, _pack_conv_dgrad(W), M, D, w * Fp, 'nr_gemm_nt[g_conv_dx]')
            dx = torch.empty(n_seq * S, D, dtype=torch.float32, device=dy.device)
            _call('nr_g_unpad_rows', _lib().nr_g_unpad_rows, _ptr(dxv), D, _ptr(dx), D, S, pad, n_seq * S, _stream())
        return dx, dW, db, None, None


class _ReluDropFn(torch.autograd.Function):
    """F.dropout(F.relu(y)) (NAML news_encoder.py:29-32): dropout site 2, counters from elem0; the backward reads the mask off the output's zeros."""

    @staticmethod
    def forward(ctx, y, p, seed, elem0):
        y = _f32c(y)
        n, F = y.shape
        act = torch.empty_like(y)
        _call('nr_g_relu_drop', _lib().nr_g_relu_drop, _ptr(y), F, _ptr(act), F, 1, 0, n, p, seed, elem0, _stream())
        ctx.save_for_backward(act)
        ctx.p = p
        return act

    @staticmethod
    def backward(ctx, g):
        (act,) = ctx.saved_tensors
        g = _f32c(g)
        out = torch.empty_like(g)
        _call('nr_g_relu', _lib().nr_g_relu, _ptr(g), _ptr(act), _ptr(out), g.numel(), 1.0 / (1.0 - ctx.p), _stream())
        return out, None, None, None


def text_encode(ids, table, conv, additive_mod, p_drop, training, seed=None, tok_offset=0):
    """TextEncoder.forward (NAML news_encoder.py:21-37; the title part of LSTUR news_encoder.py:58-67) at any geometry: ids int64 [n, L]."""
    n, L = ids.shape
    F, _, w, D = conv.weight.shape
    if w % 2 == 0 or w > WINDOW_MAX:
        raise NotImplementedError(f"window_size must be odd and <= {WINDOW_MAX} (got {w}); the reference asserts odd (LSTUR/news_encoder.py:23)")
    if D % 4 or F % 4:
        raise NotImplementedError(f"general-geometry path: word_embedding_dim and num_filters must be multiples of 4 (got {D}, {F})")
    check_geometry(S=L, what="text length")
    p = float(p_drop) if training else 0.0
    if seed is None:
        seed = ops.new_seed() if p > 0 else 0
    x = embed(ids, table, p, seed, tok_offset * D)
    y = _ConvFn.apply(x.reshape(n * L, D), conv.weight, conv.bias, n, L)
    act = _ReluDropFn.apply(y, p, seed, tok_offset * F)
    return additive(act.view(n, L, F), additive_mod)


def element_encode(ids, embedding, linear):
    """ElementEncoder.forward (NAML news_encoder.py:40-47): relu(linear(embedding(ids)))."""
    flat = ids.reshape(-1)
    e = _GatherFn.apply(flat, embedding.weight)
    y = _LinearFn.apply(e, linear.weight, linear.bias, True)          # split operands: the relu must not flip against the fp32 reference
    return _ReluDropFn.apply(y, 0.0, 0, 0).view(*ids.shape, linear.weight.shape[0])
