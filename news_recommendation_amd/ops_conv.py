"""Host wrappers (autograd) of the convolutional news encoders: NAML (title + abstract text encoders, category /
subcategory element encoders, final attention over the 4 views) and LSTUR (category / subcategory embedding rows ++
title text encoder).  Same rules as ops.py: all encoder math runs in the HIP kernels of libnr_engine.so, PyTorch owns
memory, streams and autograd bookkeeping, and the only library calls are plain bf16 GEMMs of the backward pass.
No CPU path.
"""
import os

import torch

from . import ops
from .ops import (_lib, _stream, _call, _ptr, _f32c, _bf16, _workspace, _require_cuda, pack_additive,
                  _BF16_AS_I16, NR_D, NR_KP, NR_QP)

WGRAD_CHUNKS = 64     # batched-GEMM chunks of the conv weight gradient ([320 x rows]^T x [rows x 320] per tap: ~4 output tiles each)


def conv_tuned(word_embedding_dim, num_filters, window, qdim):
    """Whether the TUNED conv / pooling kernels take this geometry (word_embedding_dim = num_filters = 300, window_size 3, query_vector_dim <=
    208: src/config.py:34,39,54,55); anything else -- any dims that are multiples of 4, any odd window <= 9 -- runs the general-geometry path
    (ops_generic.py)."""
    return word_embedding_dim == NR_D and num_filters == NR_D and window == 3 and 0 < qdim <= NR_QP


def check_conv_dims(word_embedding_dim, num_filters, window, qdim):
    """Constructor-time check of a conv text encoder's geometry: what NEITHER path can run raises here (the reference asserts an odd window,
    LSTUR/news_encoder.py:23)."""
    if conv_tuned(word_embedding_dim, num_filters, window, qdim):
        return
    from . import ops_generic
    if window < 1 or window % 2 == 0 or window > ops_generic.WINDOW_MAX:
        raise NotImplementedError(f"window_size must be odd and in [1, {ops_generic.WINDOW_MAX}] (got {window})")
    if word_embedding_dim % 4 or num_filters % 4 or qdim < 1:
        raise NotImplementedError(f"word_embedding_dim and num_filters must be multiples of 4 (got {word_embedding_dim}, {num_filters})")


def _module_tuned(conv, additive, L):
    F, _, w, D = conv.weight.shape
    return conv_tuned(D, F, w, additive.linear.weight.shape[0]) and 1 <= L <= 50


def pack_conv(W, b):
    """Conv2d(1,F,(3,D)) parameters -> (Wc, Wd, bc) bf16/f32 operands of the live parameters, cached per parameter state
    (ops._packed)."""
    def build():
        dev = W.device
        Wc = torch.empty(3, NR_KP, NR_KP, dtype=_BF16_AS_I16, device=dev)
        Wd = torch.empty(3, NR_KP, NR_KP, dtype=_BF16_AS_I16, device=dev)
        bc = torch.empty(NR_KP, dtype=torch.float32, device=dev)
        Wf, bf = _f32c(W), _f32c(b)
        _call('nr_pack_conv', _lib().nr_pack_conv, _ptr(Wf), _ptr(bf), W.shape[0], W.shape[3], _ptr(Wc), _ptr(Wd), _ptr(bc), _stream())
        return Wc, Wd, bc
    return ops._packed('conv', (W, b), build)


def pack_conv_dgrad(W):
    """Conv2d weight -> the A operand of the data gradient in GEMM form (nr_conv3_dgrad_gemm): bf16 [KP][3 KP] row-major, cached per state."""
    def build():
        Wd2 = torch.empty(NR_KP, 3 * NR_KP, dtype=_BF16_AS_I16, device=W.device)
        _call('nr_pack_conv_dgrad', _lib().nr_pack_conv_dgrad, _ptr(_f32c(W)), W.shape[0], W.shape[3], _ptr(Wd2), _stream())
        return (Wd2,)
    return ops._packed('conv_dgrad', (W,), build)[0]


def pack_conv_fwd2(W):
    """Conv2d weight -> the A operand of the training forward in GEMM form (nr_conv3_fwd_gemm): bf16 [KP][3 KP] row-major, cached per state."""
    def build():
        Wf2 = torch.empty(NR_KP, 3 * NR_KP, dtype=_BF16_AS_I16, device=W.device)
        _call('nr_pack_conv_fwd2', _lib().nr_pack_conv_fwd2, _ptr(_f32c(W)), W.shape[0], W.shape[3], _ptr(Wf2), _stream())
        return (Wf2,)
    return ops._packed('conv_fwd2', (W,), build)[0]


def conv_fwd_gemm_ok(n_seq, S):
    """The training forward as gather pass + persistent ring GEMM (csrc/k_convgemm.h): NR_CONV_FWD_GEMM (default below), output rows within 2 GiB."""
    return _CONV_FWD_GEMM and n_seq * S * NR_KP * 2 < 2 ** 31


_CONV_FWD_GEMM = os.environ.get('NR_CONV_FWD_GEMM', '0') != '0'


def _seqpad_alloc(n_seq, S):
    """(seqpad rows, chunk count, allocated rows = chunk count x chunk rows)."""
    rp = n_seq * (S + 1) + 1
    nc = WGRAD_CHUNKS if rp >= WGRAD_CHUNKS * 512 else max(1, rp // 512)
    chunk = ((rp + nc - 1) // nc + 7) // 8 * 8
    return rp, nc, nc * chunk


class _TextState:
    """What one text encoder keeps between forward and backward."""
    __slots__ = ('S', 'n_seq', 'act', 'xstore', 'aw', 'Wd', 'Wd2', 'Wap', 'bap', 'qvp', 'WaT', 'qdim', 'tok_offset', 'y', 'y_ptr', 'y_stride', 'y_version',
                 'params')


def pad_text(ids, what):
    """int64 [n, L] (L in [1, 50]) -> ([n, S] zero-padded to an instantiated length, L): ops.padded_len."""
    L = ids.shape[1]
    S = ops.padded_len(L, what)
    return (torch.nn.functional.pad(ids, (0, S - L)) if S != L else ids).contiguous(), L


def text_fwd(ids, table, conv_w, conv_b, Wa, ba, qv, p, seed, tok_offset, need_grad, out, out_stride, out_b, out_b_stride, tag, valid=None, y_keep=None):
    """gather -> dropout -> conv3 -> relu -> dropout -> additive pooling for ids int64 [n_seq, S] on the GPU.
    Pooled vectors go to `out` (f32 rows of stride out_stride, may be None) and/or `out_b` (bf16 ctx rows).  valid (<= S): real text
    length when the ids were zero-padded (pad_text): padded positions are zero vectors for the convolution and outside the pooling."""
    lib = _lib()
    n_seq, S = ids.shape
    valid = S if valid is None else int(valid)
    if not lib.nr_supported_conv_len(S):
        raise NotImplementedError(f"text length {S} is not instantiated in the HIP conv kernels (20, 50)")
    dev = table.device
    st = _TextState()
    st.S, st.n_seq, st.tok_offset, st.qdim = S, n_seq, tok_offset, Wa.shape[0]
    st.params = (conv_w, conv_b, Wa, ba, qv)              # the caller's tensor objects (nn.Parameters): ops.inplace_grads()
    Wc, st.Wd, bc = pack_conv(conv_w, conv_b)
    st.Wd2 = pack_conv_dgrad(conv_w) if need_grad else None       # operand of the data gradient as ONE GEMM over virtual 3-tap rows (csrc/k_gemm.h, NT3 form)
    st.Wap, st.bap, st.qvp = pack_additive(Wa, ba, qv)
    st.WaT = ops.pack_additive_t(Wa) if (need_grad and not ops.pool_flat_ok(S, True, n_seq, qdim=st.qdim)) else None
    st.act = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
    xs_ptr = None
    st.xstore = None
    if need_grad:
        rp, nc, ra = _seqpad_alloc(n_seq, S)
        # row 0 = the "-1" row of the tap shift; it and the tail rows past the last sequence are never written by the kernel: zeroed when the
        # buffer is created, and the previous step's buffer is reused once its backward has read it (ops.step_buffer: no fills per step)
        def init(t):
            t[0].zero_()
            t[rp + 1:].zero_()
        st.xstore = ops.step_buffer(f'xstore[{tag}]', (ra + 2, NR_KP), _BF16_AS_I16, dev, init)
        xs_ptr = st.xstore.data_ptr() + NR_KP * 2
    tab = table.detach()
    assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[1] == NR_D
    if need_grad and conv_fwd_gemm_ok(n_seq, S):
        _call(f'nr_conv3_fwd[{tag}]', lib.nr_conv3_fwd_gemm, _ptr(ids), _ptr(tab), tab.shape[0], _ptr(pack_conv_fwd2(conv_w)), _ptr(bc), _ptr(st.act),
              xs_ptr, n_seq, S, valid, p, seed, tok_offset, _stream())
    else:
        _call(f'nr_conv3_fwd[{tag}]', lib.nr_conv3_fwd_v, _ptr(ids), _ptr(tab), tab.shape[0], _ptr(Wc), _ptr(bc), _ptr(st.act), xs_ptr,
              n_seq, S, valid, p, seed, tok_offset, _stream())
    st.aw = torch.empty(n_seq, S, dtype=torch.float32, device=dev)
    # the pooled vectors in f32 are an operand of the backward (csrc/k_pool3.h): a private buffer when the caller only wants the bf16 copy;
    # a caller-owned `out` is the autograd function's own output, referenced through y_keep (detached: no reference cycle)
    st.y = None
    st.y_version = None
    if need_grad and ops.pool_flat_ok(S, True, n_seq, qdim=st.qdim):
        if out is None:
            st.y = torch.empty(n_seq, NR_D, dtype=torch.float32, device=dev)
            out, out_stride = st.y.data_ptr(), NR_D
        else:
            if y_keep is None or y_keep.data_ptr() != out or y_keep.stride(0) != out_stride:
                raise ValueError("text_fwd: y_keep must be the tensor view behind `out`")
            st.y = y_keep.detach()
            st.y_version = y_keep._version          # (the buffer is the autograd function's own output: an in-place edit before the backward would go unnoticed)
    st.y_ptr, st.y_stride = out, out_stride
    if ops.pool_fwd_flat_ok(S, n_seq, qdim=st.qdim):
        # whole sequences per wave, persistent, the projection matrix resident in LDS (csrc/k_pool4.h)
        _call(f'nr_additive_fwd[{tag}]', lib.nr_additive_fwd_flat, _ptr(st.act), _ptr(st.Wap), _ptr(st.bap), _ptr(st.qvp), out, out_stride,
              out_b, out_b_stride, _ptr(st.aw), n_seq, S, valid, st.qdim, _stream())
    else:
        _call(f'nr_additive_fwd[{tag}]', lib.nr_additive_fwd_v, _ptr(st.act), _ptr(st.Wap), _ptr(st.bap), _ptr(st.qvp), out, out_stride,
              out_b, out_b_stride, _ptr(st.aw), n_seq, S, valid, _stream())
    return st


def _pool_bwd(ctx_b, Wap, bap, qvp, aw, g, n_seq, S, qdim, tag, WaT, dy=None, p_drop=0.0, y_ptr=None, y_stride=NR_D, later=False, g_stride=NR_D):
    """Backward of one additive-attention pooling level up to the GEMM part of its input gradient.
    Returns (d_Wa, d_ba, d_qv, dgemm bf16 [n_seq*S][KP]); the caller adds the direct term aw (x) g.
    With ``dy`` (seqpad buffer of a conv text encoder whose activations ctx_b are) the gradient goes on through the relu / dropout stage
    into dy in the same call (nr_additive_bwd_act: fused into the pooling kernel's epilogue where the register-resident kernels run);
    dgemm is then only scratch.
    later: the weight-gradient part (dWa = dpre^T [ctx | 1], the sum of the dq partials) is NOT computed here: returns (fn, dgemm) with fn() ->
    (d_Wa, d_ba, d_qv); dpre / dq live in scratch keyed by `tag` until then (two-phase backward: ops.defer_wgrad)."""
    lib = _lib()
    dev = ctx_b.device
    ntok = n_seq * S
    flat = ops.pool_flat_ok(S, dy is not None, n_seq, qdim=qdim) and y_ptr is not None
    wt = tag if later else ''
    if flat:
        dpre, dq_part, dgemm = ops.pool_bwd_flat(ctx_b, Wap, bap, qvp, aw, g, y_ptr, y_stride, n_seq, S, qdim, tag, dy=dy, p_drop=p_drop, ws_tag=wt,
                                                 g_stride=g_stride, dq_ring=True)
    else:
        if g_stride != NR_D:
            raise ValueError("_pool_bwd: only the flat kernel reads a strided sequence gradient")
        nwg = lib.nr_additive_bwd_grid(n_seq, S)
        dpre = _workspace(f'dpre[{wt}]' if wt else 'dpre', (ntok, NR_QP), _BF16_AS_I16, dev)
        dq_part = ops.dq_slot(nwg, dev)
        if dq_part is None:
            dq_part = _workspace(f'dqp[{wt}]' if wt else 'dqp', (nwg, NR_QP), torch.float32, dev)
        dgemm = _workspace(f'dctx[{tag}]', (ntok, NR_KP), _BF16_AS_I16, dev)        # = dpre @ Wa, produced inside the kernel
    if flat:
        pass
    elif dy is None:
        _call(f'nr_additive_bwd[{tag}]', lib.nr_additive_bwd_ex, _ptr(ctx_b), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(aw), _ptr(g), _ptr(dpre),
              _ptr(dq_part), _ptr(WaT), _ptr(dgemm), n_seq, S, _stream())
    else:
        _call(f'nr_additive_bwd[{tag}]', lib.nr_additive_bwd_act, _ptr(ctx_b), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(aw), _ptr(g), _ptr(dpre),
              _ptr(dq_part), _ptr(WaT), _ptr(dgemm), _ptr(dy), p_drop, n_seq, S, _stream())

    ring = ops.is_dq_slot(dq_part)

    def weight_part():
        # the per-workgroup partial rows of the query-vector gradient: summed by the backward pass's one accumulate launch when the trainer owns
        # the gradient buffers (ops.PartSum -> nr_accum_many), by nr_sum_parts otherwise (ops.hand_over_grads / the callers' materialize())
        d_qv = ops.PartSum(dq_part, qdim) if ring else ops.sum_parts(dq_part)[:qdim]
        # split-K ring kernel (csrc/k_gemm.h), one 256 x 320 tile per token partition, partials summed in fixed order
        dWa_ext = ops.sum_parts(ops.gemm_tn_parts(dpre, NR_QP, ctx_b, NR_KP, f'nr_gemm_tn_dWa[{tag}]'))
        return dWa_ext[:qdim, :NR_D], dWa_ext[:qdim, NR_D], d_qv
    if later:
        return weight_part, dgemm
    return (*weight_part(), dgemm)


_G_STRIDED = os.environ.get('NR_POOL_G_STRIDED', '1') == '1'       # A/B knob: 0 = LSTUR's title third of the gradient through a contiguous copy


def text_bwd_strided_ok(st):
    """Whether text_bwd reads the pooled-vector gradient in place from wider rows (g_stride > D): the flat pooling backward does (csrc/k_pool3.h)."""
    return _G_STRIDED and ops.pool_flat_ok(st.S, True, st.n_seq, qdim=st.qdim) and st.y_ptr is not None


def text_bwd(st, g, g_stride, p, dx_out, tag, later=False):
    """Backward of text_fwd for pooled-vector gradients g (f32 device pointer/tensor rows of stride g_stride).
    Writes the token gradient (bf16 [n_seq*S][KP]) into dx_out and returns (d_conv_w, d_conv_b, d_Wa, d_ba, d_qv) -- or, with later=True,
    a function that computes them: phase 1 (here) is what the INPUT gradient needs (pooling backward with the fused activation gradient, the
    data-gradient GEMM), phase 2 (the function) the weight gradients (pooling dWa, the three conv tap gradients): under data parallelism phase
    2 runs while the table bucket, complete once every text's phase 1 and the embedding scatter are done, is on the wire."""
    lib = _lib()
    n_seq, S = st.n_seq, st.S
    dev = st.act.device
    if g_stride != NR_D and not text_bwd_strided_ok(st):
        raise ValueError("text_bwd: the pooled-vector gradient must be contiguous [n_seq, D] on this path (text_bwd_strided_ok)")
    rp, nc, ra = _seqpad_alloc(n_seq, S)
    dy = _workspace(f'dy{S}', (ra, NR_KP), _BF16_AS_I16, dev, zero=True)              # separator / tail rows stay zero
    # pooling backward and the relu / dropout gradient of the conv stage in one call: dy = (dpre @ Wa + aw (x) g) * [act != 0] / (1 - p)
    if getattr(st, 'y_version', None) is not None and st.y._version != st.y_version:
        raise RuntimeError("text_bwd: the pooled vectors were modified in place after the forward; the pooling backward needs them unchanged")
    pool_w, _ = _pool_bwd(st.act, st.Wap, st.bap, st.qvp, st.aw, g, n_seq, S, st.qdim, tag, st.WaT, dy=dy, p_drop=p,
                          y_ptr=st.y_ptr, y_stride=st.y_stride, later=True, g_stride=g_stride)
    # the data gradient as ONE GEMM over virtual 3-tap rows of dy (csrc/k_gemm.h, NT3 form)
    _call(f'nr_conv3_dgrad[{tag}]', lib.nr_conv3_dgrad_gemm, _ptr(dy), _ptr(st.Wd2), dx_out, n_seq, S, _stream())
    xstore = st.xstore

    def weight_part(dst=None):
        d_Wa, d_ba, d_qv = pool_w()
        # the three tap gradients as ONE hand-written 3-tap GEMM (csrc/k_gemm.h): out[f][w * KP + d] = sum_rows dY[row][f] X[row + w][d] -- the tap
        # shift is a row offset of the seqpad store, so the virtual operand row is [x[row], x[row + 1], x[row + 2]] and dY is fetched once for all three
        both = ops.sum_parts(ops.gemm_tn_parts(dy, NR_KP, xstore, NR_KP, f'nr_gemm_tn_dWconv[{tag}]', taps=3, n_tok=ra))
        ops.step_buffer_release(xstore)              # last reader: the next forward may overwrite the token store
        taps = [both[:, w * NR_KP:(w + 1) * NR_KP] for w in range(3)]
        d_conv_b = taps[1][:NR_D, NR_D]                                                  # X column D is 1.0 on token rows
        if dst is None and isinstance(d_qv, ops.PartSum):
            d_qv = d_qv.materialize()
        if dst is not None:
            # the trainer's persistent buffers (ops.inplace_grads): tap w of the filter gradient is the strided sub-matrix [:, 0, w, :] of the
            # [F, 1, 3, D] buffer -- queued as it is, no stacked copy
            for w in range(3):
                ops.queue_grad(dst[0][:, 0, w, :], taps[w][:NR_D, :NR_D])
            for d_, v in zip(dst[1:], (d_conv_b, d_Wa, d_ba, d_qv)):
                ops.queue_grad(d_, v)
            return None
        d_conv_w = torch.stack([t[:NR_D, :NR_D] for t in taps], dim=1).unsqueeze(1)      # [F, 1, 3, D]
        return d_conv_w, d_conv_b, d_Wa, d_ba, d_qv
    return weight_part if later else weight_part()


def finish_weight_grads(parts):
    """parts: [(function(dst) -> gradients, the parameter objects they belong to), ...] of one backward call, in its order.  Plain autograd:
    evaluate now, return the gradients (flat tuple).  A trainer with persistent gradient buffers for all of them (ops.inplace_grads): the
    functions queue their results for those buffers (ops.queue_grad: ONE accumulate launch at the end of the backward pass instead of one
    AccumulateGrad add per parameter) and autograd gets None for each -- immediately, or, when the trainer asked for the two-phase backward
    (ops.defer_wgrad), as ONE postponed phase."""
    params = [q for _, ps in parts for q in ps]
    dst = ops.inplace_grads(params)
    if dst is None:
        return tuple(v for fn, _ in parts for v in fn())
    groups, k = [], 0
    for fn, ps in parts:
        groups.append((fn, dst[k:k + len(ps)]))
        k += len(ps)

    def phase2():
        for fn, d_ in groups:
            fn(d_)
    if ops.defer_wgrad:
        ops._deferred.append(phase2)
    else:
        phase2()
    return (None,) * len(params)


def sort_tokens_async(ids_list, num_rows):
    """The token streams as one, sorted on the side stream while the forward kernels run (ops.sort_ids_async).  Streams that already lie back to
    back in one allocation (data_fast.pack_text_streams: the engine's own batches) are taken as they are; anything else is concatenated."""
    flat = [i.reshape(-1) for i in ids_list]
    one = flat[0]
    if len(flat) > 1:
        adjacent = all(a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                       and b.storage_offset() == a.storage_offset() + a.numel() for a, b in zip(flat, flat[1:]))
        one = (torch.as_strided(flat[0], (sum(f.numel() for f in flat),), (1,), flat[0].storage_offset()) if adjacent
               else torch.cat(flat))
    return ops.sort_ids_async(one, num_rows)


def embed_scatter(sorted_pack, n_tokens, dx, table, p, seed):
    """d word_embedding: one segmented reduction over all token streams (rows of dx follow the concatenation order of the ids
    that were sorted at forward time)."""
    lib = _lib()
    dst, d_table = ops.grad_target(table)
    ids_sorted, perm = ops.sorted_ids_ready(sorted_pack)
    assert ids_sorted.numel() == n_tokens
    _call('nr_embed_scatter_sorted', lib.nr_embed_scatter_sorted, _ptr(ids_sorted), _ptr(perm), _ptr(dx), NR_KP, _ptr(dst),
          table.shape[0], n_tokens, p, seed, _stream())
    ops.table_grad_ready(table)
    return d_table


def _sorted_rows_scatter(ids, src, col0, ld, num_rows, pad_row, out=None, sorted_pack=None):
    """dst[id] += sum of the f32 rows src[i, col0:col0+D] with ids[i] == id (ids > pad_row); dst = out (accumulated into) or a fresh zeroed tensor.
    sorted_pack: the ids already sorted on the side stream at forward time (ops.sort_ids_async) -- else they are sorted here, on this stream."""
    dst = torch.zeros(num_rows, NR_D, dtype=torch.float32, device=src.device) if out is None else out
    ids_sorted, perm = ops.sorted_ids_ready(sorted_pack) if sorted_pack is not None else ops.sort_ids(ids, num_rows)
    _call('nr_scatter_sorted_f32', _lib().nr_scatter_sorted_f32, _ptr(ids_sorted), _ptr(perm), src.data_ptr() + col0 * 4, ld, _ptr(dst),
          num_rows, ids.numel(), pad_row, _stream())
    return dst


# ----------------------------------------------------------------------------------------------------------
# NAML news encoder (src/model/NAML/news_encoder.py:50-115)
# ----------------------------------------------------------------------------------------------------------
class _NamlNewsFn(torch.autograd.Function):
    """news vectors f32 [T, D] (+ a bf16 ctx-layout copy for the next pooling level) from title / abstract ids and
    category / subcategory ids.  View order in the stack: title, abstract, category, subcategory (the reference's order
    depends on set iteration; the final attention is permutation invariant)."""

    @staticmethod
    def forward(ctx, title, abstract, cat, sub, table, cat_table,
                cw_t, cb_t, Wa_t, ba_t, qv_t, cw_a, cb_a, Wa_a, ba_a, qv_a,
                W_c, b_c, W_s, b_s, Wa_f, ba_f, qv_f, p, seed, valid_t=None, valid_a=None):
        lib = _lib()
        dev = table.device
        T = title.shape[0]
        need_grad = any(ctx.needs_input_grad)
        views = torch.empty(T * 4, NR_KP, dtype=_BF16_AS_I16, device=dev)
        vstride = 4 * NR_KP
        n_title_tok = title.numel()
        st_t = text_fwd(title, table, cw_t, cb_t, Wa_t, ba_t, qv_t, p, seed, 0, need_grad, None, NR_D, views.data_ptr(), vstride, 'title', valid_t)
        st_a = text_fwd(abstract, table, cw_a, cb_a, Wa_a, ba_a, qv_a, p, seed, n_title_tok, need_grad, None, NR_D,
                        views.data_ptr() + NR_KP * 2, vstride, 'abstract', valid_a)
        ncat, dcat = cat_table.shape
        E = torch.empty(2, ncat, NR_D, dtype=torch.float32, device=dev)
        embf, Wc_, bc_, Ws_, bs_ = _f32c(cat_table), _f32c(W_c), _f32c(b_c), _f32c(W_s), _f32c(b_s)
        _call('nr_element_table_fwd', lib.nr_element_table_fwd, _ptr(embf), ncat, dcat, _ptr(Wc_), _ptr(bc_), _ptr(Ws_), _ptr(bs_), _ptr(E), _stream())
        _call('nr_views_fill', lib.nr_views_fill, _ptr(cat), _ptr(sub), _ptr(E), ncat, _ptr(views), T, _stream())
        Wap, bap, qvp = pack_additive(Wa_f, ba_f, qv_f)
        WaT = ops.pack_additive_t(Wa_f) if (need_grad and not ops.pool_flat_ok(4, False, title.shape[0], qdim=Wa_f.shape[0])) else None
        out = torch.empty(T, NR_D, dtype=torch.float32, device=dev)
        out_b = torch.empty(T, NR_KP, dtype=_BF16_AS_I16, device=dev)
        aw = torch.empty(T, 4, dtype=torch.float32, device=dev)
        _call('nr_additive_fwd[views]', lib.nr_additive_fwd_ex, _ptr(views), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(out), NR_D, _ptr(out_b), NR_KP,
              _ptr(aw), T, 4, _stream())
        if need_grad:
            ctx.save_for_backward(title, abstract, cat, sub, table, embf, Wc_, Ws_, E, views, aw, Wap, bap, qvp, WaT, out)
            ctx.st = (st_t, st_a)
            ctx.meta = (p, seed, Wa_f.shape[0])
            ctx.sorted = sort_tokens_async([title, abstract], table.shape[0]) if ctx.needs_input_grad[4] else None
            # the category / subcategory ids of the element encoders' backward: sorted on the side stream too, off the backward's critical path
            ctx.sorted_elem = (ops.sort_ids_async(cat, ncat), ops.sort_ids_async(sub, ncat))
            ctx.table_param = table                  # the caller's tensor object (the nn.Parameter): ops.grad_target()
            ctx.small_params = (cat_table, W_c, b_c, W_s, b_s, Wa_f, ba_f, qv_f)         # the nn.Parameters: ops.hand_over_grads()
        ctx.mark_non_differentiable(out_b)
        ctx.set_materialize_grads(False)             # the bf16 copy never has a gradient: do not fill 17 MB of zeros for it every step
        return out, out_b

    @staticmethod
    def backward(ctx, g_out, _g_b):
        lib = _lib()
        title, abstract, cat, sub, table, embf, Wc_, Ws_, E, views, aw, Wap, bap, qvp, WaT, y = ctx.saved_tensors
        st_t, st_a = ctx.st
        p, seed, qdim = ctx.meta
        dev = views.device
        T = title.shape[0]
        if g_out is None:
            return (None,) * 27
        g_out = g_out.to(torch.float32).contiguous()
        # final attention over the 4 views
        d_Waf, d_baf, d_qvf, dgemm = _pool_bwd(views, Wap, bap, qvp, aw, g_out, T, 4, qdim, 'views', WaT, y_ptr=_ptr(y), y_stride=y.stride(0))
        gv = _workspace('gviews', (4, T, NR_D), torch.float32, dev)               # view-major: 4 contiguous [T][D] blocks
        _call('nr_additive_dx[views]', lib.nr_additive_dx, _ptr(dgemm), NR_KP, _ptr(aw), _ptr(g_out), _ptr(gv), T, 4, 1, _stream())
        # element encoders: reduce per category row, then the tiny table backward
        ncat, dcat = embf.shape
        dE = torch.zeros(2, ncat, NR_D, dtype=torch.float32, device=dev)
        _sorted_rows_scatter(cat, gv[2], 0, NR_D, ncat, -1, out=dE[0], sorted_pack=ctx.sorted_elem[0])
        _sorted_rows_scatter(sub, gv[3], 0, NR_D, ncat, -1, out=dE[1], sorted_pack=ctx.sorted_elem[1])
        dW = torch.empty(2, NR_D, dcat, dtype=torch.float32, device=dev)
        db = torch.empty(2, NR_D, dtype=torch.float32, device=dev)
        demb = torch.empty(ncat, dcat, dtype=torch.float32, device=dev)
        _call('nr_element_table_bwd', lib.nr_element_table_bwd, _ptr(embf), ncat, dcat, _ptr(Wc_), _ptr(Ws_), _ptr(E), _ptr(dE), _ptr(dW), _ptr(db),
              _ptr(demb), _stream())
        # text encoders; token gradients of both texts land in one buffer -> one embedding scatter
        nt, na = title.numel(), abstract.numel()
        dx = _workspace('dx_tok', (nt + na, NR_KP), _BF16_AS_I16, dev)
        gt = text_bwd(st_t, gv[0], NR_D, p, dx.data_ptr(), 'title', later=True)
        ga = text_bwd(st_a, gv[1], NR_D, p, dx.data_ptr() + nt * NR_KP * 2, 'abstract', later=True)
        d_table = embed_scatter(ctx.sorted, nt + na, dx, ctx.table_param, p, seed) if ctx.needs_input_grad[4] else None
        # the weight gradients of the two text encoders (pooling dWa, conv taps): after the scatter -- under data parallelism while the table flies
        wg = finish_weight_grads([(gt, st_t.params), (ga, st_a.params)])
        ctx.st = None
        small = ops.hand_over_grads(ctx.small_params, (demb, dW[0], db[0], dW[1], db[1], d_Waf, d_baf, d_qvf))
        return (None, None, None, None, d_table, small[0], *wg, *small[1:], None, None, None, None)


def naml_news(title, abstract, cat, sub, table, cat_table, text_t, text_a, elem_c, elem_s, final_att, p_drop, training):
    _require_cuda(table, "word_embedding.weight")
    p = float(p_drop) if training else 0.0
    seed = ops.new_seed() if p > 0 else 0
    a_t, a_a = text_t.additive_attention, text_a.additive_attention
    title, valid_t = pad_text(title, "num_words_title")
    abstract, valid_a = pad_text(abstract, "num_words_abstract")
    return _NamlNewsFn.apply(title, abstract, cat, sub, table, cat_table,
                             text_t.CNN.weight, text_t.CNN.bias, a_t.linear.weight, a_t.linear.bias, a_t.attention_query_vector,
                             text_a.CNN.weight, text_a.CNN.bias, a_a.linear.weight, a_a.linear.bias, a_a.attention_query_vector,
                             elem_c.linear.weight, elem_c.linear.bias, elem_s.linear.weight, elem_s.linear.bias,
                             final_att.linear.weight, final_att.linear.bias, final_att.attention_query_vector, p, seed, valid_t, valid_a)


# ----------------------------------------------------------------------------------------------------------
# additive pooling of dense vectors that already exist as bf16 ctx rows (NAML user encoder, user_encoder.py:18)
# ----------------------------------------------------------------------------------------------------------
class _PoolFn(torch.autograd.Function):
    """out[n, D] = AdditiveAttention(x) for x f32 [n, S, D] whose bf16 ctx-layout copy x_b [n*S][KP] is supplied by the producer
    (no re-quantisation pass); gradient flows to x."""

    @staticmethod
    def forward(ctx, x, x_b, Wa, ba, qv, valid=None):
        lib = _lib()
        n, S, _ = x.shape
        if not lib.nr_supported_pool_len(S):
            raise NotImplementedError(f"sequence length {S} is not instantiated in the pooling kernels (4, 20, 50)")
        valid = S if valid is None else int(valid)
        dev = x.device
        Wap, bap, qvp = pack_additive(Wa, ba, qv)
        out = torch.empty(n, NR_D, dtype=torch.float32, device=dev)
        aw = torch.empty(n, S, dtype=torch.float32, device=dev)
        _call(f'nr_additive_fwd[user S={S}]', lib.nr_additive_fwd_v, _ptr(x_b), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(out), NR_D, None, 0,
              _ptr(aw), n, S, valid, _stream())
        ctx.save_for_backward(x_b, aw, Wap, bap, qvp, None if ops.pool_flat_ok(S, False, n, qdim=Wa.shape[0]) else ops.pack_additive_t(Wa), out)
        ctx.qdim = Wa.shape[0]
        ctx.x_ref = x.detach()                       # address of the input: ops.grad_dst()
        ctx.wparams = (Wa, ba, qv)                   # the nn.Parameters: ops.hand_over_grads()
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        x_b, aw, Wap, bap, qvp, WaT, y = ctx.saved_tensors
        n, S = aw.shape
        g = g.to(torch.float32).contiguous()
        d_Wa, d_ba, d_qv, dgemm = _pool_bwd(x_b, Wap, bap, qvp, aw, g, n, S, ctx.qdim, f'user S={S}', WaT, y_ptr=_ptr(y), y_stride=y.stride(0))
        dx = ops.grad_dst(ctx.x_ref, (n, S, NR_D))
        if dx is None:
            dx = torch.empty(n, S, NR_D, dtype=torch.float32, device=g.device)
        _call('nr_additive_dx[user]', lib.nr_additive_dx, _ptr(dgemm), NR_KP, _ptr(aw), _ptr(g), _ptr(dx), n, S, 0, _stream())
        return (dx, None, *ops.hand_over_grads(ctx.wparams, (d_Wa, d_ba, d_qv)), None)


def pool_rows(x, x_b, additive):
    """AdditiveAttention over the N news vectors of each user (NAML user encoder); N in [1, 50]: padded to 20 / 50 with zero rows that the
    pooling excludes (ops.padded_len)."""
    _require_cuda(x, "clicked_news_vector")
    n, N, d = x.shape
    if not ops.tuned_dims(d, ops.NR_HEADS, additive.linear.weight.shape[0], N):
        return ops.additive_dense(x, additive)
    S = ops.padded_len(N, "num_clicked_news_a_user")
    if S != N:
        x = torch.nn.functional.pad(x, (0, 0, 0, S - N))
        xb = torch.zeros(n, S, NR_KP, dtype=_BF16_AS_I16, device=x_b.device)
        xb[:, :N] = x_b.view(n, N, NR_KP)
        x_b = xb.view(n * S, NR_KP)
    return _PoolFn.apply(x, x_b, additive.linear.weight, additive.linear.bias, additive.attention_query_vector, N)


def to_ctx_rows(x):
    """f32 [n, S, D] -> bf16 ctx-layout rows [n*S][KP] (col D = 1.0); only for callers that do not already hold such a copy
    (evaluate.py hands get_user_vector a freshly stacked f32 tensor)."""
    n, S, d = x.shape
    cb = torch.zeros(n * S, NR_KP, dtype=torch.bfloat16, device=x.device)
    cb[:, :NR_D] = x.detach().reshape(n * S, d).to(torch.bfloat16)
    cb[:, NR_D] = 1.0
    return cb.view(_BF16_AS_I16)


# ----------------------------------------------------------------------------------------------------------
# LSTUR news encoder (src/model/LSTUR/news_encoder.py:32-76): [category row | subcategory row | title vector]
# ----------------------------------------------------------------------------------------------------------
class _LsturNewsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, title, cat, sub, table, cat_table, cw, cb, Wa, ba, qv, p, seed, valid=None):
        lib = _lib()
        dev = table.device
        T = title.shape[0]
        need_grad = any(ctx.needs_input_grad)
        out = torch.empty(T, 3 * NR_D, dtype=torch.float32, device=dev)
        ct = _f32c(cat_table)
        for j, ids in enumerate((cat, sub)):
            _call('nr_gather_rows_strided', lib.nr_gather_rows_strided, _ptr(ids), _ptr(ct), ct.shape[0], NR_D, None, out.data_ptr() + j * NR_D * 4,
                  3 * NR_D, T, _stream())
        st = text_fwd(title, table, cw, cb, Wa, ba, qv, p, seed, 0, need_grad, out.data_ptr() + 2 * NR_D * 4, 3 * NR_D, None, 0, 'title', valid,
                      y_keep=out[:, 2 * NR_D:])
        if need_grad:
            ctx.save_for_backward(title, cat, sub, table)
            ctx.st = st
            ctx.meta = (p, seed, cat_table.shape[0])
            ctx.sorted = sort_tokens_async([title], table.shape[0]) if ctx.needs_input_grad[3] else None
            ctx.sorted_elem = (ops.sort_ids_async(cat, cat_table.shape[0]), ops.sort_ids_async(sub, cat_table.shape[0]))     # side stream, as the tokens
            ctx.table_param = table                  # the caller's tensor object (the nn.Parameter): ops.grad_target()
            ctx.cat_param = cat_table
        return out

    @staticmethod
    def backward(ctx, g):
        title, cat, sub, table = ctx.saved_tensors
        st = ctx.st
        p, seed, ncat = ctx.meta
        dev = g.device
        g = g.to(torch.float32).contiguous()
        T = title.shape[0]
        # category_embedding (padding_idx = 0): two segmented reductions over column blocks of g
        d_cat = _sorted_rows_scatter(cat, g, 0, 3 * NR_D, ncat, 0, sorted_pack=ctx.sorted_elem[0])
        _sorted_rows_scatter(sub, g, NR_D, 3 * NR_D, ncat, 0, out=d_cat, sorted_pack=ctx.sorted_elem[1])
        # the title encoder's third of the [T, 3F] gradient: read in place by the flat pooling backward (no 32 MB contiguous copy per step)
        strided = text_bwd_strided_ok(st)
        g_title = g[:, 2 * NR_D:] if strided else g[:, 2 * NR_D:].contiguous()
        dx = _workspace('dx_tok', (title.numel(), NR_KP), _BF16_AS_I16, dev)
        gt = text_bwd(st, g_title, 3 * NR_D if strided else NR_D, p, dx.data_ptr(), 'title', later=True)
        d_table = embed_scatter(ctx.sorted, title.numel(), dx, ctx.table_param, p, seed) if ctx.needs_input_grad[3] else None
        wg = finish_weight_grads([(gt, st.params)])
        ctx.st = None
        (d_cat,) = ops.hand_over_grads((ctx.cat_param,), (d_cat,))
        return (None, None, None, d_table, d_cat, *wg, None, None, None)


def lstur_news(title, cat, sub, table, cat_table, conv, additive, p_drop, training):
    _require_cuda(table, "word_embedding.weight")
    if not _module_tuned(conv, additive, title.shape[1]):
        # any other geometry (LSTUR/news_encoder.py:52-76): the two category rows and the title vector side by side, from the general kernels
        from . import ops_generic
        return torch.cat([ops_generic._GatherFn.apply(cat, cat_table), ops_generic._GatherFn.apply(sub, cat_table),
                          ops_generic.text_encode(title, table, conv, additive, p_drop, training)], dim=1)
    p = float(p_drop) if training else 0.0
    seed = ops.new_seed() if p > 0 else 0
    title, valid = pad_text(title, "num_words_title")
    return _LsturNewsFn.apply(title, cat, sub, table, cat_table, conv.weight, conv.bias, additive.linear.weight, additive.linear.bias,
                              additive.attention_query_vector, p, seed, valid)


# ----------------------------------------------------------------------------------------------------------
# one text view on its own (NAML TextEncoder.forward used directly)
# ----------------------------------------------------------------------------------------------------------
class _TextFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, cw, cb, Wa, ba, qv, p, seed, valid=None):
        need_grad = any(ctx.needs_input_grad)
        out = torch.empty(ids.shape[0], NR_D, dtype=torch.float32, device=table.device)
        st = text_fwd(ids, table, cw, cb, Wa, ba, qv, p, seed, 0, need_grad, out.data_ptr(), NR_D, None, 0, 'text', valid, y_keep=out)
        if need_grad:
            ctx.save_for_backward(ids, table)
            ctx.st = st
            ctx.meta = (p, seed)
            ctx.sorted = sort_tokens_async([ids], table.shape[0]) if ctx.needs_input_grad[1] else None
            ctx.table_param = table                  # the caller's tensor object (the nn.Parameter): ops.grad_target()
        return out

    @staticmethod
    def backward(ctx, g):
        ids, table = ctx.saved_tensors
        p, seed = ctx.meta
        g = g.to(torch.float32).contiguous()
        dx = _workspace('dx_tok', (ids.numel(), NR_KP), _BF16_AS_I16, g.device)
        gt = text_bwd(ctx.st, g, NR_D, p, dx.data_ptr(), 'text')
        d_table = embed_scatter(ctx.sorted, ids.numel(), dx, ctx.table_param, p, seed) if ctx.needs_input_grad[1] else None
        ctx.st = None
        return (None, d_table, *gt, None, None, None)


def text_only(ids, table, conv, additive, p_drop, training):
    _require_cuda(table, "word_embedding.weight")
    if not _module_tuned(conv, additive, ids.shape[1]):
        from . import ops_generic
        return ops_generic.text_encode(ids, table, conv, additive, p_drop, training)
    p = float(p_drop) if training else 0.0
    seed = ops.new_seed() if p > 0 else 0
    ids, valid = pad_text(ids, "text length")
    return _TextFn.apply(ids, table, conv.weight, conv.bias, additive.linear.weight, additive.linear.bias,
                         additive.attention_query_vector, p, seed, valid)


# ----------------------------------------------------------------------------------------------------------
# one element view on its own (NAML ElementEncoder.forward used directly, src/model/NAML/news_encoder.py:40-47)
# ----------------------------------------------------------------------------------------------------------
class _ElementFn(torch.autograd.Function):
    """relu(linear(embedding(ids))): the table form (one row of E per category id, nr_element_table_fwd) + a row gather."""

    @staticmethod
    def forward(ctx, ids, emb, W, b):
        lib = _lib()
        dev = emb.device
        ncat, dcat = emb.shape
        F = W.shape[0]
        if F != NR_D:
            raise NotImplementedError(f"ElementEncoder output dim must be {NR_D} (got {F})")
        embf, Wf, bf = _f32c(emb), _f32c(W), _f32c(b)
        E = torch.empty(2, ncat, NR_D, dtype=torch.float32, device=dev)
        _call('nr_element_table_fwd', lib.nr_element_table_fwd, _ptr(embf), ncat, dcat, _ptr(Wf), _ptr(bf), _ptr(Wf), _ptr(bf), _ptr(E), _stream())
        flat = ids.reshape(-1).contiguous()
        out = torch.empty(flat.numel(), NR_D, dtype=torch.float32, device=dev)
        _call('nr_gather_rows_strided', lib.nr_gather_rows_strided, _ptr(flat), _ptr(E), ncat, NR_D, None, _ptr(out), NR_D, flat.numel(), _stream())
        ctx.save_for_backward(flat, embf, Wf, E)
        ctx.shape = tuple(ids.shape)
        return out.view(*ids.shape, NR_D)

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        flat, embf, Wf, E = ctx.saved_tensors
        ncat, dcat = embf.shape
        dev = embf.device
        g = g.reshape(-1, NR_D).to(torch.float32).contiguous()
        dE = torch.stack([_sorted_rows_scatter(flat, g, 0, NR_D, ncat, -1), torch.zeros(ncat, NR_D, dtype=torch.float32, device=dev)])
        dW = torch.empty(2, NR_D, dcat, dtype=torch.float32, device=dev)
        db = torch.empty(2, NR_D, dtype=torch.float32, device=dev)
        demb = torch.empty(ncat, dcat, dtype=torch.float32, device=dev)
        _call('nr_element_table_bwd', lib.nr_element_table_bwd, _ptr(embf), ncat, dcat, _ptr(Wf), _ptr(Wf), _ptr(E), _ptr(dE), _ptr(dW), _ptr(db),
              _ptr(demb), _stream())
        return None, demb, dW[0], db[0]


def element_only(ids, embedding, linear):
    _require_cuda(embedding.weight, "category embedding")
    ops.check_ids(ids, embedding.weight.shape[0], "category / subcategory id")
    if linear.weight.shape[0] != NR_D:
        from . import ops_generic
        return ops_generic.element_encode(ids, embedding, linear)
    return _ElementFn.apply(ids, embedding.weight, linear.weight, linear.bias)
