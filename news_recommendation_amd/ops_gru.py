"""Host wrappers (autograd) of the LSTUR user encoder: user_embedding row gather with whole-row masking
(src/model/LSTUR/__init__.py:38-42,74-77) and the GRU over the click history (src/model/LSTUR/user_encoder.py:16-45).

The recurrence runs in the per-step HIP kernels (csrc/k_gru.h), queued back to back on the current stream; the
time-independent products around it are plain bf16 GEMMs: the hoisted input projection x W_ih^T, and in the backward
dX = dGi W_ih, dW_ih = dGi^T X, dW_hh = dGh^T H.  No CPU path.
"""
import ctypes
import os
import torch

from . import ops
from .ops import _lib, _stream, _call, _ptr, _f32c, _workspace, _require_cuda, _BF16_AS_I16


def gru_dims(Hd):
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ops._ck(_lib().nr_gru_dims(Hd, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def _ceil(v, m):
    return (v + m - 1) // m * m


def rows_to_bf16(src, d, dp):
    """f32 [n, >=d] (row stride = src.stride(0)) -> bf16 [n, dp] with col d = 1.0."""
    n = src.shape[0]
    dst = torch.empty(n, dp, dtype=_BF16_AS_I16, device=src.device)
    _call('nr_rows_to_bf16', _lib().nr_rows_to_bf16, _ptr(src), src.stride(0), d, _ptr(dst), dp, n, _stream())
    return dst


class _UserRowsFn(torch.autograd.Function):
    """out[b] = row_scale[b] * table[ids[b]]: nn.Embedding(padding_idx=0) forward fused with F.dropout2d's per-sample factor
    keep_b / (1 - p) (row_scale = None in eval mode); backward scatters B rows into the dense table gradient, row 0 skipped."""

    @staticmethod
    def forward(ctx, ids, table, row_scale):
        sync = getattr(table, '_nr_row_sync', None)
        if sync is not None:
            sync(ids)                            # row-sparse optimiser (optim.EngineAdam): replay the idle Adam steps these rows missed
        tab = _f32c(table)
        B, d = ids.shape[0], tab.shape[1]
        out = torch.empty(B, d, dtype=torch.float32, device=tab.device)
        _call('nr_gather_rows_strided[user]', _lib().nr_gather_rows_strided, _ptr(ids), _ptr(tab), tab.shape[0], d, _ptr(row_scale), _ptr(out), d, B, _stream())
        ctx.save_for_backward(ids, row_scale)
        ctx.shape = tuple(tab.shape)
        ctx.table_param = table                  # the caller's tensor object (the nn.Parameter): ops.grad_target()
        return out

    @staticmethod
    def backward(ctx, g):
        ids, row_scale = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        sink = getattr(ctx.table_param, '_nr_row_sink', None)
        if sink is not None:                     # row-sparse optimiser: hand over (ids, gradient rows), never build the dense table gradient
            sink(ids, g if row_scale is None else g * row_scale.unsqueeze(1))
            return None, None, None
        dst, d_table = ops.grad_target(ctx.table_param)
        _call('nr_rows_scatter_add[user]', _lib().nr_rows_scatter_add, _ptr(ids), _ptr(g), g.shape[1], _ptr(row_scale), _ptr(dst), ctx.shape[0],
              ctx.shape[1], ids.shape[0], 0, _stream())
        return None, d_table, None


def user_rows(ids, table, row_scale=None):
    _require_cuda(table, "user_embedding.weight")
    return _UserRowsFn.apply(ids.contiguous(), table, row_scale)


def _wih_t(Wih_p, Hg, Ip, Kp):
    """W_ih^T as the K-contiguous B operand of dX = dGi W_ih: bf16 [Ip][Kp] (rows = input features, columns = the 3 Hg gate rows, K padding
    zero).  Cached like every other packed operand (ops._packed, keyed on the packed W_ih it derives from -- a new tensor per parameter state --
    per device and parameter, dropped by ops.invalidate_packed())."""
    def build():
        WihT = torch.zeros(Ip, Kp, dtype=_BF16_AS_I16, device=Wih_p.device)
        _call('nr_transpose_bf16', _lib().nr_transpose_bf16, _ptr(Wih_p), 3 * Hg, Ip, Ip, _ptr(WihT), Kp, _stream())
        return WihT
    return ops._packed('gru_wih_t', (Wih_p,), build)


# The state history H_all bf16 [T + 1][B][Hp] (48 MB at B = 512, N = 50, Hd = 900) is written by the sweep in every column <= Hd of every row,
# every call; only the padding columns rely on a zero fill: ops.step_buffer hands the previous step's buffer out again (no 48 MB fill per step).


class _GruFn(torch.autograd.Function):
    """h_last[b] = GRU state after consuming x[b, 0:len[b]] starting from h0[b] (len >= 1)."""

    @staticmethod
    def forward(ctx, x, h0, lens_dev, T, W_ih, W_hh, b_ih, b_hh):
        lib = _lib()
        B, N, I = x.shape
        Hd = W_hh.shape[1]
        Hg, Hp, Kp = gru_dims(Hd)
        Ip = _ceil(I + 1, 32)
        dev = x.device
        need_grad = any(ctx.needs_input_grad)
        def build():         # packed once per parameter state (ops._packed), not once per call
            Wih_p = torch.empty(3 * Hg, Ip, dtype=_BF16_AS_I16, device=dev)
            Whh_p = torch.empty(3 * Hg, Hp, dtype=_BF16_AS_I16, device=dev)
            WhhT = torch.empty(Hp, Kp, dtype=_BF16_AS_I16, device=dev)
            Wi, Wh = _f32c(W_ih), _f32c(W_hh)
            _call('nr_pack_gru', lib.nr_pack_gru, _ptr(Wi), Hd, I, Ip, _ptr(Wih_p), None, 0, _stream())            # row-major: GEMM operand
            _call('nr_pack_gru', lib.nr_pack_gru, _ptr(Wh), Hd, Hd, Hp, _ptr(Whh_p), _ptr(WhhT), 1, _stream())     # tile order: step-kernel operands
            return Wih_p, Whh_p, WhhT
        Wih_p, Whh_p, WhhT = ops._packed('gru', (W_ih, W_hh), build)
        bi, bh = _f32c(b_ih), _f32c(b_hh)
        xf = _f32c(x).view(B * N, I)
        Xb = rows_to_bf16(xf, I, Ip)                                                         # [B*N][Ip], col I = 1.0
        # hoisted input projection [B*N][3*Hg] f32, hand-written NT kernel (csrc/k_gemm.h): K = Ip (column I of Xb is 1.0, of W_ih the padding 0)
        gi = ops.gemm_nt(Xb, Wih_p, B * N, 3 * Hg, Ip, 'nr_gemm_nt_gru_gi')
        H_all = (ops.step_buffer('gru_H_all', (T + 1, B, Hp), _BF16_AS_I16, dev, lambda t: t.zero_()) if need_grad
                 else torch.zeros(1, B, Hp, dtype=_BF16_AS_I16, device=dev))
        hf = torch.zeros(2, B, Hp, dtype=torch.float32, device=dev)
        if h0 is not None:
            hf[0][:, :Hd].copy_(h0)
        _call('nr_rows_to_bf16', lib.nr_rows_to_bf16, _ptr(hf[0]), Hp, Hd, _ptr(H_all[0]), Hp, B, _stream())
        gates = torch.empty(T, B, 4, Hg, dtype=_BF16_AS_I16, device=dev) if need_grad else None
        # step-to-step operand, tile order: a ping-pong pair, or one buffer per step when the shape gets the persistent kernel (the rows /
        # columns of padding are never written, so the zero fill of the cached workspace is done once)
        nbuf = lib.nr_gru_seq_buffers(B, Hd, N)                      # sized for the longest history, so that the shape does not change with T
        ht = _workspace('gru_h_t', (nbuf, _ceil(B, 16) * Hp), _BF16_AS_I16, dev, zero=True)
        _call('nr_tile_rows_bf16', lib.nr_tile_rows_bf16, _ptr(H_all[0]), B, Hp, _ptr(ht[0]), _stream())
        # the whole recurrence from one C call (on MI355X: one persistent launch, csrc/k_gru_persist.h; elsewhere T step launches); a profile
        # records it as ONE entry -- the per-step entry points remain for the kernel tests and tools/prof_gru.py
        ops.seq_launches['nr_gru_fwd_seq'] = T
        _call('nr_gru_fwd_seq', lib.nr_gru_fwd_seq_n, _ptr(gi), _ptr(Whh_p), _ptr(bi), _ptr(bh), _ptr(lens_dev), _ptr(ht), nbuf,
              _ptr(H_all) if need_grad else None, _ptr(hf), _ptr(gates) if need_grad else None, B, N, Hd, T, _stream())
        out = hf[T % 2][:, :Hd].contiguous()
        if need_grad:
            ctx.save_for_backward(Xb, H_all, gates, lens_dev, Wih_p, WhhT)
            ctx.meta = (B, N, I, Hd, T, h0 is not None)
            ctx.x_ref = x.detach()                           # address of the input: ops.grad_dst()
            ctx.wparams = (W_ih, W_hh, b_ih, b_hh)           # the nn.Parameters: ops.inplace_grads()
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        Xb, H_all, gates, lens_dev, Wih_p, WhhT = ctx.saved_tensors
        B, N, I, Hd, T, has_h0 = ctx.meta
        Hg, Hp, Kp = gru_dims(Hd)
        Ip = Xb.shape[1]
        dev = g.device
        g = g.to(torch.float32).contiguous()
        dgi = _workspace('gru_dgi', (B * N, Kp), _BF16_AS_I16, dev, zero=True)                # K padding stays zero
        if T < N:
            dgi.view(B, N, Kp)[:, T:].zero_()                                                # steps nobody reached
        dgh = _workspace('gru_dgh', (T, B, Kp), _BF16_AS_I16, dev, zero=True)
        carry = torch.empty(2, B, Hp, dtype=torch.float32, device=dev)
        nbuf = lib.nr_gru_seq_buffers(B, Hd, N)
        dght = _workspace('gru_dgh_t', (nbuf, _ceil(B, 16) * Kp), _BF16_AS_I16, dev, zero=True)   # step-to-step operand, tile order (see forward)
        ops.seq_launches['nr_gru_bwd_seq'] = T + 1
        _call('nr_gru_bwd_seq', lib.nr_gru_bwd_seq_n, _ptr(g), _ptr(WhhT), _ptr(gates), _ptr(H_all), _ptr(lens_dev), _ptr(dgi), _ptr(dgh),
              _ptr(dght), nbuf, _ptr(carry), B, N, Hd, T, _stream())
        d_h0 = carry[T % 2][:, :Hd].contiguous() if has_h0 else None
        # the three time-independent products of the backward in the hand-written ring kernels (csrc/k_gemm.h): dX = dGi W_ih as an NT product
        # against W_ih^T (re-packed once per optimiser step, K padding zero), fp32 result in place of a bf16 one + conversion pass; the two
        # weight gradients as split-K TN products over the (sample, step) rows, partials summed in a fixed order
        d_x = None
        if ctx.needs_input_grad[0]:
            WihT = _wih_t(Wih_p, Hg, Ip, Kp)
            # when x is the history part of split_rows(): straight into its half of the shared gradient buffer
            d_x = ops.gemm_nt(dgi, WihT, B * N, I, Kp, 'nr_gemm_nt_gru_dx', out=ops.grad_dst(ctx.x_ref, (B * N, I))).view(B, N, I)
        dWi = ops.sum_parts(ops.gemm_tn_parts(dgi, Kp, Xb, Ip, 'nr_gemm_tn_gru_dWih'))                      # [Kp][Ip]; col I = bias gradient
        dWh = ops.sum_parts(ops.gemm_tn_parts(dgh.view(T * B, Kp), Kp, H_all.view((T + 1) * B, Hp), Hp, 'nr_gemm_tn_gru_dWhh', n_tok=T * B))      # [Kp][Hp]; col Hd = bias gradient
        ops.step_buffer_release(H_all)                       # last reader on the stream: the next forward may overwrite it
        dst = ops.inplace_grads(ctx.wparams) if all(ctx.needs_input_grad[4:8]) else None
        if dst is not None:
            # the trainer's persistent buffers: gate q of each gradient is a row block of the padded product -- twelve strided items of the
            # backward pass's ONE accumulate launch (ops.queue_grad) instead of four concatenations and four AccumulateGrad adds
            for q in range(3):
                rows = slice(q * Hg, q * Hg + Hd)
                ops.queue_grad(dst[0][q * Hd:(q + 1) * Hd], dWi[rows, :I])
                ops.queue_grad(dst[1][q * Hd:(q + 1) * Hd], dWh[rows, :Hd])
                ops.queue_grad(dst[2][q * Hd:(q + 1) * Hd], dWi[rows, I])
                ops.queue_grad(dst[3][q * Hd:(q + 1) * Hd], dWh[rows, Hd])
            return (d_x, d_h0, None, None, None, None, None, None)
        unpad = lambda m, ncol: torch.cat([m[q * Hg:q * Hg + Hd, :ncol] for q in range(3)], dim=0)
        return (d_x, d_h0, None, None, unpad(dWi, I), unpad(dWh, Hd), unpad(dWi, I + 1)[:, I].contiguous(), unpad(dWh, Hd + 1)[:, Hd].contiguous())


def persist_status():
    """(forward, backward) STICKY error bits of the persistent GRU sweeps since the last fault_clear() (csrc/k_gru_persist.h; nr_gru_persist_status):
    0 = clean.  SYNCHRONISES the device -- call it where the host waits anyway (end of a timed region, a logging step, a test), never inside a
    stream capture."""
    f, b = ctypes.c_int32(0), ctypes.c_int32(0)
    _call('nr_gru_persist_status', _lib().nr_gru_persist_status, ctypes.byref(f), ctypes.byref(b))
    return f.value, b.value


def fault_state():
    """The four fault words of the process (include/nr_engine.h): (forward bits, backward bits, first skipped optimiser step, spare).  While the
    first two are non-zero the optimiser kernels apply no update.  SYNCHRONISES."""
    w = (ctypes.c_uint32 * 4)()
    _call('nr_fault_state', _lib().nr_fault_state, w)
    return tuple(int(x) for x in w)


def fault_clear():
    _call('nr_fault_clear', _lib().nr_fault_clear)


def persist_check():
    """Raises when a persistent sweep since the last fault_clear() gave up a wait or found its XCD team wrong: its outputs were garbage, and
    every optimiser step from that one on has been skipped (the parameters are those of the last good step).  Callers that can repeat steps
    (train_fast.py) handle the words themselves; everybody else stops here.  The step-per-launch form (NR_GRU_PERSIST=0) has no such failure mode."""
    st = fault_state()
    if st[0] or st[1]:
        raise RuntimeError(f'persistent GRU sweep failed (sticky error bits forward / backward = {st[:2]}, optimiser steps skipped from step {st[2]}): '
                           'results since then are invalid; run with NR_GRU_PERSIST=0')


def gru_last_state(x, h0, clicked_news_length, gru):
    """x f32 [B, N, I] on the GPU, h0 f32 [B, Hd] or None (zeros), clicked_news_length: CPU (or device) integer tensor, already >= 1."""
    _require_cuda(x, "clicked_news_vector")
    N = x.shape[1]
    if clicked_news_length.is_cuda:
        # lengths already on the device: run all N steps (finished samples keep their state, so the result is the same) rather
        # than fetch max(length) -- a device-to-host round trip in the middle of the step drains the stream
        T = N
        lens_dev = clicked_news_length.detach().clamp(min=1, max=N).to(torch.int32)
    else:
        lens = clicked_news_length.detach().clamp(min=1, max=N)
        T = int(lens.max())
        lens_dev = ops.to_device_async(lens.to(torch.int32), x.device)
    return _GruFn.apply(x, h0, lens_dev, T, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)


# from this many histories on, gru_last_state_rows runs the recurrent product as a library GEMM per step (NR_GRU_GEMM_MIN_B; 0 = never)
_GEMM_STEP_MIN_B = int(os.environ.get('NR_GRU_GEMM_MIN_B', '8192')) or (1 << 62)


@torch.no_grad()
def gru_last_state_rows(table, rows, h0, lengths, gru):
    """Inference form of gru_last_state for histories that index a table: x[b, t] = table[rows[b, t]] (table f32 [R, I] on the GPU, rows integer
    [B, N], lengths host integers [B], already >= 1).  x_t W_ih^T is computed once per TABLE row (one GEMM over R rows instead of B * N),
    the step kernels fetch it through the index (nr_gru_fwd_seq_rows), and the histories are processed longest first so that step t only
    launches the rows that are still running -- pack_padded_sequence's own schedule (src/model/LSTUR/user_encoder.py:27-45)."""
    import numpy as np
    lib = _lib()
    _require_cuda(table, "news vectors")
    dev = table.device
    B, N = rows.shape
    I = table.shape[1]
    W_ih, W_hh = gru.weight_ih_l0, gru.weight_hh_l0
    Hd = W_hh.shape[1]
    Hg, Hp, Kp = gru_dims(Hd)
    Ip = _ceil(I + 1, 32)
    if B == 0:
        return torch.zeros(0, Hd, device=dev)

    def build():
        Wih_p = torch.empty(3 * Hg, Ip, dtype=_BF16_AS_I16, device=dev)
        Whh_p = torch.empty(3 * Hg, Hp, dtype=_BF16_AS_I16, device=dev)
        WhhT = torch.empty(Hp, Kp, dtype=_BF16_AS_I16, device=dev)
        Wi, Wh = _f32c(W_ih), _f32c(W_hh)
        _call('nr_pack_gru', lib.nr_pack_gru, _ptr(Wi), Hd, I, Ip, _ptr(Wih_p), None, 0, _stream())
        _call('nr_pack_gru', lib.nr_pack_gru, _ptr(Wh), Hd, Hd, Hp, _ptr(Whh_p), _ptr(WhhT), 1, _stream())
        return Wih_p, Whh_p, WhhT
    Wih_p, Whh_p, _ = ops._packed('gru', (W_ih, W_hh), build)
    bi, bh = _f32c(gru.bias_ih_l0), _f32c(gru.bias_hh_l0)
    Xb = rows_to_bf16(_f32c(table), I, Ip)
    gi = ops.gemm_nt(Xb, Wih_p, Xb.shape[0], 3 * Hg, Ip, 'nr_gemm_nt_gru_gi[table]')         # [R][3*Hg] f32, hand-written NT kernel
    lens = np.clip(np.asarray(lengths, dtype=np.int64), 1, N)
    order = np.argsort(-lens, kind='stable')                                                 # longest first
    T = int(lens[order[0]])
    active = np.ascontiguousarray((lens[order][None, :] > np.arange(T)[:, None]).sum(axis=1).astype(np.int32))
    order_d = torch.from_numpy(order).to(dev)
    rows_s = rows.to(dev)[order_d].to(torch.int32).contiguous()
    lens_s = torch.from_numpy(lens[order].astype(np.int32)).to(dev)
    hf = torch.zeros(B, Hp, dtype=torch.float32, device=dev)
    if h0 is not None:
        hf[:, :Hd].copy_(h0[order_d])
    ht = torch.zeros(2, _ceil(B, 16) * Hp, dtype=_BF16_AS_I16, device=dev)
    hb = torch.empty(B, Hp, dtype=_BF16_AS_I16, device=dev)
    _call('nr_rows_to_bf16', lib.nr_rows_to_bf16, _ptr(hf), Hp, Hd, _ptr(hb), Hp, B, _stream())
    if B >= _GEMM_STEP_MIN_B:
        # large batches: the recurrent product as one NT GEMM per step (nr_gemm_nt) + the gate kernel (csrc/k_gru.h gru_gate_rows_kernel): the fused
        # step kernel's 16-unit workgroups would each re-read the state rows (57 x per step)
        def build_rm():
            Whh_rm = torch.empty(3 * Hg, Hp, dtype=_BF16_AS_I16, device=dev)
            _call('nr_pack_gru', lib.nr_pack_gru, _ptr(_f32c(W_hh)), Hd, Hd, Hp, _ptr(Whh_rm), None, 0, _stream())
            return (Whh_rm,)
        (Whh_rm,) = ops._packed('gru_rm', (W_hh,), build_rm)
        gh = torch.empty(B, 3 * Hg, dtype=torch.float32, device=dev)
        for t in range(T):
            Bt = int(active[t])
            if Bt == 0:
                break
            ops.gemm_nt(hb, Whh_rm, Bt, 3 * Hg, Hp, 'nr_gemm_nt_gru_gh', out=gh, ldc=3 * Hg)          # hand-written NT kernel, fp32 result
            _call('nr_gru_gate_rows', lib.nr_gru_gate_rows, _ptr(gi), _ptr(rows_s), _ptr(gh), _ptr(bi), _ptr(bh), _ptr(lens_s), _ptr(hf), _ptr(hb),
                  Bt, N, Hd, t, _stream())
    else:
        _call('nr_tile_rows_bf16', lib.nr_tile_rows_bf16, _ptr(hb), B, Hp, _ptr(ht[0]), _stream())
        _call('nr_gru_fwd_seq_rows', lib.nr_gru_fwd_seq_rows, _ptr(gi), _ptr(rows_s), _ptr(Whh_p), _ptr(bi), _ptr(bh), _ptr(lens_s), _ptr(ht), _ptr(hf),
              active.ctypes.data, B, N, Hd, T, _stream())
    out = torch.empty(B, Hd, dtype=torch.float32, device=dev)
    out[order_d] = hf[:, :Hd]
    return out
