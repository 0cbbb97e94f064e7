"""Parity checks of the GRU step kernels (LSTUR user encoder) through the C-ABI, emulator or cuda:0.
The per-step launches are driven exactly as the host layer drives them; the plain GEMMs around them (input projection,
weight gradients) are done in numpy here.  Reference: the oracle's explicit recurrence (oracle/lstur_torch.py) in float64."""
import ctypes
import numpy as np
import torch

from tests.backends import bf16_to_f32, f32_to_bf16, bf16_round
from tests.kernel_checks import untile, ck
from oracle.lstur_torch import OracleLSTURUserEncoder


def dims(be, Hd):
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ck(be, be.lib.nr_gru_dims(Hd, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def gate_pad(a, Hd, Hg):
    """[..., 3*Hd] -> [..., 3*Hg] (gate q of unit j at q*Hg + j)."""
    out = np.zeros(a.shape[:-1] + (3 * Hg,), dtype=a.dtype)
    for q in range(3):
        out[..., q * Hg:q * Hg + Hd] = a[..., q * Hd:(q + 1) * Hd]
    return out


def gate_unpad(a, Hd, Hg):
    return np.concatenate([a[..., q * Hg:q * Hg + Hd] for q in range(3)], axis=-1)


def persistent_bwd_active(be, B, Hd, T):
    """True when nr_gru_bwd_seq takes the persistent XCD-local form (csrc/k_gru_persist.h) for this shape on this backend: its contraction is
    split over eight waves (a different summation order from the step kernels), so the sweep equals the step launches to rounding, not bit
    for bit.  The forward persistent form keeps the step kernels' summation order and IS compared bit for bit."""
    import os
    if be.name != 'gpu' or not (int(os.environ.get('NR_GRU_PERSIST', '3')) & 2):
        return False
    return B <= 512 and Hd in (900, 450) and T >= 1 and be.torch.cuda.get_device_properties(0).multi_processor_count == 256


def same_sweep(a, b, exact, what):
    """bf16 (uint16) or f32 arrays of the two forms of a sweep: identical, or (persistent backward) equal to 2 % of the tensor's scale with
    at most rounding-sized differences in the median."""
    if exact:
        assert np.array_equal(a, b), what
        return
    fa = bf16_to_f32(a) if a.dtype == np.uint16 else a.astype(np.float32)
    fb = bf16_to_f32(b) if b.dtype == np.uint16 else b.astype(np.float32)
    d = np.abs(fa.astype(np.float64) - fb)
    scale = np.abs(fb).max() + 1e-30
    assert d.max() <= 2e-2 * scale, f'{what}: max diff {d.max():.3g} on scale {scale:.3g}'
    assert np.median(d) <= 1e-3 * scale, f'{what}: median diff {np.median(d):.3g} on scale {scale:.3g}'


def check_gru(be, B=5, N=4, Hd=900, I=900, seed=0, lens=None):
    rng = np.random.default_rng(seed)
    Hg, Hp, Kp = dims(be, Hd)
    assert Hg % 16 == 0 and Hp % 32 == 0 and Hp > Hd and Kp >= 3 * Hg
    k = 1.0 / np.sqrt(Hd)
    W_ih = rng.uniform(-k, k, size=(3 * Hd, I)).astype(np.float32)
    W_hh = rng.uniform(-k, k, size=(3 * Hd, Hd)).astype(np.float32)
    b_ih = rng.uniform(-k, k, size=3 * Hd).astype(np.float32)
    b_hh = rng.uniform(-k, k, size=3 * Hd).astype(np.float32)
    x = rng.normal(0, 0.5, size=(B, N, I)).astype(np.float32)
    h0 = rng.normal(0, 0.5, size=(B, Hd)).astype(np.float32)
    lens = np.asarray(lens if lens is not None else rng.integers(1, N + 1, size=B), dtype=np.int32)
    lens[0] = N
    T = int(lens.max())
    # ---- operand packing through the library ----------------------------------------------------------------------
    Whh_p = be.poison((3 * Hg, Hp), np.uint16)
    WhhT_p = be.poison((Hp, Kp), np.uint16)
    Whh_rm, WhhT_rm = be.poison((3 * Hg, Hp), np.uint16), be.poison((Hp, Kp), np.uint16)
    ck(be, be.lib.nr_pack_gru(be.ptr(be.dev(W_hh)), Hd, Hd, Hp, be.ptr(Whh_rm), be.ptr(WhhT_rm), 0, be.stream))
    ck(be, be.lib.nr_pack_gru(be.ptr(be.dev(W_hh)), Hd, Hd, Hp, be.ptr(Whh_p), be.ptr(WhhT_p), 1, be.stream))
    be.sync()
    Wq = bf16_round(W_hh)
    got = bf16_to_f32(be.np(Whh_rm)).reshape(3, Hg, Hp)
    assert np.array_equal(got[:, :Hd, :Hd], Wq.reshape(3, Hd, Hd)) and not got[:, Hd:].any() and not got[:, :, Hd:].any()
    assert np.array_equal(untile(be.np(Whh_p), 3 * Hg, Hp), be.np(Whh_rm)) and np.array_equal(untile(be.np(WhhT_p), Hp, Kp), be.np(WhhT_rm))
    gotT = bf16_to_f32(be.np(WhhT_rm))
    assert np.array_equal(gate_unpad(gotT[:Hd, :3 * Hg], Hd, Hg), Wq.T) and not gotT[Hd:].any() and not gotT[:, 3 * Hg:].any()
    # ---- forward sweep ------------------------------------------------------------------------------------------------
    gi = (bf16_round(x).astype(np.float64).reshape(B * N, I) @ bf16_round(W_ih).astype(np.float64).T).astype(np.float32)
    h_gi = be.dev(gate_pad(gi, Hd, Hg))
    hb_ih, hb_hh, hlen = be.dev(b_ih), be.dev(b_hh), be.dev(lens)
    H_all = [be.empty((B, Hp), np.uint16) for _ in range(T + 1)]
    h0p = np.zeros((B, Hp), dtype=np.float32); h0p[:, :Hd] = h0
    hf = [be.dev(h0p), be.empty((B, Hp), np.float32)]
    ck(be, be.lib.nr_rows_to_bf16(be.ptr(hf[0]), Hp, Hd, be.ptr(H_all[0]), Hp, B, be.stream))
    gates = [be.poison((B, 4, Hg), np.uint16) for _ in range(T)]
    B16 = (B + 15) // 16 * 16
    ht = [be.poison((B16, Hp), np.uint16), be.dev(np.zeros((B16, Hp), dtype=np.uint16))]
    ck(be, be.lib.nr_tile_rows_bf16(be.ptr(H_all[0]), B, Hp, be.ptr(ht[0]), be.stream))
    be.sync()
    t0 = untile(be.np(ht[0]), B16, Hp)
    assert np.array_equal(t0[:B], be.np(H_all[0])) and not t0[B:].any()
    for t in range(T):
        ck(be, be.lib.nr_gru_fwd_step(be.ptr(h_gi), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen), be.ptr(ht[t % 2]),
                                      be.ptr(H_all[t + 1]), be.ptr(ht[(t + 1) % 2]), be.ptr(hf[t % 2]), be.ptr(hf[(t + 1) % 2]), be.ptr(gates[t]),
                                      B, N, Hd, t, be.stream))
    be.sync()
    assert np.array_equal(untile(be.np(ht[T % 2]), B16, Hp)[:B], be.np(H_all[T]))          # both forms of h_T agree
    # the whole recurrence in one call (nr_gru_fwd_seq) = the same launches: bit-identical state, saved states and gates
    ht2 = np.zeros((2, B16, Hp), dtype=np.uint16); ht2[0] = be.np(ht[0]) if T % 2 == 0 else 0
    ht2_d = be.dev(ht2)
    ck(be, be.lib.nr_tile_rows_bf16(be.ptr(H_all[0]), B, Hp, be.ptr(ht2_d), be.stream))
    hf2 = np.zeros((2, B, Hp), dtype=np.float32); hf2[0] = h0p
    hf2_d = be.dev(hf2)
    Hs = np.zeros((T + 1, B, Hp), dtype=np.uint16); Hs[0] = be.np(H_all[0])
    Hs_d = be.dev(Hs)
    gs_d = be.poison((T, B, 4, Hg), np.uint16)
    ck(be, be.lib.nr_gru_fwd_seq(be.ptr(h_gi), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen), be.ptr(ht2_d), be.ptr(Hs_d), be.ptr(hf2_d),
                                 be.ptr(gs_d), B, N, Hd, T, be.stream))
    be.sync()
    assert np.array_equal(be.np(hf2_d)[T % 2], be.np(hf[T % 2]))
    assert all(np.array_equal(be.np(Hs_d)[t], be.np(H_all[t])) for t in range(T + 1))
    assert all(np.array_equal(be.np(gs_d)[t], be.np(gates[t])) for t in range(T))
    # ... and through the n-buffer entry point (nr_gru_fwd_seq_n; n_buf = 2 since the persistent form was removed: the ping-pong pair)
    nbuf = be.lib.nr_gru_seq_buffers(B, Hd, T)
    assert nbuf == 2
    htn = np.zeros((nbuf, B16, Hp), dtype=np.uint16)
    htn_d = be.dev(htn)
    ck(be, be.lib.nr_tile_rows_bf16(be.ptr(H_all[0]), B, Hp, be.ptr(htn_d), be.stream))
    hfn_d = be.dev(hf2)
    Hsn_d = be.dev(Hs)
    gsn_d = be.poison((T, B, 4, Hg), np.uint16)
    ck(be, be.lib.nr_gru_fwd_seq_n(be.ptr(h_gi), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen), be.ptr(htn_d), nbuf, be.ptr(Hsn_d),
                                   be.ptr(hfn_d), be.ptr(gsn_d), B, N, Hd, T, be.stream))
    be.sync()
    assert np.array_equal(be.np(hfn_d)[T % 2], be.np(hf[T % 2]))
    assert all(np.array_equal(be.np(Hsn_d)[t], be.np(H_all[t])) for t in range(T + 1))
    assert all(np.array_equal(be.np(gsn_d)[t], be.np(gates[t])) for t in range(T))
    assert np.array_equal(untile(be.np(htn_d)[T % nbuf if nbuf == 2 else T], B16, Hp)[:B], be.np(H_all[T]))
    h_last = be.np(hf[T % 2])[:, :Hd]
    # reference: the oracle recurrence in float64 on the same operands
    enc = OracleLSTURUserEncoder(Hd // 3 if Hd % 3 == 0 and I == Hd else 1, 'ini')
    enc.gru.dh = Hd
    to = lambda a: torch.nn.Parameter(torch.from_numpy(np.asarray(a, dtype=np.float64)))
    enc.gru.weight_ih_l0, enc.gru.weight_hh_l0, enc.gru.bias_ih_l0, enc.gru.bias_hh_l0 = to(W_ih), to(W_hh), to(b_ih), to(b_hh)
    xt = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    h0t = torch.from_numpy(h0.astype(np.float64)).requires_grad_(True)
    ref = enc(h0t, torch.from_numpy(lens.astype(np.int64)), xt)
    err = np.abs(h_last - ref.detach().numpy()).max()
    assert err < 2e-2, f'gru fwd: max err {err:.3g}'          # bf16 operands, up to N recurrent steps
    hb = be.np(H_all[T])
    assert np.array_equal(hb[:, :Hd], f32_to_bf16(h_last)) and (hb[:, Hd] == 0x3F80).all() and not hb[:, Hd + 1:].any()
    # ---- backward sweep -----------------------------------------------------------------------------------------------
    g = rng.normal(size=(B, Hd)).astype(np.float32)
    ref.backward(torch.from_numpy(g.astype(np.float64)))
    hg = be.dev(g)
    dgi0 = np.full((B * N, Kp), 0x3F80, dtype=np.uint16)              # rows t < T must be fully overwritten up to column 3*Hg;
    dgi0[:, 3 * Hg:] = 0                                              # the K padding beyond is zero-filled once by the host
    dgi = be.dev(dgi0)
    dgh = [be.empty((B, Kp), np.uint16) for _ in range(T)]
    carry = [be.poison((B, Hp), np.float32), be.poison((B, Hp), np.float32)]
    dght = [be.dev(np.zeros((B16, Kp), dtype=np.uint16)) for _ in range(2)]
    for i, t in enumerate(range(T - 1, -2, -1)):
        first = 1 if i == 0 else 0
        ck(be, be.lib.nr_gru_bwd_step(be.ptr(hg) if first else None, None if first else be.ptr(dght[(i + 1) % 2]),
                                      None if first else be.ptr(carry[(i + 1) % 2]), be.ptr(WhhT_p), be.ptr(gates[t]) if t >= 0 else None,
                                      be.ptr(H_all[t]) if t >= 0 else None, be.ptr(hlen), be.ptr(dgi) if t >= 0 else None,
                                      be.ptr(dgh[t]) if t >= 0 else None, be.ptr(dght[i % 2]) if t >= 0 else None, be.ptr(carry[i % 2]), B, N, Hd, t,
                                      first, be.stream))
        if t >= 0 and t in (T - 1, 0):
            be.sync()
            assert np.array_equal(untile(be.np(dght[i % 2]), B16, Kp)[:B, :3 * Hg], be.np(dgh[t])[:, :3 * Hg])
    be.sync()
    n_calls = T + 1
    dh0 = be.np(carry[(n_calls - 1) % 2])[:, :Hd]
    # nr_gru_bwd_seq: same launches in one call
    dgi2 = be.dev(dgi0); dgh2 = be.dev(np.zeros((T, B, Kp), dtype=np.uint16))
    dght2 = be.dev(np.zeros((2, B16, Kp), dtype=np.uint16)); carry2 = be.poison((2, B, Hp), np.float32)
    ck(be, be.lib.nr_gru_bwd_seq(be.ptr(hg), be.ptr(WhhT_p), be.ptr(gs_d), be.ptr(Hs_d), be.ptr(hlen), be.ptr(dgi2), be.ptr(dgh2), be.ptr(dght2),
                                 be.ptr(carry2), B, N, Hd, T, be.stream))
    be.sync()
    exact = not persistent_bwd_active(be, B, Hd, T)
    same_sweep(be.np(carry2)[T % 2][:, :Hd], be.np(carry[T % 2])[:, :Hd], exact, 'dh_0 (nr_gru_bwd_seq)')
    same_sweep(be.np(dgi2), be.np(dgi), exact, 'dgi (nr_gru_bwd_seq)')
    for t in range(T):
        same_sweep(be.np(dgh2)[t][:, :3 * Hg], be.np(dgh[t])[:, :3 * Hg], exact, f'dgh[{t}] (nr_gru_bwd_seq)')
    dgi3 = be.dev(dgi0); dgh3 = be.dev(np.zeros((T, B, Kp), dtype=np.uint16))
    dght3 = be.dev(np.zeros((nbuf, B16, Kp), dtype=np.uint16)); carry3 = be.poison((2, B, Hp), np.float32)
    ck(be, be.lib.nr_gru_bwd_seq_n(be.ptr(hg), be.ptr(WhhT_p), be.ptr(gs_d), be.ptr(Hs_d), be.ptr(hlen), be.ptr(dgi3), be.ptr(dgh3), be.ptr(dght3),
                                   nbuf, be.ptr(carry3), B, N, Hd, T, be.stream))
    be.sync()
    same_sweep(be.np(carry3)[T % 2][:, :Hd], be.np(carry[T % 2])[:, :Hd], exact, 'dh_0 (nr_gru_bwd_seq_n)')
    same_sweep(be.np(dgi3), be.np(dgi), exact, 'dgi (nr_gru_bwd_seq_n)')
    for t in range(T):
        same_sweep(be.np(dgh3)[t][:, :3 * Hg], be.np(dgh[t])[:, :3 * Hg], exact, f'dgh[{t}] (nr_gru_bwd_seq_n)')
    if not exact:
        # the oracle comparisons below use the step launches' buffers: repeat the two that matter most on the persistent sweep's own outputs
        dgi_p = bf16_to_f32(be.np(dgi2)).reshape(B, N, Kp).astype(np.float64)
        dGi_p = gate_unpad(dgi_p[:, :, :3 * Hg], Hd, Hg)
        dGi_p[:, T:] = 0
        relp = lambda a, b: np.abs(np.asarray(a, dtype=np.float64) - b).max() / (np.abs(b).max() + 1e-30)
        assert relp(be.np(carry2)[T % 2][:, :Hd], h0t.grad.numpy()) < 3e-2
        assert relp((dGi_p.reshape(B * N, 3 * Hd) @ W_ih.astype(np.float64)).reshape(B, N, I), xt.grad.numpy()) < 3e-2
        dGh_p = np.stack([gate_unpad(bf16_to_f32(be.np(dgh2)[t]).astype(np.float64)[:, :3 * Hg], Hd, Hg) for t in range(T)])
        Hprev_p = np.stack([bf16_to_f32(be.np(H_all[t]))[:, :Hd].astype(np.float64) for t in range(T)])
        assert relp(np.einsum('tbk,tbj->kj', dGh_p, Hprev_p), enc.gru.weight_hh_l0.grad.numpy()) < 3e-2
    rel = lambda a, b: np.abs(np.asarray(a, dtype=np.float64) - b).max() / (np.abs(b).max() + 1e-30)
    assert rel(dh0, h0t.grad.numpy()) < 3e-2, rel(dh0, h0t.grad.numpy())
    dgi_np = bf16_to_f32(be.np(dgi)).reshape(B, N, Kp).astype(np.float64)
    assert not dgi_np[:, :T, 3 * Hg:].any()
    for b in range(B):
        assert not dgi_np[b, lens[b]:T].any(), 'finished samples must contribute zero gate gradients'
        assert (be.np(dgi).reshape(B, N, Kp)[b, T:, :3 * Hg] == 0x3F80).all()          # rows t >= T are left to the host
    dGi = gate_unpad(dgi_np[:, :, :3 * Hg], Hd, Hg)
    dGi[:, T:] = 0
    dx = dGi.reshape(B * N, 3 * Hd) @ W_ih.astype(np.float64)
    assert rel(dx.reshape(B, N, I), xt.grad.numpy()) < 3e-2
    assert rel(dGi.reshape(B * N, -1).T @ x.reshape(B * N, I).astype(np.float64), enc.gru.weight_ih_l0.grad.numpy()) < 3e-2
    assert rel(dGi.sum((0, 1)), enc.gru.bias_ih_l0.grad.numpy()) < 3e-2
    dGh = np.stack([gate_unpad(bf16_to_f32(be.np(d)).astype(np.float64)[:, :3 * Hg], Hd, Hg) for d in dgh])       # [T,B,3Hd]
    Hprev = np.stack([bf16_to_f32(be.np(H_all[t]))[:, :Hd].astype(np.float64) for t in range(T)])
    assert rel(np.einsum('tbk,tbj->kj', dGh, Hprev), enc.gru.weight_hh_l0.grad.numpy()) < 3e-2
    assert rel(dGh.sum((0, 1)), enc.gru.bias_hh_l0.grad.numpy()) < 3e-2
    return err


def check_gru_rows(be, B=9, N=5, R=7, Hd=900, I=900, seed=3):
    """nr_gru_fwd_seq_rows (histories index a table of per-item input projections, longest-first step schedule) == nr_gru_fwd_seq_n on the
    gathered projections, bit for bit; nr_gru_gate_rows on gh = bf16(h) @ bf16(W_hh)^T (numpy) follows the same recurrence to fp32 rounding."""
    rng = np.random.default_rng(seed)
    Hg, Hp, Kp = dims(be, Hd)
    k = 1.0 / np.sqrt(Hd)
    W_hh = rng.uniform(-k, k, size=(3 * Hd, Hd)).astype(np.float32)
    b_ih = rng.uniform(-k, k, size=3 * Hd).astype(np.float32)
    b_hh = rng.uniform(-k, k, size=3 * Hd).astype(np.float32)
    gi_tab = gate_pad(rng.normal(0, 0.5, size=(R + 1, 3 * Hd)).astype(np.float32), Hd, Hg)
    gi_tab[R] = 0.0                                                    # the zero row padded slots point at
    h0 = rng.normal(0, 0.5, size=(B, Hd)).astype(np.float32)
    lens = np.sort(rng.integers(1, N + 1, size=B))[::-1].astype(np.int32).copy()       # longest first
    lens[0] = N
    T = int(lens[0])
    rows = rng.integers(0, R, size=(B, N)).astype(np.int32)
    for b in range(B):
        rows[b, lens[b]:] = R
    active = np.ascontiguousarray((lens[None, :] > np.arange(T)[:, None]).sum(axis=1).astype(np.int32))
    Whh_p, Whh_rm = be.poison((3 * Hg, Hp), np.uint16), be.poison((3 * Hg, Hp), np.uint16)
    ck(be, be.lib.nr_pack_gru(be.ptr(be.dev(W_hh)), Hd, Hd, Hp, be.ptr(Whh_p), None, 1, be.stream))
    ck(be, be.lib.nr_pack_gru(be.ptr(be.dev(W_hh)), Hd, Hd, Hp, be.ptr(Whh_rm), None, 0, be.stream))
    hb_ih, hb_hh, hlen = be.dev(b_ih), be.dev(b_hh), be.dev(lens)
    h0p = np.zeros((B, Hp), dtype=np.float32); h0p[:, :Hd] = h0
    B16 = (B + 15) // 16 * 16

    def start():
        hf = be.dev(h0p.copy())
        hb = be.empty((B, Hp), np.uint16)
        ck(be, be.lib.nr_rows_to_bf16(be.ptr(hf), Hp, Hd, be.ptr(hb), Hp, B, be.stream))
        ht = be.dev(np.zeros((2, B16, Hp), dtype=np.uint16))
        ck(be, be.lib.nr_tile_rows_bf16(be.ptr(hb), B, Hp, be.ptr(ht), be.stream))
        return hf, hb, ht
    # (a) reference: the gathered projections through the ordinary sweep
    gi_g = be.dev(np.ascontiguousarray(gi_tab[rows].reshape(B * N, 3 * Hg)))
    hf2 = be.dev(np.stack([h0p, np.zeros_like(h0p)]))
    _, _, ht_a = start()
    ck(be, be.lib.nr_gru_fwd_seq_n(be.ptr(gi_g), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen), be.ptr(ht_a), 2, None, be.ptr(hf2), None,
                                   B, N, Hd, T, be.stream))
    be.sync()
    ref = be.np(hf2).reshape(2, B, Hp)[T % 2][:, :Hd]
    # (b) row-indexed sweep with the shrinking batch
    hf_b, _, ht_b = start()
    ck(be, be.lib.nr_gru_fwd_seq_rows(be.ptr(be.dev(gi_tab)), be.ptr(be.dev(rows)), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen),
                                      be.ptr(ht_b), be.ptr(hf_b), active.ctypes.data, B, N, Hd, T, be.stream))
    be.sync()
    assert np.array_equal(be.np(hf_b)[:, :Hd], ref), 'row-indexed sweep differs from the sweep on gathered projections'
    bad = active.copy(); bad[-1] = B + 1
    assert be.lib.nr_gru_fwd_seq_rows(be.ptr(be.dev(gi_tab)), be.ptr(be.dev(rows)), be.ptr(Whh_p), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen),
                                      be.ptr(ht_b), be.ptr(hf_b), bad.ctypes.data, B, N, Hd, T, be.stream) != 0
    # (c) gate kernel on a host-side recurrent product
    hf_c, hb_c, _ = start()
    Wq = bf16_to_f32(be.np(Whh_rm)).astype(np.float64)                 # [3*Hg][Hp]
    h_gi, h_rows = be.dev(gi_tab), be.dev(rows)
    for t in range(T):
        Bt = int(active[t])
        be.sync()
        gh = (bf16_to_f32(be.np(hb_c))[:Bt].astype(np.float64) @ Wq.T).astype(np.float32)
        ck(be, be.lib.nr_gru_gate_rows(be.ptr(h_gi), be.ptr(h_rows), be.ptr(be.dev(np.ascontiguousarray(gh))), be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen),
                                       be.ptr(hf_c), be.ptr(hb_c), Bt, N, Hd, t, be.stream))
    be.sync()
    got = be.np(hf_c)[:, :Hd]
    assert np.abs(got - ref).max() <= 1e-3, f'gate-kernel recurrence differs by {np.abs(got - ref).max()}'
    hb_fin = be.np(hb_c)
    assert (hb_fin[:, Hd] == 0x3F80).all() and np.array_equal(hb_fin[:, :Hd], f32_to_bf16(got))
    assert be.lib.nr_gru_gate_rows(be.ptr(h_gi), be.ptr(h_rows), None, be.ptr(hb_ih), be.ptr(hb_hh), be.ptr(hlen), be.ptr(hf_c), be.ptr(hb_c), B, N, Hd, 0,
                                   be.stream) != 0
