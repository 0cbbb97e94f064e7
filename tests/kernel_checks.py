"""Backend-agnostic parity checks of the HIP kernels against the oracle.
Called with the CPU wave-emulation backend (not gpu) and with the real library on cuda:0 (gpu)."""
import numpy as np

from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_NP, NR_QP, NR_HEADS
from oracle import nrms_numpy as onp
from tests.backends import bf16_to_f32, f32_to_bf16, bf16_round

H = NR_HEADS


def ck(be, rc):
    _capi.check(be.lib, rc)


def make_params(seed=0, V=300, qdim=200, emb_std=0.5):
    rng = np.random.default_rng(seed)
    return onp.random_nrms_params(rng, V, NR_D, qdim, np.float32, emb_std=emb_std)


# ---------------------------------------------------------------------------------------------------
def check_probe_mfma(be):
    rng = np.random.default_rng(1)
    A = bf16_round(rng.normal(size=(16, 32)).astype(np.float32))
    B = bf16_round(rng.normal(size=(32, 16)).astype(np.float32))
    B[3, 5] = 7.0     # asymmetric marker
    d = be.poison((16, 16), np.float32)
    ck(be, be.lib.nr_probe_mfma(be.ptr(be.dev(f32_to_bf16(A))), be.ptr(be.dev(f32_to_bf16(B))), be.ptr(d), be.stream))
    be.sync()
    np.testing.assert_allclose(be.np(d), A.astype(np.float64) @ B.astype(np.float64), rtol=1e-5, atol=1e-5)


def check_probe_tr16(be):
    """ds_read_b64_tr_b16 (lds_tr16_b64): in each group of 16 lanes, lane i receives element j = P_{4 j + i // 4}[i % 4] of the group's 16
    four-element pieces -- with the canonical row-major 4 x 16 block addresses, and with arbitrary 8-byte-aligned per-lane addresses."""
    rng = np.random.default_rng(31)
    canon = np.array([(64 * (l >> 4) + 4 * (l & 15)) * 2 + 512 for l in range(64)], dtype=np.int32)
    strided = np.array([((l >> 4) * 8 + (l & 15) // 4) * 272 + ((l & 15) % 4) * 8 + 64 for l in range(64)], dtype=np.int32)   # rows of a [tok][136] tile
    rand = (rng.integers(0, 1000, size=64) * 8).astype(np.int32)
    for offs in (canon, strided, rand):
        out = be.poison((64, 4), np.uint16)
        ck(be, be.lib.nr_probe_tr16(be.ptr(be.dev(offs)), be.ptr(out), be.stream))
        be.sync()
        got = be.np(out)
        ref = np.zeros((64, 4), dtype=np.uint16)
        for l in range(64):
            grp, i = l & ~15, l & 15
            for j in range(4):
                ref[l, j] = offs[grp + 4 * j + i // 4] // 2 + i % 4
        assert np.array_equal(got, ref), f'ds_read_b64_tr_b16 semantics differ from the model:\n{got[:16]}\nexpected\n{ref[:16]}'


def check_gather(be, n_tokens=1000, V=777):
    rng = np.random.default_rng(2)
    table = rng.normal(size=(V, NR_D)).astype(np.float32)
    ids = rng.integers(0, V, size=n_tokens).astype(np.int64)
    ids[:3] = [0, V - 1, 0]
    out = be.poison((n_tokens, NR_D), np.float32)
    ck(be, be.lib.nr_gather_rows_f32(be.ptr(be.dev(ids)), be.ptr(be.dev(table)), be.ptr(out), n_tokens, NR_D, V, be.stream))
    be.sync()
    assert np.array_equal(be.np(out), table[ids])          # bit exact


def pack_qkv(be, params, prefix):
    m = prefix + 'multihead_self_attention.'
    Wp = be.poison((3 * NR_NP, NR_KP), np.uint16)
    bp = be.poison((3 * NR_NP,), np.float32)
    hs = [be.dev(params[m + n]) for n in ('W_Q.weight', 'W_Q.bias', 'W_K.weight', 'W_K.bias', 'W_V.weight', 'W_V.bias')]
    ck(be, be.lib.nr_pack_qkv(*[be.ptr(h) for h in hs], be.ptr(Wp), be.ptr(bp), be.stream))
    return Wp, bp


def pack_additive(be, params, prefix):
    a = prefix + 'additive_attention.'
    W, b, q = params[a + 'linear.weight'], params[a + 'linear.bias'], params[a + 'attention_query_vector']
    Wap = be.poison((NR_QP, NR_KP), np.uint16)
    bap = be.poison((NR_QP,), np.float32)
    qvp = be.poison((NR_QP,), np.float32)
    ck(be, be.lib.nr_pack_additive(be.ptr(be.dev(W)), be.ptr(be.dev(b)), be.ptr(be.dev(q)), W.shape[0],
                                   be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.stream))
    return Wap, bap, qvp


def untile(a, R, K):
    """Tile order (include/nr_engine.h) -> row-major [R][K]: blocks (row tile, k-step) of 64 lane fragments (g, li) x 8."""
    return np.asarray(a).reshape(R // 16, K // 32, 4, 16, 8).transpose(0, 3, 1, 2, 4).reshape(R, K)


def check_pack(be):
    params = make_params(3)
    Wp, bp = pack_qkv(be, params, 'news_encoder.')
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    be.sync()
    m = 'news_encoder.multihead_self_attention.'
    Wp, bp = untile(be.np(Wp), 3 * NR_NP, NR_KP), be.np(bp)
    for i, n in enumerate(('W_Q', 'W_K', 'W_V')):
        blk = Wp[i * NR_NP:(i + 1) * NR_NP]
        assert np.array_equal(blk[:NR_D, :NR_D], f32_to_bf16(params[m + n + '.weight']))
        assert not blk[NR_D:].any() and not blk[:, NR_D:].any()
        assert np.array_equal(bp[i * NR_NP:i * NR_NP + NR_D], params[m + n + '.bias'])
        assert not bp[i * NR_NP + NR_D:(i + 1) * NR_NP].any()
    a = 'news_encoder.additive_attention.'
    Wap = untile(be.np(Wap), NR_QP, NR_KP)
    assert np.array_equal(Wap[:200, :NR_D], f32_to_bf16(params[a + 'linear.weight']))
    assert not Wap[200:].any() and not Wap[:, NR_D:].any()
    assert np.array_equal(be.np(bap)[:200], params[a + 'linear.bias']) and not be.np(bap)[200:].any()
    assert np.array_equal(be.np(qvp)[:200], params[a + 'attention_query_vector']) and not be.np(qvp)[200:].any()


# ---------------------------------------------------------------------------------------------------
def mhsa_quantized_oracle(x, params, prefix, mask2=None, scale=1.0, key_len=None):
    """MHSA with the engine's rounding points (bf16 X, W, QKV, P; fp32 accumulate), float64 arithmetic between.  key_len [B]: the
    `length` mask of multihead_self.py:60-70 (exp(scores) of keys >= length multiplied by 0 before the row sum)."""
    m = prefix + 'multihead_self_attention.'
    xq = bf16_round(x.astype(np.float32)).astype(np.float64)
    W = {n: bf16_round(params[m + n + '.weight']).astype(np.float64) for n in ('W_Q', 'W_K', 'W_V')}
    b = {n: params[m + n + '.bias'].astype(np.float64) for n in ('W_Q', 'W_K', 'W_V')}
    B, S, D = x.shape
    dk = D // H

    def proj(n):
        y = bf16_round((xq @ W[n].T + b[n]).astype(np.float32)).astype(np.float64)
        return y.reshape(B, S, H, dk).transpose(0, 2, 1, 3)
    q, k, v = proj('W_Q'), proj('W_K'), proj('W_V')
    s = (q @ np.swapaxes(k, -1, -2)) / np.sqrt(np.float32(dk)).astype(np.float64)
    e = np.exp(s)
    if key_len is not None:
        e = e * (np.arange(S)[None, :] < np.asarray(key_len)[:, None])[:, None, None, :]
    pr = e / (e.sum(-1, keepdims=True) + 1e-8)
    pr = bf16_round(pr.astype(np.float32)).astype(np.float64)
    ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, D)
    if mask2 is not None:
        ctx = ctx * mask2 * scale
    return ctx


def run_mhsa(be, params, prefix, S, n_seq, ids=None, table=None, x=None, p_drop=0.0, seed=0, save=False, key_len=None):
    Wp, bp = pack_qkv(be, params, prefix)
    if key_len is not None:
        hl = be.dev(np.asarray(key_len, dtype=np.int32))
        ctx = be.poison((n_seq * S, NR_KP), np.uint16)
        sp4 = (S + 3) // 4 * 4
        sv = (be.empty((n_seq * S, NR_KP), np.uint16), be.empty((n_seq * S, NR_KP), np.uint16),
              be.empty((n_seq, H, 20, sp4), np.uint16)) if save else (None, None, None)
        hi, ht = (be.dev(ids.astype(np.int64)), be.dev(table)) if ids is not None else (None, None)
        hx = be.dev(x.astype(np.float32)) if ids is None else None
        ck(be, be.lib.nr_mhsa_fwd_len(be.ptr(hi), be.ptr(ht), table.shape[0] if ids is not None else 0, be.ptr(hx), be.ptr(Wp), be.ptr(bp),
                                      be.ptr(ctx), be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), None, be.ptr(hl), n_seq, S, p_drop, seed, be.stream))
        be.sync()
        return (be.np(ctx), sv) if save else (be.np(ctx), ctx)
    ctx = be.poison((n_seq * S, NR_KP), np.uint16)
    sp4 = (S + 3) // 4 * 4
    sv = (be.empty((n_seq * S, NR_KP), np.uint16), be.empty((n_seq * S, NR_KP), np.uint16),
          be.empty((n_seq, H, 20, sp4), np.uint16)) if save else (None, None, None)
    if ids is not None:
        hi, ht = be.dev(ids.astype(np.int64)), be.dev(table)
        ck(be, be.lib.nr_mhsa_fwd(be.ptr(hi), be.ptr(ht), table.shape[0], None, be.ptr(Wp), be.ptr(bp), be.ptr(ctx),
                                  be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), n_seq, S, p_drop, seed, be.stream))
    else:
        hx = be.dev(x.astype(np.float32))
        ck(be, be.lib.nr_mhsa_fwd(None, None, 0, be.ptr(hx), be.ptr(Wp), be.ptr(bp), be.ptr(ctx),
                                  be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), n_seq, S, p_drop, seed, be.stream))
    be.sync()
    if save:
        return be.np(ctx), sv
    return be.np(ctx), ctx


def export_mask(be, n_elem, p, seed, site):
    m = be.poison((n_elem,), np.float32)
    ck(be, be.lib.nr_dropout_mask(be.ptr(m), n_elem, p, seed, site, be.stream))
    be.sync()
    return be.np(m)


def assert_ctx_close(ctx_u16, ref, what):
    got = bf16_to_f32(ctx_u16[:, :NR_D]).astype(np.float64)
    assert (ctx_u16[:, NR_D] == 0x3F80).all() and not ctx_u16[:, NR_D + 1:].any(), f'{what}: ctx K-padding wrong'
    ref = ref.reshape(got.shape)
    err = np.abs(got - ref)
    scale = np.abs(ref).max()
    # 2 bf16 ulps relative + a small absolute floor (bf16 rounding of P/QKV can flip the last bit)
    bad = err > (2.0 ** -7) * np.abs(ref) + 4e-3 * scale
    assert not bad.any(), f'{what}: {bad.sum()} / {bad.size} elements off, max err {err.max():.4g} (scale {scale:.3g})'
    return err.max() / scale


def check_mhsa_gather(be, n_seq=6, V=300, p_drop=0.0, seed=1234):
    S = 20
    params = make_params(4, V)
    rng = np.random.default_rng(5)
    ids = rng.integers(1, V, size=(n_seq, S))
    ids[:, 13:] = 0                       # right padding, row 0 is an ordinary row
    ids[1] = 0
    table = params['news_encoder.word_embedding.weight']
    ctx_np, _ = run_mhsa(be, params, 'news_encoder.', S, n_seq, ids=ids, table=table, p_drop=p_drop, seed=seed)
    x = table[ids].astype(np.float64)
    mask2, scale = None, 1.0
    if p_drop > 0:
        scale = np.float32(1.0 / (1.0 - p_drop))
        m1 = export_mask(be, n_seq * S * NR_D, p_drop, seed, 1).reshape(n_seq, S, NR_D)
        mask2 = export_mask(be, n_seq * S * NR_D, p_drop, seed, 2).reshape(n_seq, S, NR_D)
        keep = m1.mean()
        assert abs(keep - (1 - p_drop)) < 0.02, keep
        assert abs(mask2.mean() - (1 - p_drop)) < 0.02
        assert not np.array_equal(m1, mask2)
        x = x * m1 * scale
    ref = mhsa_quantized_oracle(x, params, 'news_encoder.', mask2, scale)
    rel = assert_ctx_close(ctx_np, ref, f'mhsa gather S=20 p={p_drop}')
    # and against the un-quantized fp64 oracle at bf16-level tolerance
    m = 'news_encoder.multihead_self_attention.'
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    full, _ = onp.mhsa(x, p64[m + 'W_Q.weight'], p64[m + 'W_Q.bias'], p64[m + 'W_K.weight'], p64[m + 'W_K.bias'],
                       p64[m + 'W_V.weight'], p64[m + 'W_V.bias'], H)
    if mask2 is not None:
        full = full * mask2 * scale
    got = bf16_to_f32(ctx_np[:, :NR_D]).reshape(full.shape)
    assert np.abs(got - full).max() < 0.03 * np.abs(full).max()
    return rel


def check_mhsa_key_len(be, S=20, n_seq=7):
    """nr_mhsa_fwd_len: per-sequence key lengths (the `length` argument of MultiHeadSelfAttention.forward), all queries computed."""
    params = make_params(6)
    rng = np.random.default_rng(17)
    x = rng.normal(0, 0.7, size=(n_seq, S, NR_D)).astype(np.float32)
    key_len = rng.integers(1, S + 1, size=n_seq)
    key_len[0], key_len[-1] = S, 1
    ctx_np, _ = run_mhsa(be, params, 'user_encoder.', S, n_seq, x=x, key_len=key_len)
    ref = mhsa_quantized_oracle(x, params, 'user_encoder.', key_len=key_len)
    assert_ctx_close(ctx_np, ref, f'mhsa key_len S={S}')
    # a sequence truncated to its key length gives the same rows for the queries below it
    L = int(key_len[1])
    ref_t = mhsa_quantized_oracle(x[1:2, :L], params, 'user_encoder.')
    np.testing.assert_allclose(ref[1, :L], ref_t[0], rtol=1e-12, atol=1e-12)


def check_mhsa_dense(be, n_seq=3):
    S = 50
    params = make_params(6)
    rng = np.random.default_rng(7)
    x = rng.normal(0, 0.7, size=(n_seq, S, NR_D)).astype(np.float32)
    x[0, :20] = 0.0                       # left-padded history slots (zero vectors at eval time)
    ctx_np, _ = run_mhsa(be, params, 'user_encoder.', S, n_seq, x=x)
    ref = mhsa_quantized_oracle(x, params, 'user_encoder.')
    return assert_ctx_close(ctx_np, ref, 'mhsa dense S=50')


def check_additive_valid(be, S=20, n_seq=6, valid=13):
    """nr_additive_fwd_v: pooling over the first `valid` tokens == the reference pooling of the truncated sequences; weights beyond are 0."""
    params = make_params(8)
    rng = np.random.default_rng(19)
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    out = be.poison((n_seq, NR_D), np.float32)
    aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd_v(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), NR_D, None, 0, be.ptr(aw),
                                    n_seq, S, valid, be.stream))
    be.sync()
    a = 'news_encoder.additive_attention.'
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)[:, :valid]
    ref, w, _ = onp.additive(x, bf16_round(params[a + 'linear.weight']).astype(np.float64), params[a + 'linear.bias'].astype(np.float64),
                             params[a + 'attention_query_vector'].astype(np.float64))
    got_w = be.np(aw)
    np.testing.assert_allclose(got_w[:, :valid], w, rtol=2e-4, atol=2e-6)
    assert not got_w[:, valid:].any()
    np.testing.assert_allclose(be.np(out), ref, rtol=2e-4, atol=2e-5)
    assert be.lib.nr_additive_fwd_v(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), NR_D, None, 0, be.ptr(aw),
                                    n_seq, S, S + 1, be.stream) != 0


def check_additive(be, S=20, n_seq=6):
    params = make_params(8)
    rng = np.random.default_rng(9)
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    out = be.poison((n_seq, NR_D), np.float32)
    aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), be.ptr(aw),
                                  n_seq, S, be.stream))
    be.sync()
    a = 'news_encoder.additive_attention.'
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)
    ref, w, _ = onp.additive(x, bf16_round(params[a + 'linear.weight']).astype(np.float64),
                             params[a + 'linear.bias'].astype(np.float64),
                             params[a + 'attention_query_vector'].astype(np.float64))
    np.testing.assert_allclose(be.np(aw), w, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(be.np(out), ref, rtol=2e-4, atol=2e-5)


def check_score_dot(be, B=37, C=3):
    rng = np.random.default_rng(10)
    cand = rng.normal(size=(B, C, NR_D)).astype(np.float32)
    user = rng.normal(size=(B, NR_D)).astype(np.float32)
    out = be.poison((B, C), np.float32)
    ck(be, be.lib.nr_score_dot(be.ptr(be.dev(cand)), be.ptr(be.dev(user)), be.ptr(out), B, C, NR_D, be.stream))
    be.sync()
    np.testing.assert_allclose(be.np(out), onp.dot_score(cand.astype(np.float64), user.astype(np.float64)), rtol=1e-5, atol=1e-4)


def check_score_csr(be, n_news=50, n_users=7, n_impr=11):
    rng = np.random.default_rng(11)
    news = rng.normal(size=(n_news, NR_D)).astype(np.float32)
    users = rng.normal(size=(n_users, NR_D)).astype(np.float32)
    lens = rng.integers(0, 9, size=n_impr)
    lens[2] = 0                                       # empty impression
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(ptr[-1])
    cidx = rng.integers(0, n_news, size=nnz).astype(np.int32)
    cidx[::5] = -1                                    # PADDED_NEWS
    uidx = rng.integers(0, n_users, size=n_impr).astype(np.int32)
    out = be.poison((nnz,), np.float32)
    ck(be, be.lib.nr_score_csr(be.ptr(be.dev(news)), be.ptr(be.dev(users)), be.ptr(be.dev(cidx)), be.ptr(be.dev(ptr)),
                               be.ptr(be.dev(uidx)), be.ptr(out), n_impr, nnz, NR_D, be.stream))
    be.sync()
    ref = np.zeros(nnz)
    for i in range(n_impr):
        for j in range(ptr[i], ptr[i + 1]):
            ref[j] = 0.0 if cidx[j] < 0 else news[cidx[j]].astype(np.float64) @ users[uidx[i]].astype(np.float64)
    np.testing.assert_allclose(be.np(out), ref, rtol=1e-5, atol=1e-4)


def check_bad_args(be):
    assert be.lib.nr_mhsa_fwd(None, None, 0, None, None, None, None, None, None, None, 1, 20, 0.0, 0, be.stream) != 0
    assert b'nr_mhsa_fwd' in be.lib.nr_last_error()
    Wp = be.empty((3 * NR_NP, NR_KP), np.uint16); bp = be.empty((3 * NR_NP,), np.float32)
    ctx = be.empty((4 * 33, NR_KP), np.uint16); x = be.empty((4, 33, NR_D), np.float32)
    rc = be.lib.nr_mhsa_fwd(None, None, 0, be.ptr(x), be.ptr(Wp), be.ptr(bp), be.ptr(ctx), None, None, None, 4, 33, 0.0, 0, be.stream)
    assert rc == -1 and be.lib.nr_supported_seq_len(33) == 0 and be.lib.nr_supported_seq_len(20) == 1
    # the entry points added in round 2 follow the same convention: non-zero return code, text in nr_last_error(), nothing launched
    a = be.empty((64,), np.float32)
    pa = be.ptr(a)
    assert be.lib.nr_additive_bwd_act(pa, pa, pa, pa, pa, pa, pa, pa, pa, pa, None, 0.2, 4, 20, be.stream) != 0 and b'nr_additive_bwd_act' in be.lib.nr_last_error()
    assert be.lib.nr_additive_bwd_act(pa, pa, pa, pa, pa, pa, pa, pa, pa, pa, pa, 1.0, 4, 20, be.stream) != 0      # p_drop out of range
    assert be.lib.nr_gru_fwd_seq_n(pa, pa, pa, pa, pa, pa, 1, None, pa, None, 4, 3, 48, 2, be.stream) != 0 and b'nr_gru_fwd_seq' in be.lib.nr_last_error()
    assert be.lib.nr_gru_bwd_seq_n(pa, pa, pa, pa, pa, pa, pa, pa, 1, pa, 4, 3, 48, 2, be.stream) != 0
    assert be.lib.nr_gru_seq_buffers(4, 48, 3) == 2 and be.lib.nr_gru_seq_buffers(512, 900, 0) == 2
    assert be.lib.nr_sort_ids(None, 0, 0, None, None, None, 0, be.stream) != 0 if hasattr(be.lib, 'nr_sort_ids') else True


# ---------------------------------------------------------------------------------------------------
# backward kernels
# ---------------------------------------------------------------------------------------------------
def _note_kernel_err(what, rms, worst):
    """Measured errors of a GPU run go to gpurun_out/measured_kernel_err.jsonl (the source of the rms bounds below; profiles/r05_measured_kernel_err.txt)."""
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir) and os.environ.get('NR_NOTE_KERNEL_ERR', '1') == '1':
        try:
            with open(os.path.join(out_dir, 'measured_kernel_err.jsonl'), 'a') as f:
                f.write(json.dumps({"what": what, "rms_rel": rms, "max_over_scale": worst}) + "\n")
        except OSError:
            pass


def close_bf16(got, ref, what, rel=2.0 ** -6, floor=6e-3, rms_max=5e-3):
    """Element by element: |err| <= rel |ref| + floor * max|ref| (a bf16 result of bf16-operand arithmetic); AND in aggregate: the rms error is at
    most rms_max of the rms of the reference -- the element bound alone would let a kernel drop one k-step of one tile in thirty (every element
    still inside its floor); the aggregate moves by an order of magnitude when that happens.  Measured on MI355X (profiles/r05_measured_kernel_err.txt):
    1.65e-3 - 1.71e-3 for single-product kernels (the bf16 rounding of the result), 1.8e-3 - 2.4e-3 for the attention backward (P and dS are bf16
    operands of its second products: its callers pass 7.5e-3); the defaults are ~3x those."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    scale = np.abs(ref).max() + 1e-30
    bad = err > rel * np.abs(ref) + floor * scale
    assert not bad.any(), f'{what}: {bad.sum()}/{bad.size} off, max err {err.max():.4g}, scale {scale:.4g}'
    rms = float(np.sqrt((err ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-30))
    _note_kernel_err(what, rms, float(err.max() / scale))
    assert rms <= rms_max, f'{what}: rms error {rms:.3g} of the reference rms (bound {rms_max:.3g})'
    return err.max() / scale


def check_attn_bwd(be, S=20, n_seq=5, p_drop=0.0, seed=77, with_key_len=False, ldc=NR_D):
    """ldc: row stride of the dctx_gemm input (NR_KP = the training layout, which the S = 20 kernel copies through LDS)."""
    from news_recommendation_amd._capi import NR_LDG
    params = make_params(12)
    rng = np.random.default_rng(13)
    x = rng.normal(0, 0.7, size=(n_seq, S, NR_D)).astype(np.float32)
    key_len = None
    if with_key_len:
        key_len = rng.integers(1, S + 1, size=n_seq)
        key_len[0] = S
    ctx_np, sv = run_mhsa(be, params, 'user_encoder.', S, n_seq, x=x, p_drop=0.0, save=True, key_len=key_len)
    qs, ks, vts = [be.np(h) for h in sv]
    dk = 20
    q = bf16_to_f32(qs[:, :NR_D]).astype(np.float64).reshape(n_seq, S, H, dk).transpose(0, 2, 1, 3)
    k = bf16_to_f32(ks[:, :NR_D]).astype(np.float64).reshape(n_seq, S, H, dk).transpose(0, 2, 1, 3)
    v = bf16_to_f32(vts).astype(np.float64)[:, :, :, :S].transpose(0, 1, 3, 2)            # [n,H,S,dk]
    # the saved tensors are what the forward used: check them against the quantised oracle
    m = 'user_encoder.multihead_self_attention.'
    xq = bf16_round(x).astype(np.float64)
    for name, t in (('W_Q', q), ('W_K', k), ('W_V', v)):
        ref = xq @ bf16_round(params[m + name + '.weight']).astype(np.float64).T + params[m + name + '.bias']
        ref = ref.reshape(n_seq, S, H, dk).transpose(0, 2, 1, 3)
        close_bf16(t, ref, 'saved ' + name, rel=2.0 ** -7, floor=1e-3)
    dg = rng.normal(0, 0.05, size=(n_seq * S, NR_D)).astype(np.float32)
    dg_u = f32_to_bf16(dg)
    dg_in = np.full((n_seq * S, ldc), 0x7FC0, dtype=np.uint16)          # padding columns hold NaN: they must never be read as data
    dg_in[:, :NR_D] = dg_u
    aw = rng.random(size=(n_seq, S)).astype(np.float32); aw /= aw.sum(1, keepdims=True)
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    dqkv = be.empty((n_seq * S, NR_LDG), np.uint16)
    if key_len is None:
        ck(be, be.lib.nr_attn_bwd(be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), be.ptr(be.dev(dg_in)), ldc, be.ptr(be.dev(aw)),
                                  be.ptr(be.dev(go)), be.ptr(dqkv), n_seq, S, p_drop, seed, be.stream))
    else:
        ck(be, be.lib.nr_attn_bwd_len(be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), be.ptr(be.dev(dg_in)), ldc, be.ptr(be.dev(aw)),
                                      be.ptr(be.dev(go)), be.ptr(dqkv), be.ptr(be.dev(np.asarray(key_len, dtype=np.int32))), n_seq, S, p_drop, seed,
                                      be.stream))
    be.sync()
    out = be.np(dqkv)
    dC = bf16_to_f32(dg_u).astype(np.float64).reshape(n_seq, S, NR_D) + aw[:, :, None].astype(np.float64) * go[:, None, :]
    if p_drop > 0:
        mask2 = export_mask(be, n_seq * S * NR_D, p_drop, seed, 2).reshape(n_seq, S, NR_D)
        dC = dC * mask2 * np.float32(1.0 / (1.0 - p_drop))
    dC = bf16_round(dC.astype(np.float32)).astype(np.float64)
    g = dC.reshape(n_seq, S, H, dk).transpose(0, 2, 1, 3)
    e = np.exp(q @ np.swapaxes(k, -1, -2) / np.sqrt(np.float32(dk)).astype(np.float64))
    if key_len is not None:
        e = e * (np.arange(S)[None, :] < np.asarray(key_len)[:, None])[:, None, None, :]
    attn = e / (e.sum(-1, keepdims=True) + 1e-8)
    dattn = g @ np.swapaxes(v, -1, -2)
    dv = np.swapaxes(attn, -1, -2) @ g
    dS = attn * (dattn - (attn * dattn).sum(-1, keepdims=True)) / np.sqrt(dk)
    dq = dS @ k
    dkk = np.swapaxes(dS, -1, -2) @ q
    mg = lambda t: t.transpose(0, 2, 1, 3).reshape(n_seq * S, NR_D)
    rels = []
    for i, (name, ref) in enumerate((('dQ', dq), ('dK', dkk), ('dV', dv))):
        got = bf16_to_f32(out[:, i * NR_KP:i * NR_KP + NR_D])
        rels.append(close_bf16(got, mg(ref), f'attn_bwd {name} S={S}', rms_max=7.5e-3))
        assert not out[:, i * NR_KP + NR_D:(i + 1) * NR_KP].any()
    return rels


def check_wgrad_unpack(be, nc_w=3, nc_a=2, nwg=11, qdim=200):
    """nr_wgrad_unpack: chunk partials in the packed operand geometry -> accumulated (+=) into the nine parameter-shaped gradients."""
    rng = np.random.default_rng(41)
    dW = rng.normal(0, 1.0, size=(nc_w, 3 * NR_KP, NR_KP)).astype(np.float32)
    dWa = rng.normal(0, 1.0, size=(nc_a, NR_QP, NR_KP)).astype(np.float32)
    dq = rng.normal(0, 1.0, size=(nwg, NR_QP)).astype(np.float32)
    shapes = [(NR_D, NR_D), (NR_D,)] * 3 + [(qdim, NR_D), (qdim,), (qdim,)]
    init = [rng.normal(0, 1.0, size=sh).astype(np.float32) for sh in shapes]        # pre-existing gradient content: the kernel accumulates
    dst = [be.dev(a) for a in init]
    ck(be, be.lib.nr_wgrad_unpack(be.ptr(be.dev(dW)), nc_w, be.ptr(be.dev(dWa)), nc_a, be.ptr(be.dev(dq)), nwg, qdim,
                                  *[be.ptr(d) for d in dst], be.stream))
    be.sync()
    sW, sWa, sq = dW.astype(np.float64).sum(0), dWa.astype(np.float64).sum(0), dq.astype(np.float64).sum(0)
    ref = []
    for i in range(3):
        ref += [sW[i * NR_KP:i * NR_KP + NR_D, :NR_D], sW[i * NR_KP:i * NR_KP + NR_D, NR_D]]
    ref += [sWa[:qdim, :NR_D], sWa[:qdim, NR_D], sq[:qdim]]
    for k, (d, a, r) in enumerate(zip(dst, init, ref)):
        np.testing.assert_allclose(be.np(d), a.astype(np.float64) + r, rtol=0, atol=2e-5 * max(nc_w, nc_a, nwg) ** 0.5, err_msg=f"wgrad_unpack destination {k}")
    # destinations that are only 4-byte aligned (parameters packed back to back in a flat buffer): the element-wise form
    flat = be.dev(np.zeros(1 + sum(int(np.prod(sh)) for sh in shapes), dtype=np.float32))
    offs, o = [], 1
    for sh in shapes:
        offs.append(o); o += int(np.prod(sh))
    ck(be, be.lib.nr_wgrad_unpack(be.ptr(be.dev(dW)), nc_w, be.ptr(be.dev(dWa)), nc_a, be.ptr(be.dev(dq)), nwg, qdim,
                                  *[be.ptr(flat) + 4 * o for o in offs], be.stream))
    be.sync()
    got = be.np(flat)
    assert got[0] == 0
    for k, (o, sh, r) in enumerate(zip(offs, shapes, ref)):
        np.testing.assert_allclose(got[o:o + int(np.prod(sh))].reshape(sh), r, rtol=0, atol=2e-5 * max(nc_w, nc_a, nwg) ** 0.5, err_msg=f"wgrad_unpack (unaligned) destination {k}")
    # argument checks
    assert be.lib.nr_wgrad_unpack(None, nc_w, be.ptr(dst[0]), nc_a, be.ptr(dst[0]), nwg, qdim, *[be.ptr(d) for d in dst], be.stream) != 0
    assert be.lib.nr_wgrad_unpack(be.ptr(dst[0]), 0, be.ptr(dst[0]), nc_a, be.ptr(dst[0]), nwg, qdim, *[be.ptr(d) for d in dst], be.stream) != 0
    assert be.lib.nr_wgrad_unpack(be.ptr(dst[0]), 1, be.ptr(dst[0]), 1, be.ptr(dst[0]), nwg, NR_QP + 1, *[be.ptr(d) for d in dst], be.stream) != 0


def check_additive_bwd(be, S=20, n_seq=6, valid=None):
    """valid < S: sequences zero-padded to the instantiated length (config knobs num_words_title / num_words_abstract /
    num_clicked_news_a_user below 20 / 50): the forward pools the first `valid` positions (nr_additive_fwd_v), the backward sees zero
    attention weights beyond them and must produce exact zeros there and the truncated sequences' gradients before."""
    params = make_params(14)
    rng = np.random.default_rng(15)
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    hctx = be.dev(ctx_u)
    out = be.poison((n_seq, NR_D), np.float32)
    aw = be.poison((n_seq, S), np.float32)
    if valid is None:
        ck(be, be.lib.nr_additive_fwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), be.ptr(aw), n_seq, S, be.stream))
    else:
        ck(be, be.lib.nr_additive_fwd_v(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), NR_D, None, 0, be.ptr(aw), n_seq, S, valid,
                                        be.stream))
    V_ = S if valid is None else valid
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    nwg = be.lib.nr_additive_bwd_grid(n_seq, S)
    dpre = be.empty((n_seq * S, NR_QP), np.uint16)
    dqp = be.poison((nwg, NR_QP), np.float32)
    a_ = 'news_encoder.additive_attention.'
    WaT = be.poison((NR_KP, 224), np.uint16)
    ck(be, be.lib.nr_pack_additive_t(be.ptr(be.dev(params[a_ + 'linear.weight'])), 200, be.ptr(WaT), be.stream))
    dctx = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_additive_bwd_ex(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(be.dev(go)),
                                     be.ptr(dpre), be.ptr(dqp), be.ptr(WaT), be.ptr(dctx), n_seq, S, be.stream))
    be.sync()
    # fused input-gradient product: dctx[:, :D] = bf16(dpre) @ bf16(Wa), bit-level inputs as the kernel sees them
    # WaT = Wa^T in tile order with the pair-permuted contraction index (csrc/k_misc.h): slot 32 ks + 8 g + j <-> q = 16 (2 ks + j // 4) + 4 g + j % 4
    kap = np.arange(224)
    qmap = 16 * (2 * (kap >> 5) + ((kap & 7) >> 2)) + 4 * ((kap & 31) >> 3) + (kap & 3)
    wat = np.zeros((NR_KP, 256), dtype=np.uint16)
    wat[:, qmap] = untile(be.np(WaT), NR_KP, 224)
    assert sorted(qmap.tolist()) == list(range(224))
    assert np.array_equal(wat[:NR_D, :200], f32_to_bf16(params[a_ + 'linear.weight']).T) and not wat[NR_D:].any() and not wat[:, 200:].any()
    dref = bf16_to_f32(be.np(dpre)).astype(np.float64)[:, :200] @ bf16_round(params[a_ + 'linear.weight']).astype(np.float64)
    close_bf16(bf16_to_f32(be.np(dctx)[:, :NR_D]), dref, f'additive_bwd fused dctx S={S}', rel=2.0 ** -7, floor=1e-3)
    # and the plain entry point gives the same dpre / dq
    dpre2 = be.empty((n_seq * S, NR_QP), np.uint16); dqp2 = be.poison((nwg, NR_QP), np.float32)
    ck(be, be.lib.nr_additive_bwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(be.dev(go)),
                                  be.ptr(dpre2), be.ptr(dqp2), n_seq, S, be.stream))
    be.sync()
    assert np.array_equal(be.np(dpre), be.np(dpre2)) and np.array_equal(be.np(dqp), be.np(dqp2))
    a = 'news_encoder.additive_attention.'
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)[:, :V_]
    W = bf16_round(params[a + 'linear.weight']).astype(np.float64)
    b = params[a + 'linear.bias'].astype(np.float64); qv = params[a + 'attention_query_vector'].astype(np.float64)
    _, w, temp = onp.additive(x, W, b, qv)
    g = go.astype(np.float64)
    dw = np.einsum('bd,bsd->bs', g, x)
    ds = w * (dw - (w * dw).sum(1, keepdims=True))
    dpre_ref = ds[:, :, None] * qv[None, None, :] * (1 - temp * temp)
    dq_ref = np.einsum('bs,bsq->q', ds, temp)
    got = bf16_to_f32(be.np(dpre)).reshape(n_seq, S, NR_QP)
    assert not got[:, V_:].any() and not bf16_to_f32(be.np(dctx)).reshape(n_seq, S, NR_KP)[:, V_:, :NR_D].any(), 'padded positions must get exact zeros'
    got = got[:, :V_].reshape(-1, NR_QP)
    close_bf16(got[:, :200], dpre_ref.reshape(-1, 200), 'additive_bwd dpre', rel=2.0 ** -7, floor=2e-3)
    assert not got[:, 200:].any()
    dq = be.np(dqp).astype(np.float64).sum(0)
    np.testing.assert_allclose(dq[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())
    # cross-check the full additive backward formula chain against the oracle's additive_bwd
    dx_ref, dW_ref, db_ref, dqv_ref = onp.additive_bwd(g, x, w, temp, W, qv)
    np.testing.assert_allclose(dq_ref, dqv_ref, rtol=1e-9)
    np.testing.assert_allclose(dpre_ref.reshape(-1, 200).T @ x.reshape(-1, NR_D), dW_ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w[:, :, None] * g[:, None, :] + dpre_ref @ W, dx_ref, rtol=1e-9, atol=1e-12)


def check_gather_bf16(be, n_tokens=257, V=400, p_drop=0.2, seed=5):
    rng = np.random.default_rng(16)
    table = rng.normal(size=(V, NR_D)).astype(np.float32)
    ids = rng.integers(0, V, size=n_tokens).astype(np.int64)
    Xb = be.poison((n_tokens, NR_KP), np.uint16)
    ck(be, be.lib.nr_gather_bf16(be.ptr(be.dev(ids)), be.ptr(be.dev(table)), V, None, be.ptr(Xb), n_tokens, p_drop, seed, be.stream))
    be.sync()
    m1 = export_mask(be, n_tokens * NR_D, p_drop, seed, 1).reshape(n_tokens, NR_D)
    ref = f32_to_bf16((table[ids] * m1 * np.float32(1.0 / (1.0 - p_drop))).astype(np.float32))
    got = be.np(Xb)
    assert np.array_equal(bf16_to_f32(got[:, :NR_D]), bf16_to_f32(ref))      # value compare: -0 == +0
    assert (got[:, NR_D] == 0x3F80).all() and not got[:, NR_D + 1:].any()
    # dense mode, no dropout
    x = rng.normal(size=(n_tokens, NR_D)).astype(np.float32)
    ck(be, be.lib.nr_gather_bf16(None, None, 0, be.ptr(be.dev(x)), be.ptr(Xb), n_tokens, 0.0, 0, be.stream))
    be.sync()
    assert np.array_equal(be.np(Xb)[:, :NR_D], f32_to_bf16(x))


def check_scatter_add(be, n_tokens=999, V=50, p_drop=0.2, seed=6):
    rng = np.random.default_rng(17)
    ids = rng.integers(0, V, size=n_tokens).astype(np.int64)
    ids[::7] = 0
    dx = rng.normal(size=(n_tokens, NR_D)).astype(np.float32)
    dxu = f32_to_bf16(dx)
    grad = be.dev(np.full((V, NR_D), 0.5, dtype=np.float32))
    ck(be, be.lib.nr_embed_scatter_add(be.ptr(be.dev(ids)), be.ptr(be.dev(dxu)), NR_D, be.ptr(grad), V, n_tokens, p_drop, seed, be.stream))
    be.sync()
    m1 = export_mask(be, n_tokens * NR_D, p_drop, seed, 1).reshape(n_tokens, NR_D)
    ref = np.full((V, NR_D), 0.5, dtype=np.float64)
    contrib = bf16_to_f32(dxu).astype(np.float64) * m1 * np.float64(np.float32(1.0 / (1.0 - p_drop)))
    nz = ids != 0
    np.add.at(ref, ids[nz], contrib[nz])
    got = be.np(grad)
    assert np.all(got[0] == 0.5)                       # padding_idx row untouched
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-4)


def check_scatter_sorted(be, n_tokens=999, V=50, p_drop=0.2, seed=6):
    rng = np.random.default_rng(19)
    ids = np.minimum(rng.zipf(1.3, size=n_tokens), V - 1).astype(np.int64)      # heavy duplicates: long runs
    ids[::3] = 0
    dx = rng.normal(size=(n_tokens, NR_D)).astype(np.float32)
    dxu = f32_to_bf16(dx)
    perm = np.argsort(ids, kind='stable').astype(np.int64)
    init = rng.normal(size=(V, NR_D)).astype(np.float32)          # the kernel ACCUMULATES: the destination may be a live .grad buffer
    init[1] = 0
    grad = be.dev(init.copy())
    ck(be, be.lib.nr_embed_scatter_sorted(be.ptr(be.dev(ids[perm])), be.ptr(be.dev(perm)), be.ptr(be.dev(dxu)), NR_D, be.ptr(grad),
                                          V, n_tokens, p_drop, seed, be.stream))
    be.sync()
    ref = init.astype(np.float64)
    contrib = bf16_to_f32(dxu).astype(np.float64)
    if p_drop > 0:
        m1 = export_mask(be, n_tokens * NR_D, p_drop, seed, 1).reshape(n_tokens, NR_D)
        contrib = contrib * m1 * np.float64(np.float32(1.0 / (1.0 - p_drop)))
    nz = ids != 0
    np.add.at(ref, ids[nz], contrib[nz])
    got = be.np(grad)
    assert np.array_equal(got[0], init[0])             # padding_idx row untouched
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-4)


def check_score_bwd(be, B=33, C=3):
    rng = np.random.default_rng(18)
    cand = rng.normal(size=(B, C, NR_D)).astype(np.float32)
    user = rng.normal(size=(B, NR_D)).astype(np.float32)
    dl = rng.normal(size=(B, C)).astype(np.float32)
    dc = be.poison((B, C, NR_D), np.float32); du = be.poison((B, NR_D), np.float32)
    ck(be, be.lib.nr_score_dot_bwd(be.ptr(be.dev(dl)), be.ptr(be.dev(cand)), be.ptr(be.dev(user)), be.ptr(dc), be.ptr(du), B, C, NR_D, be.stream))
    be.sync()
    np.testing.assert_allclose(be.np(dc), dl[:, :, None] * user[:, None, :], rtol=1e-6)
    np.testing.assert_allclose(be.np(du), np.einsum('bc,bcd->bd', dl, cand), rtol=1e-5, atol=1e-5)


def check_score_ce(be, B=33, C=3, with_target=False, with_scale=False, ld_extra=0):
    """nr_score_ce_fwd / nr_score_ce_bwd against DotProductClickPredictor + CrossEntropyLoss in float64 (dot_product.py:8-19, train.py:205-206):
    logits bit-identical to nr_score_dot, the mean loss, the logit gradient, and the vector gradients written with row strides."""
    import ctypes
    rng = np.random.default_rng(41)
    cand = rng.normal(size=(B, C, NR_D)).astype(np.float32) * 0.2
    user = rng.normal(size=(B, NR_D)).astype(np.float32) * 0.2
    target = rng.integers(0, C, size=B).astype(np.int64) if with_target else np.zeros(B, dtype=np.int64)
    dc_, du_ = be.dev(cand), be.dev(user)
    t_ = be.dev(target) if with_target else None
    ref_logits_dev = be.poison((B, C), np.float32)
    ck(be, be.lib.nr_score_dot(be.ptr(dc_), be.ptr(du_), be.ptr(ref_logits_dev), B, C, NR_D, be.stream))
    logits = be.poison((B, C), np.float32); dl = be.poison((B, C), np.float32)
    rows = be.poison((B,), np.float32); loss = be.poison((1,), np.float32)
    ck(be, be.lib.nr_score_ce_fwd(be.ptr(dc_), be.ptr(du_), be.ptr(t_) if with_target else None, be.ptr(logits), be.ptr(dl), be.ptr(rows),
                                  be.ptr(loss), B, C, NR_D, be.stream))
    be.sync()
    assert np.array_equal(be.np(logits), be.np(ref_logits_dev))                     # bit exact
    lg = onp.dot_score(cand.astype(np.float64), user.astype(np.float64))
    m = lg.max(axis=1, keepdims=True)
    lse = np.log(np.exp(lg - m).sum(axis=1)) + m[:, 0]
    ref_rows = lse - lg[np.arange(B), target]
    sm = np.exp(lg - lse[:, None])
    ref_dl = sm.copy(); ref_dl[np.arange(B), target] -= 1.0; ref_dl /= B
    np.testing.assert_allclose(be.np(rows), ref_rows, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(be.np(loss)[0], ref_rows.mean(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(be.np(dl), ref_dl, rtol=2e-4, atol=2e-7)
    # backward, into strided destinations
    ldc, ldu = NR_D + ld_extra, NR_D + 2 * ld_extra
    g = np.array([0.37], dtype=np.float32) if with_scale else None
    dcand = be.poison((B * C, ldc), np.float32); duser = be.poison((B, ldu), np.float32)
    ck(be, be.lib.nr_score_ce_bwd(be.ptr(dl), be.ptr(be.dev(g)) if with_scale else None, be.ptr(dc_), be.ptr(du_), be.ptr(dcand), ldc,
                                  be.ptr(duser), ldu, B, C, NR_D, be.stream))
    be.sync()
    gs = 0.37 if with_scale else 1.0
    dlv = be.np(dl).astype(np.float64) * np.float64(np.float32(gs))
    np.testing.assert_allclose(be.np(dcand)[:, :NR_D].reshape(B, C, NR_D), dlv[:, :, None] * user[:, None, :], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(be.np(duser)[:, :NR_D], np.einsum('bc,bcd->bd', dlv, cand), rtol=1e-5, atol=1e-8)
    # bad arguments: C beyond one wave, misaligned d
    assert be.lib.nr_score_ce_fwd(be.ptr(dc_), be.ptr(du_), None, None, be.ptr(dl), be.ptr(rows), be.ptr(loss), B, 65, NR_D, be.stream) == -2
    assert be.lib.nr_score_ce_fwd(be.ptr(dc_), be.ptr(du_), None, None, be.ptr(dl), be.ptr(rows), be.ptr(loss), B, C, NR_D + 2, be.stream) == -2


def check_rows_to_f32(be, n=37):
    rng = np.random.default_rng(44)
    src = bf16_round(rng.normal(size=(n, NR_KP)).astype(np.float32))
    dst = be.poison((n, NR_D + 8), np.float32)
    before = be.np(dst).copy()
    ck(be, be.lib.nr_rows_to_f32(be.ptr(be.dev(f32_to_bf16(src))), NR_KP, NR_D, be.ptr(dst), NR_D + 8, n, be.stream))
    be.sync()
    got = be.np(dst)
    assert np.array_equal(got[:, :NR_D], src[:, :NR_D])
    assert np.array_equal(got[:, NR_D:].view(np.uint32), before[:, NR_D:].view(np.uint32))          # columns beyond d untouched


def check_rows_to_bf16(be, n=37):
    """nr_rows_to_bf16: f32 rows (strided) -> bf16 rows [n][dp], column d = 1.0, the rest 0 -- the four-columns-per-lane form (everything a
    multiple of 4) and the element form (odd width / stride), both bit-exact against round-to-nearest-even."""
    rng = np.random.default_rng(45)
    for d, dp, ld in ((900, 928, 900), (300, 320, 904), (450, 480, 451), (7, 9, 11), (300, 300, 300)):
        src = rng.normal(size=(n, ld)).astype(np.float32)
        src[0, :4] = [np.float32(1.0) + np.float32(2.0 ** -8), -0.0, 3.0e38, 1e-40]          # a tie, signed zero, near overflow, a denormal
        dst = be.poison((n, dp), np.uint16)
        ck(be, be.lib.nr_rows_to_bf16(be.ptr(be.dev(src)), ld, d, be.ptr(dst), dp, n, be.stream))
        be.sync()
        want = np.zeros((n, dp), dtype=np.uint16)
        want[:, :d] = f32_to_bf16(src[:, :d])
        if d < dp:
            want[:, d] = 0x3F80
        assert np.array_equal(be.np(dst), want), (d, dp, ld)


def check_accum_many(be, n_items=5):
    """nr_accum_many: dst[r, c] += src[r, c] over strided sub-matrices, all items in one launch; more items than one launch holds."""
    import ctypes
    rng = np.random.default_rng(43)

    class Item(ctypes.Structure):
        _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('src_ld', ctypes.c_int64), ('dst_ld', ctypes.c_int64),
                    ('rows', ctypes.c_int32), ('cols', ctypes.c_int32), ('parts', ctypes.c_int32), ('reserved', ctypes.c_int32),
                    ('part_stride', ctypes.c_int64)]
    items = (Item * n_items)()
    keep = []
    for i in range(n_items):
        rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        if i == 0:
            rows, cols = 300, 301
        if i == 1:
            rows, cols = 1, 1
        parts = 1
        if i == 2:
            rows, cols, parts = 1, 200, 256                # the query-vector gradient of a pooling level: one partial row per workgroup
        if i == 3:
            parts = 7
        sld, dld = cols + int(rng.integers(0, 9)), cols + int(rng.integers(0, 5))
        src = rng.normal(size=(parts, rows, sld)).astype(np.float32); dst = rng.normal(size=(rows, dld)).astype(np.float32)
        hs, hd = be.dev(src), be.dev(dst)
        items[i] = Item(be.ptr(hs), be.ptr(hd), sld, dld, rows, cols, parts, 0, rows * sld)
        keep.append((src, dst, hs, hd, rows, cols))
    ck(be, be.lib.nr_accum_many(ctypes.cast(items, ctypes.c_void_p), n_items, be.stream))
    be.sync()
    for src, dst, hs, hd, rows, cols in keep:
        tot = np.zeros((rows, cols), dtype=np.float32)
        P = src.shape[0]
        if P == 1:
            tot += src[0, :, :cols]
        else:                                               # the kernel's order: 16 groups of ceil(P / 16) consecutive parts, each summed in part order; then the groups in order
            per = (P + 15) // 16
            for g_ in range(16):
                sub = np.zeros((rows, cols), dtype=np.float32)
                for q in range(g_ * per, min(P, (g_ + 1) * per)):
                    sub += src[q, :, :cols]
                tot += sub
        ref = dst.copy(); ref[:, :cols] += tot
        assert np.array_equal(be.np(hd), ref)              # fp32 adds in a fixed order: bit exact; columns beyond `cols` untouched
    items[0].src_ld = 3
    assert be.lib.nr_accum_many(ctypes.cast(items, ctypes.c_void_p), n_items, be.stream) == -2


def check_mhsa_x_save(be, S=20, n_seq=7, V=300, p_drop=0.2, seed=99):
    """nr_mhsa_fwd_ex's x_save output == nr_gather_bf16 of the same ids / dropout stream (bit exact), ctx unchanged."""
    params = make_params(20, V)
    rng = np.random.default_rng(21)
    ids = rng.integers(0, V, size=(n_seq, S)).astype(np.int64)
    table = params['news_encoder.word_embedding.weight']
    Wp, bp = pack_qkv(be, params, 'news_encoder.')
    hi, ht = be.dev(ids), be.dev(table)
    sp4 = (S + 3) // 4 * 4
    outs = []
    for with_x in (False, True):
        ctx = be.poison((n_seq * S, NR_KP), np.uint16)
        sv = [be.empty((n_seq * S, NR_KP), np.uint16), be.empty((n_seq * S, NR_KP), np.uint16), be.empty((n_seq, H, 20, sp4), np.uint16)]
        xs = be.poison((n_seq * S, NR_KP), np.uint16) if with_x else None
        ck(be, be.lib.nr_mhsa_fwd_ex(be.ptr(hi), be.ptr(ht), V, None, be.ptr(Wp), be.ptr(bp), be.ptr(ctx), be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]),
                                     be.ptr(xs), n_seq, S, p_drop, seed, be.stream))
        be.sync()
        outs.append((be.np(ctx), xs))
    assert np.array_equal(outs[0][0], outs[1][0])
    Xb = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_gather_bf16(be.ptr(hi), be.ptr(ht), V, None, be.ptr(Xb), n_seq * S, p_drop, seed, be.stream))
    be.sync()
    assert np.array_equal(bf16_to_f32(be.np(outs[1][1])), bf16_to_f32(be.np(Xb)))


def check_impression_metrics(be, n_impr=40, seed=5):
    """nr_impression_metrics vs the oracle's restatement of src/evaluate.py:24-42,160-168 (pinned to the reference's own metric
    functions by tests/golden/metrics.npz), incl. score ties and single-class impressions."""
    from oracle import metrics as om
    rng = np.random.default_rng(seed)
    ys, ss = [], []
    for i in range(n_impr):
        n = int(rng.integers(2, 90))
        y = rng.integers(0, 2, size=n)
        if i % 9 == 0:
            y[:] = 0
        elif i % 9 == 1:
            y[:] = 1
        else:
            y[0], y[-1] = 1, 0
        s = rng.normal(size=n).astype(np.float32)
        if i % 4 == 0:
            s = s.round(1)                                        # ties
        ys.append(y.astype(np.int32)); ss.append(s)
    ptr = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    out = be.poison((n_impr, 4), np.float32)
    ck(be, be.lib.nr_impression_metrics(be.ptr(be.dev(np.concatenate(ss))), be.ptr(be.dev(np.concatenate(ys))), be.ptr(be.dev(ptr)), be.ptr(out),
                                        n_impr, be.stream))
    be.sync()
    got = be.np(out)
    for i in range(n_impr):
        ref = om.single_impression_metrics(ys[i], ss[i].astype(np.float64))
        if np.isnan(ref[1]):                                          # all-negative impression: 0 / 0 in all four (src/evaluate.py:38,31)
            assert np.isnan(got[i]).all() and not ys[i].any(), i
            continue
        if np.isnan(ref[0]):                                          # all-positive: AUC undefined (NaN), MRR / nDCG are those of a perfect ranking
            assert np.isnan(got[i][0]) and ys[i].all(), i
        else:
            assert abs(got[i][0] - ref[0]) < 1e-5, (i, got[i], ref)   # AUC: tie handling is exact (average ranks)
        if len(np.unique(ss[i])) == len(ss[i]):                      # MRR / nDCG depend on the tie ORDER of argsort; exact without ties
            np.testing.assert_allclose(got[i][1:], ref[1:], rtol=2e-5, atol=2e-6, err_msg=str(i))
    return got


def check_dropout_mask_statistics(be):
    """The counter-based dropout generator (csrc/nr_common.h, 3 integer multiplies per 4 elements): keep rate, and no visible
    correlation between neighbouring elements, the elements of a quad, sites and seeds (1 M elements: sigma ~ 1e-3)."""
    import numpy as np
    n = 1 << 20

    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))

    for p in (0.2, 0.5):
        m1 = export_mask(be, n, p, 777, 1).astype(np.float64)
        m2 = export_mask(be, n, p, 777, 2).astype(np.float64)
        m3 = export_mask(be, n, p, 778, 1).astype(np.float64)
        sig = np.sqrt(p * (1 - p) / n)
        assert abs(m1.mean() - (1 - p)) < 4 * sig and abs(m2.mean() - (1 - p)) < 4 * sig
        for k in (1, 2, 3, 4, 300, 1200):
            assert abs(corr(m1[:-k], m1[k:])) < 5e-3, k
        assert abs(corr(m1, m2)) < 5e-3 and abs(corr(m1, m3)) < 5e-3
        q = m1.reshape(-1, 4)
        assert np.all(np.abs(q.mean(0) - (1 - p)) < 5 * 2 * sig)
        for i in range(4):
            for j in range(i + 1, 4):
                assert abs(corr(q[:, i], q[:, j])) < 6e-3, (i, j)


def check_dropout_under_step_counter(be, value=5):
    """Every kernel that draws a dropout mask folds the device step counter into its key exactly ONCE (drop_resolve, csrc/nr_common.h): with a
    counter attached (HIP-graph replays, graph.py) the site-1 mask of the separate gather pass, of both embedding scatters and of the
    encoder's own x_save must still be the mask nr_dropout_mask exports -- and a different one from the counter-less call."""
    n = 4096
    plain = export_mask(be, n, 0.2, 99, 1)
    ctr = be.dev(np.array([value], dtype=np.int32))
    ck(be, be.lib.nr_set_step_counter(be.ptr(ctr)))
    try:
        with_ctr = export_mask(be, n, 0.2, 99, 1)
        assert not np.array_equal(plain, with_ctr), 'the step counter does not reach the dropout key'
        check_gather_bf16(be, n_tokens=129, V=200)
        check_scatter_add(be, n_tokens=333, V=40)
        check_scatter_sorted(be, n_tokens=333, V=40)
        check_mhsa_x_save(be, n_seq=5)
    finally:
        ck(be, be.lib.nr_set_step_counter(None))
    assert np.array_equal(plain, export_mask(be, n, 0.2, 99, 1))


def check_additive_bwd_scale(be, S=20, n_seq=27136, chunk=2048):
    """The pooling backward (nr_additive_bwd_ex: pool2_bwd_kernel from 2,048 sequences up) at the bench's launch size against the numpy
    restatement of AdditiveAttention's autograd (additive.py:27-53) directly, the oracle evaluated in fp64 over chunks of sequences: dpre, the
    fused dctx = dpre @ Wa and the query-vector gradient summed over ALL workgroups' partial rows (a dropped tile would show in all three)."""
    params = make_params(14)
    rng = np.random.default_rng(151)
    ctx_u = np.zeros((n_seq * S, NR_KP), dtype=np.uint16)
    ctx_u[:, :NR_D] = f32_to_bf16(rng.normal(0, 0.6, size=(n_seq * S, NR_D)).astype(np.float32))
    ctx_u[:, NR_D] = 0x3F80
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    hctx = be.dev(ctx_u)
    out = be.poison((n_seq, NR_D), np.float32)
    aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), be.ptr(aw), n_seq, S, be.stream))
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    nwg = be.lib.nr_additive_bwd_grid(n_seq, S)
    dpre = be.empty((n_seq * S, NR_QP), np.uint16)
    dqp = be.poison((nwg, NR_QP), np.float32)
    a_ = 'news_encoder.additive_attention.'
    WaT = be.poison((NR_KP, 224), np.uint16)
    ck(be, be.lib.nr_pack_additive_t(be.ptr(be.dev(params[a_ + 'linear.weight'])), 200, be.ptr(WaT), be.stream))
    dctx = be.poison((n_seq * S, NR_KP), np.uint16)
    ck(be, be.lib.nr_additive_bwd_ex(be.ptr(hctx), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(aw), be.ptr(be.dev(go)),
                                     be.ptr(dpre), be.ptr(dqp), be.ptr(WaT), be.ptr(dctx), n_seq, S, be.stream))
    be.sync()
    dpre_n, dctx_n, aw_n = be.np(dpre), be.np(dctx), be.np(aw)
    W = bf16_round(params[a_ + 'linear.weight']).astype(np.float64)
    b = params[a_ + 'linear.bias'].astype(np.float64)
    qv = params[a_ + 'attention_query_vector'].astype(np.float64)
    dq_ref = np.zeros(200)
    for lo in range(0, n_seq, chunk):
        hi = min(lo + chunk, n_seq)
        x = bf16_to_f32(ctx_u[lo * S:hi * S, :NR_D]).astype(np.float64).reshape(hi - lo, S, NR_D)
        _, w, temp = onp.additive(x, W, b, qv)
        np.testing.assert_allclose(aw_n[lo:hi], w, rtol=0, atol=2e-5, err_msg=f'attention weights, seqs {lo}..{hi}')
        g = go[lo:hi].astype(np.float64)
        dw = np.einsum('bd,bsd->bs', g, x)
        ds = w * (dw - (w * dw).sum(1, keepdims=True))
        dpre_ref = ds[:, :, None] * qv[None, None, :] * (1 - temp * temp)
        dq_ref += np.einsum('bs,bsq->q', ds, temp)
        got = bf16_to_f32(dpre_n[lo * S:hi * S])
        close_bf16(got[:, :200], dpre_ref.reshape(-1, 200), f'pooling bwd dpre, seqs {lo}..{hi}', rel=2.0 ** -7, floor=2e-3)
        assert not got[:, 200:].any()
        dref = got[:, :200].astype(np.float64) @ W
        close_bf16(bf16_to_f32(dctx_n[lo * S:hi * S, :NR_D]), dref, f'pooling bwd fused dctx, seqs {lo}..{hi}', rel=2.0 ** -7, floor=1e-3)
    dq = be.np(dqp).astype(np.float64).sum(0)
    np.testing.assert_allclose(dq[:200], dq_ref, rtol=2e-3, atol=2e-4 * np.abs(dq_ref).max())


def check_additive_flat(be, S=20, n_seq=11, valid=None, seed=29):
    """nr_additive_fwd_flat (csrc/k_pool4.h: whole sequences per wave, persistent, weighted sum on the matrix core) vs the numpy restatement of
    additive.py:27-53: pooled vectors (f32 rows and the bf16 ctx-layout copy) and attention weights; groups of floor(64 / S) sequences incl. a
    ragged last group; optional valid prefix."""
    params = make_params(8)
    rng = np.random.default_rng(seed)
    V_ = S if valid is None else valid
    ctx = np.zeros((n_seq * S, NR_KP), dtype=np.float32)
    ctx[:, :NR_D] = rng.normal(0, 0.6, size=(n_seq * S, NR_D))
    ctx[:, NR_D] = 1.0
    ctx_u = f32_to_bf16(ctx)
    Wap, bap, qvp = pack_additive(be, params, 'news_encoder.')
    stride, bstride = NR_D + 4, NR_KP + 8
    out = be.poison((n_seq, stride), np.float32)
    out_b = be.poison((n_seq, bstride), np.uint16)
    aw = be.poison((n_seq, S), np.float32)
    ck(be, be.lib.nr_additive_fwd_flat(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), stride, be.ptr(out_b), bstride,
                                       be.ptr(aw), n_seq, S, V_, 200, be.stream))
    be.sync()
    a = 'news_encoder.additive_attention.'
    x = bf16_to_f32(ctx_u)[:, :NR_D].reshape(n_seq, S, NR_D).astype(np.float64)[:, :V_]
    ref, w, _ = onp.additive(x, bf16_round(params[a + 'linear.weight']).astype(np.float64), params[a + 'linear.bias'].astype(np.float64),
                             params[a + 'attention_query_vector'].astype(np.float64))
    got_w = be.np(aw)
    np.testing.assert_allclose(got_w[:, :V_], w, rtol=2e-4, atol=2e-6)
    assert not got_w[:, V_:].any()
    go = be.np(out)
    np.testing.assert_allclose(go[:, :NR_D], ref, rtol=2e-4, atol=2e-5)
    assert np.isnan(go[:, NR_D:]).all(), 'columns >= D of the f32 rows must not be written'
    gb = be.np(out_b)
    assert np.array_equal(gb[:, :NR_D], f32_to_bf16(go[:, :NR_D])), 'bf16 copy = the f32 result rounded'
    assert (gb[:, NR_D] == 0x3F80).all() and not gb[:, NR_D + 1:NR_KP].any(), 'ctx layout: col D = 1.0, rest 0'
    assert (gb[:, NR_KP:] == 0xFFFF).all()
    assert be.lib.nr_additive_fwd_flat(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), stride, None, 0, be.ptr(aw), n_seq, 15,
                                       15, 200, be.stream) != 0            # S < 16
    assert be.lib.nr_additive_fwd_flat(be.ptr(be.dev(ctx_u)), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out), stride, None, 0, be.ptr(aw), n_seq, S,
                                       V_, 208, be.stream) != 0            # query_vector_dim > 200
