"""Backend-agnostic parity checks of the split training forward (csrc/k_proj.h): nr_pack_qkv32, nr_qkv_proj_fwd, nr_attn_fwd and
nr_attn_bwd_hm against the numpy oracle and against the register-resident kernels they replace in training."""
import numpy as np

from news_recommendation_amd._capi import NR_D, NR_KP, NR_NP, NR_QP, NR_HEADS, NR_LDG, NR_QKV_HM_SEQ, NR_K16
from tests.backends import bf16_to_f32, f32_to_bf16, bf16_round
from tests import kernel_checks as kc

H = NR_HEADS
DK = 20
S = 20


def untile32(a):
    """tile32 order (include/nr_engine.h) -> row-major [3*NP][304]."""
    return np.asarray(a).reshape(3 * NR_NP // 32, NR_K16, 2, 32, 8).transpose(0, 3, 1, 2, 4).reshape(3 * NR_NP, NR_K16 * 16)


def pack_qkv32(be, params, prefix):
    m = prefix + 'multihead_self_attention.'
    Wp32 = be.poison((3 * NR_NP * NR_K16 * 16,), np.uint16)
    bp = be.poison((3 * NR_NP,), np.float32)
    hs = [be.dev(params[m + n]) for n in ('W_Q.weight', 'W_Q.bias', 'W_K.weight', 'W_K.bias', 'W_V.weight', 'W_V.bias')]
    kc.ck(be, be.lib.nr_pack_qkv32(*[be.ptr(h) for h in hs], be.ptr(Wp32), be.ptr(bp), be.stream))
    return Wp32, bp


def packed_rows():
    """packed row c of a projection -> output feature: 64-row groups of 3 heads (60 features) + 4 zero rows (csrc/k_proj.h)."""
    c = np.arange(NR_NP)
    feat = 60 * (c // 64) + c % 64
    return np.where(c % 64 < 60, feat, -1)


def check_pack32(be):
    params = kc.make_params(3)
    Wp32, bp = pack_qkv32(be, params, 'news_encoder.')
    be.sync()
    W = untile32(be.np(Wp32))
    m = 'news_encoder.multihead_self_attention.'
    feat = packed_rows()
    real = feat >= 0
    assert sorted(feat[real].tolist()) == list(range(NR_D))
    for i, n in enumerate(('W_Q', 'W_K', 'W_V')):
        blk = W[i * NR_NP:(i + 1) * NR_NP]
        assert np.array_equal(blk[real][:, :NR_D], f32_to_bf16(params[m + n + '.weight'])[feat[real]])
        assert not blk[~real].any() and not blk[:, NR_D + 2:].any()
        b = be.np(bp)[i * NR_NP:(i + 1) * NR_NP]
        assert np.array_equal(b[real], params[m + n + '.bias'][feat[real]]) and not b[~real].any()
        # columns D, D + 1: the bias as two bf16 numbers (hi + lo) -- it rides the contraction against the token rows' two 1.0 columns
        bias = params[m + n + '.bias'][feat[real]]
        hi = f32_to_bf16(bias)
        lo = f32_to_bf16(bias - bf16_to_f32(hi))
        assert np.array_equal(blk[real][:, NR_D], hi) and np.array_equal(blk[real][:, NR_D + 1], lo)


def check_pack_encoder(be):
    """nr_pack_encoder (every operand packing of one encoder in one launch) == the single entry points, bit for bit; null outputs are skipped."""
    from news_recommendation_amd._capi import NR_QP
    lib = be.lib
    params = kc.make_params(3)
    m, a = 'news_encoder.multihead_self_attention.', 'news_encoder.additive_attention.'
    W = [be.dev(params[m + n + '.weight']) for n in ('W_Q', 'W_K', 'W_V')]
    b = [be.dev(params[m + n + '.bias']) for n in ('W_Q', 'W_K', 'W_V')]
    Wa, ba, qv = be.dev(params[a + 'linear.weight']), be.dev(params[a + 'linear.bias']), be.dev(params[a + 'attention_query_vector'])
    qdim = params[a + 'linear.weight'].shape[0]
    args = [be.ptr(x) for x in (W[0], b[0], W[1], b[1], W[2], b[2])]
    pool = [be.ptr(Wa), be.ptr(ba), be.ptr(qv), qdim]
    names = ('Wp', 'bp', 'Wp32', 'bp32', 'WdX', 'Wap', 'bap', 'qvp', 'WaT')

    def bufs():
        return dict(Wp=be.poison((3 * NR_NP, NR_KP), np.uint16), bp=be.poison((3 * NR_NP,), np.float32),
                    Wp32=be.poison((3 * NR_NP * NR_K16 * 16,), np.uint16), bp32=be.poison((3 * NR_NP,), np.float32),
                    WdX=be.poison((60 * 10 * 64 * 8,), np.uint16), Wap=be.poison((NR_QP, NR_KP), np.uint16), bap=be.poison((NR_QP,), np.float32),
                    qvp=be.poison((NR_QP,), np.float32), WaT=be.poison((NR_KP, 224), np.uint16))
    A, B, C = bufs(), bufs(), bufs()
    kc.ck(be, lib.nr_pack_qkv(*args, be.ptr(A['Wp']), be.ptr(A['bp']), be.stream))
    kc.ck(be, lib.nr_pack_qkv32(*args, be.ptr(A['Wp32']), be.ptr(A['bp32']), be.stream))
    kc.ck(be, lib.nr_pack_qkv_dx(be.ptr(W[0]), be.ptr(W[1]), be.ptr(W[2]), be.ptr(A['WdX']), be.stream))
    kc.ck(be, lib.nr_pack_additive(*pool, be.ptr(A['Wap']), be.ptr(A['bap']), be.ptr(A['qvp']), be.stream))
    kc.ck(be, lib.nr_pack_additive_t(be.ptr(Wa), qdim, be.ptr(A['WaT']), be.stream))
    kc.ck(be, lib.nr_pack_encoder(*args, *pool, *[be.ptr(B[k]) for k in names], be.stream))
    be.sync()
    for k in names:
        assert np.array_equal(be.np(A[k]).view(np.uint8), be.np(B[k]).view(np.uint8)), k
    only = ('Wp32', 'bp32', 'Wap', 'bap', 'qvp')
    before = {k: be.np(C[k]).copy() for k in names}
    kc.ck(be, lib.nr_pack_encoder(*args, *pool, *[be.ptr(C[k]) if k in only else None for k in names], be.stream))
    be.sync()
    for k in names:
        want = be.np(A[k]) if k in only else before[k]
        assert np.array_equal(be.np(C[k]).view(np.uint8), want.view(np.uint8)), k
    assert lib.nr_pack_encoder(None, *args[1:], *pool, *[be.ptr(C[k]) for k in names], be.stream) != 0 and b'nr_pack_encoder' in lib.nr_last_error()
    assert lib.nr_pack_encoder(*args, *pool[:3], 300, *[be.ptr(C[k]) for k in names], be.stream) != 0           # query_vector_dim > 208


def hm_split(qkv_u16, n_seq):
    """head-major buffer -> float64 Q, K, V as [n_seq, H, S, dk]."""
    a = bf16_to_f32(np.asarray(qkv_u16).reshape(n_seq, H, 3, S * DK)).astype(np.float64)
    q = a[:, :, 0].reshape(n_seq, H, S, DK)
    k = a[:, :, 1].reshape(n_seq, H, S, DK)
    v = a[:, :, 2].reshape(n_seq, H, S, DK)
    return q, k, v


def hm_from_rowmajor(qs, ks, vts, n_seq):
    """q_save / k_save [n_seq*S][KP], vt_save [n_seq][H][20][S] (nr_mhsa_fwd) -> head-major uint16 buffer."""
    out = np.zeros((n_seq, H, 3, S * DK), dtype=np.uint16)
    q = np.asarray(qs)[:, :NR_D].reshape(n_seq, S, H, DK).transpose(0, 2, 1, 3)
    k = np.asarray(ks)[:, :NR_D].reshape(n_seq, S, H, DK).transpose(0, 2, 1, 3)
    out[:, :, 0] = q.reshape(n_seq, H, S * DK)
    out[:, :, 1] = k.reshape(n_seq, H, S * DK)
    out[:, :, 2] = np.asarray(vts)[:, :, :, :S].transpose(0, 1, 3, 2).reshape(n_seq, H, S * DK)        # [dv][token] -> [token][dv]
    return out.reshape(-1)


def run_proj(be, params, ids, table, p_drop=0.0, seed=0, x_save=True):
    n_seq = ids.shape[0]
    Wp32, bp = pack_qkv32(be, params, 'news_encoder.')
    qkv = be.poison((n_seq * NR_QKV_HM_SEQ,), np.uint16)
    xs = be.poison((n_seq * S, NR_KP), np.uint16) if x_save else None
    kc.ck(be, be.lib.nr_qkv_proj_fwd(be.ptr(be.dev(ids.astype(np.int64))), be.ptr(be.dev(table)), table.shape[0], be.ptr(Wp32), be.ptr(bp),
                                     be.ptr(qkv), be.ptr(xs), n_seq, S, p_drop, seed, be.stream))
    be.sync()
    return qkv, xs


def check_qkv_proj(be, n_seq=13, V=300, p_drop=0.0, seed=4321):
    """Q, K, V of nr_qkv_proj_fwd == bf16(bf16(dropout(table[ids])) @ bf16(W)^T + b); x_save == nr_gather_bf16."""
    params = kc.make_params(4, V)
    rng = np.random.default_rng(21)
    ids = rng.integers(1, V, size=(n_seq, S))
    ids[:, 13:] = 0
    ids[1] = 0
    table = params['news_encoder.word_embedding.weight']
    qkv, xs = run_proj(be, params, ids, table, p_drop, seed)
    x = table[ids].astype(np.float64)
    if p_drop > 0:
        m1 = kc.export_mask(be, n_seq * S * NR_D, p_drop, seed, 1).reshape(n_seq, S, NR_D)
        x = x * m1 * np.float32(1.0 / (1.0 - p_drop))
    xq = bf16_round(x.astype(np.float32))
    got_x = be.np(xs)
    assert np.array_equal(bf16_to_f32(got_x[:, :NR_D]), xq.reshape(-1, NR_D)), 'x_save differs from the masked bf16 token matrix'
    # K padding: columns D and D + 1 are 1.0 (the projection's bias parts multiply them; the weight-gradient GEMM reads its bias gradient from
    # column D and ignores the duplicate), the rest zero
    assert (got_x[:, NR_D] == 0x3F80).all() and (got_x[:, NR_D + 1] == 0x3F80).all() and not got_x[:, NR_D + 2:].any(), 'x_save K padding wrong'
    q, k, v = hm_split(be.np(qkv), n_seq)
    m = 'news_encoder.multihead_self_attention.'
    rels = []
    for name, t in (('W_Q', q), ('W_K', k), ('W_V', v)):
        ref = xq.astype(np.float64) @ bf16_round(params[m + name + '.weight']).astype(np.float64).T + params[m + name + '.bias']
        ref = ref.reshape(n_seq, S, H, DK).transpose(0, 2, 1, 3)
        rels.append(kc.close_bf16(t, ref, 'proj ' + name, rel=2.0 ** -7, floor=1e-3))
    return rels


def check_proj_attn(be, n_seq=9, V=300, p_drop=0.0, seed=99, with_key_len=False):
    """nr_qkv_proj_fwd + nr_attn_fwd == the quantised MHSA oracle (the check nr_mhsa_fwd's gather form passes), ctx padding included."""
    params = kc.make_params(4, V)
    rng = np.random.default_rng(22)
    ids = rng.integers(1, V, size=(n_seq, S))
    ids[:, 15:] = 0
    ids[2] = 0
    table = params['news_encoder.word_embedding.weight']
    key_len = None
    if with_key_len:
        key_len = rng.integers(1, S + 1, size=n_seq)
        key_len[0], key_len[-1] = S, 1
    qkv, _ = run_proj(be, params, ids, table, p_drop, seed, x_save=False)
    ctx = be.poison((n_seq * S, NR_KP), np.uint16)
    hl = be.dev(np.asarray(key_len, dtype=np.int32)) if key_len is not None else None
    kc.ck(be, be.lib.nr_attn_fwd(be.ptr(qkv), be.ptr(ctx), be.ptr(hl), n_seq, S, p_drop, seed, be.stream))
    be.sync()
    x = table[ids].astype(np.float64)
    mask2, scale = None, 1.0
    if p_drop > 0:
        scale = np.float32(1.0 / (1.0 - p_drop))
        m1 = kc.export_mask(be, n_seq * S * NR_D, p_drop, seed, 1).reshape(n_seq, S, NR_D)
        mask2 = kc.export_mask(be, n_seq * S * NR_D, p_drop, seed, 2).reshape(n_seq, S, NR_D)
        x = x * m1 * scale
    ref = kc.mhsa_quantized_oracle(x, params, 'news_encoder.', mask2, scale, key_len=key_len)
    return kc.assert_ctx_close(be.np(ctx), ref, f'proj + attn_fwd p={p_drop} key_len={with_key_len}')


def check_attn_fwd_matches_fused(be, n_seq=6, V=300):
    """Same saved Q / K / V^T -> nr_attn_fwd's ctx == nr_mhsa_fwd's ctx: same formulas on the same bf16 operands.  Bit for bit on the emulator;
    on the matrix core the two kernels feed a head's 20 features through different k-slots (the fused kernel shares 16-row tiles between
    neighbouring heads), so fp32 sums may round differently and flip the last bf16 bit of P or ctx: held to 2 bf16 ulps there."""
    params = kc.make_params(4, V)
    rng = np.random.default_rng(23)
    ids = rng.integers(0, V, size=(n_seq, S))
    table = params['news_encoder.word_embedding.weight']
    ctx_ref, sv = kc.run_mhsa(be, params, 'news_encoder.', S, n_seq, ids=ids, table=table, save=True)
    qkv = hm_from_rowmajor(be.np(sv[0]), be.np(sv[1]), be.np(sv[2]), n_seq)
    ctx = be.poison((n_seq * S, NR_KP), np.uint16)
    kc.ck(be, be.lib.nr_attn_fwd(be.ptr(be.dev(qkv)), be.ptr(ctx), None, n_seq, S, 0.0, 0, be.stream))
    be.sync()
    got = be.np(ctx)
    if be.name == 'emu':
        assert np.array_equal(got, ctx_ref), 'attn_fwd differs from the fused kernel on identical operands'
    else:
        assert np.array_equal(got[:, NR_D:], ctx_ref[:, NR_D:])
        a, b = bf16_to_f32(got[:, :NR_D]).astype(np.float64), bf16_to_f32(ctx_ref[:, :NR_D]).astype(np.float64)
        bad = np.abs(a - b) > 2.0 ** -7 * np.abs(b) + 1e-3 * np.abs(b).max()
        assert not bad.any(), f'{bad.sum()} / {bad.size} ctx elements differ from the fused kernel by more than 2 bf16 ulps'
        assert (got[:, :NR_D] == ctx_ref[:, :NR_D]).mean() > 0.9


def check_attn_pool(be, n_seq=9, V=300, p_drop=0.0, seed=0, with_key_len=False, valid=S):
    """nr_attn_pool_fwd == nr_attn_fwd followed by nr_additive_fwd_v on the ctx it wrote: ctx bit for bit, attention weights and pooled
    vectors to fp32 rounding (the pooled form runs the same pooling code on the same bf16 tile, taken from LDS instead of HBM).  n_seq not a multiple of 4 leaves
    a partially filled workgroup whose title-less waves still take part in the pooling."""
    params = kc.make_params(4, V)
    rng = np.random.default_rng(29)
    ids = rng.integers(0, V, size=(n_seq, S))
    table = params['news_encoder.word_embedding.weight']
    qkv, _ = run_proj(be, params, ids, table, p_drop=p_drop, seed=seed)
    key_len = None
    if with_key_len:
        key_len = rng.integers(1, S + 1, size=n_seq).astype(np.int32)
        key_len[0] = S
    hl = be.dev(key_len) if key_len is not None else None
    Wap, bap, qvp = kc.pack_additive(be, params, 'news_encoder.')
    ctx0 = be.poison((n_seq * S, NR_KP), np.uint16)
    kc.ck(be, be.lib.nr_attn_fwd(be.ptr(qkv), be.ptr(ctx0), be.ptr(hl), n_seq, S, p_drop, seed, be.stream))
    out0, aw0 = be.poison((n_seq, NR_D), np.float32), be.poison((n_seq, S), np.float32)
    kc.ck(be, be.lib.nr_additive_fwd_v(be.ptr(ctx0), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out0), NR_D, None, 0, be.ptr(aw0), n_seq, S, valid,
                                       be.stream))
    ctx1 = be.poison((n_seq * S, NR_KP), np.uint16)
    out1, aw1 = be.poison((n_seq, NR_D), np.float32), be.poison((n_seq, S), np.float32)
    kc.ck(be, be.lib.nr_attn_pool_fwd(be.ptr(qkv), be.ptr(ctx1), be.ptr(hl), be.ptr(Wap), be.ptr(bap), be.ptr(qvp), be.ptr(out1), NR_D, be.ptr(aw1),
                                      n_seq, S, valid, p_drop, seed, be.stream))
    be.sync()
    assert np.array_equal(be.np(ctx1), be.np(ctx0)), 'ctx of the pooled form differs'
    # the per-token score is a sum of per-wave partial sums; 8 waves over 4 titles split the 13 x 5 (column tile, token tile) units differently
    # from the stand-alone kernel's 4 waves over 2 titles, so the fp32 additions associate differently: a few ulps of the score
    da = np.abs(be.np(aw1) - be.np(aw0)).max()
    do = np.abs(be.np(out1) - be.np(out0)).max()
    assert da <= 2e-6, f'attention weights of the pooled form differ by {da}'
    assert do <= 1e-5 * max(1.0, np.abs(be.np(out0)).max()), f'pooled vectors of the pooled form differ by {do}'
    assert np.isfinite(be.np(out1)).all() and be.np(out1).any()
    if valid < S:
        assert (be.np(aw1)[:, valid:] == 0).all()


def check_attn_bwd_hm(be, n_seq=5, p_drop=0.0, seed=77, with_key_len=False):
    """nr_attn_bwd_hm on the head-major saves == nr_attn_bwd_len on the row-major saves of the same values, bit for bit."""
    params = kc.make_params(12)
    rng = np.random.default_rng(13)
    x = rng.normal(0, 0.7, size=(n_seq, S, NR_D)).astype(np.float32)
    key_len = None
    if with_key_len:
        key_len = rng.integers(1, S + 1, size=n_seq)
        key_len[0] = S
    _, sv = kc.run_mhsa(be, params, 'user_encoder.', S, n_seq, x=x, save=True, key_len=key_len)
    dg_u = f32_to_bf16(rng.normal(0, 0.05, size=(n_seq * S, NR_D)).astype(np.float32))
    aw = rng.random(size=(n_seq, S)).astype(np.float32)
    aw /= aw.sum(1, keepdims=True)
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    hl = be.dev(np.asarray(key_len, dtype=np.int32)) if key_len is not None else None
    ref = be.empty((n_seq * S, NR_LDG), np.uint16)
    kc.ck(be, be.lib.nr_attn_bwd_len(be.ptr(sv[0]), be.ptr(sv[1]), be.ptr(sv[2]), be.ptr(be.dev(dg_u)), NR_D, be.ptr(be.dev(aw)), be.ptr(be.dev(go)),
                                     be.ptr(ref), be.ptr(hl), n_seq, S, p_drop, seed, be.stream))
    qkv = hm_from_rowmajor(be.np(sv[0]), be.np(sv[1]), be.np(sv[2]), n_seq)
    got = be.empty((n_seq * S, NR_LDG), np.uint16)
    kc.ck(be, be.lib.nr_attn_bwd_hm(be.ptr(be.dev(qkv)), be.ptr(be.dev(dg_u)), NR_D, be.ptr(be.dev(aw)), be.ptr(be.dev(go)), be.ptr(got), be.ptr(hl),
                                    n_seq, S, p_drop, seed, be.stream))
    be.sync()
    assert np.array_equal(be.np(got), be.np(ref))
    assert be.np(ref).any()


def check_dx_gemm(be, n_tok=300, seed=31):
    """nr_dx_gemm == bf16(dqkv) @ bf16([Wq; Wk; Wv]) with fp32 accumulation, rounded to bf16; padding columns exact zeros."""
    params = kc.make_params(9)
    rng = np.random.default_rng(seed)
    m = 'news_encoder.multihead_self_attention.'
    Ws = [params[m + n + '.weight'] for n in ('W_Q', 'W_K', 'W_V')]
    WdX = be.poison((60 * 10 * 64 * 8,), np.uint16)
    kc.ck(be, be.lib.nr_pack_qkv_dx(*[be.ptr(be.dev(W)) for W in Ws], be.ptr(WdX), be.stream))
    dq = np.zeros((n_tok, NR_LDG), dtype=np.float32)
    for i in range(3):
        dq[:, i * NR_KP:i * NR_KP + NR_D] = rng.normal(0, 0.3, size=(n_tok, NR_D))
    dq_u = f32_to_bf16(dq)
    dX = be.poison((n_tok, NR_KP), np.uint16)
    kc.ck(be, be.lib.nr_dx_gemm(be.ptr(be.dev(dq_u)), be.ptr(WdX), be.ptr(dX), n_tok, be.stream))
    be.sync()
    got = be.np(dX)
    ref = sum(bf16_to_f32(dq_u[:, i * NR_KP:i * NR_KP + NR_D]).astype(np.float64) @ bf16_round(Ws[i]).astype(np.float64) for i in range(3))
    kc.close_bf16(bf16_to_f32(got[:, :NR_D]), ref, 'dx_gemm', rel=2.0 ** -7, floor=1e-3)
    assert not got[:, NR_D:].any(), 'padding columns of dX must be exact zeros'
    assert be.lib.nr_dx_gemm(None, be.ptr(WdX), be.ptr(dX), n_tok, be.stream) != 0 and b'nr_dx_gemm' in be.lib.nr_last_error()
    assert be.lib.nr_dx_gemm(be.ptr(dX), be.ptr(WdX), be.ptr(dX), 0, be.stream) == 0


def check_tn_gemm(be, n_tok=300, M=NR_LDG, ldg=None, seed=41, P=None):
    """nr_tn_gemm: sum over the partitions of out[p] == G^T X (bf16 operands, fp32 accumulation); every partition is written (empty ones
    as zeros); rows >= M untouched."""
    ldg = M if ldg is None else ldg
    rng = np.random.default_rng(seed)
    G = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok, ldg)).astype(np.float32))
    X = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok, NR_KP)).astype(np.float32))
    X[:, NR_D] = 0x3F80
    P = be.lib.nr_tn_gemm_parts(M, n_tok) if P is None else P
    assert P > 0 and P % 8 == 0
    out = be.poison((P, M, NR_KP), np.float32)
    zeros = be.empty((64,), np.uint16)
    kc.ck(be, be.lib.nr_tn_gemm(be.ptr(be.dev(G)), ldg, M, be.ptr(be.dev(X)), be.ptr(zeros), be.ptr(out), n_tok, P, be.stream))
    be.sync()
    got = be.np(out).astype(np.float64)
    assert np.isfinite(got).all()
    ref = bf16_to_f32(G[:, :M]).astype(np.float64).T @ bf16_to_f32(X).astype(np.float64)
    np.testing.assert_allclose(got.sum(0), ref, rtol=0, atol=2e-4 * np.sqrt(n_tok) * 0.25 + 1e-5)
    # partition p covers tokens [p * tpp, (p + 1) * tpp): check one partition on its own
    tpp = ((n_tok + P - 1) // P + 31) // 32 * 32
    lo, hi = 0, min(tpp, n_tok)
    ref0 = bf16_to_f32(G[lo:hi, :M]).astype(np.float64).T @ bf16_to_f32(X[lo:hi]).astype(np.float64)
    np.testing.assert_allclose(got[0], ref0, rtol=0, atol=1e-4 * np.sqrt(hi - lo) + 1e-5)
    if (P - 1) * tpp >= n_tok:
        assert not got[P - 1].any(), 'an empty partition must be written as zeros'
    assert be.lib.nr_tn_gemm(None, ldg, M, be.ptr(zeros), be.ptr(zeros), be.ptr(out), n_tok, P, be.stream) != 0 and b'nr_tn_gemm' in be.lib.nr_last_error()
    assert be.lib.nr_tn_gemm(be.ptr(zeros), ldg, M, be.ptr(zeros), be.ptr(zeros), be.ptr(out), n_tok, 12, be.stream) != 0      # P not a multiple of 8


def check_proj_bad_args(be):
    a = be.empty((64,), np.float32)
    pa = be.ptr(a)
    assert be.lib.nr_qkv_proj_fwd(None, pa, 10, pa, pa, pa, None, 1, 20, 0.0, 0, be.stream) != 0 and b'nr_qkv_proj_fwd' in be.lib.nr_last_error()
    assert be.lib.nr_qkv_proj_fwd(pa, pa, 10, pa, pa, pa, None, 1, 50, 0.0, 0, be.stream) == -1
    assert be.lib.nr_qkv_proj_fwd(pa, pa, 10, pa, pa, pa, None, 1, 20, 1.0, 0, be.stream) != 0
    assert be.lib.nr_attn_fwd(None, pa, None, 1, 20, 0.0, 0, be.stream) != 0 and b'nr_attn_fwd' in be.lib.nr_last_error()
    assert be.lib.nr_attn_fwd(pa, pa, None, 1, 50, 0.0, 0, be.stream) == -1
    assert be.lib.nr_attn_bwd_hm(pa, pa, NR_D, pa, pa, pa, None, 1, 50, 0.0, 0, be.stream) == -1
    assert be.lib.nr_attn_bwd_hm(None, pa, NR_D, pa, pa, pa, None, 1, 20, 0.0, 0, be.stream) != 0
    assert be.lib.nr_qkv_proj_fwd(pa, pa, 10, pa, pa, pa, None, 0, 20, 0.0, 0, be.stream) == 0          # empty batch: nothing launched


def check_attn_bwd_hm_oracle(be, n_seq=7, p_drop=0.0, seed=78, with_key_len=False, chunk=2048):
    """nr_attn_bwd_hm against the numpy restatement of ScaledDotProductAttention's autograd (multihead_self.py:15-23) DIRECTLY -- not through
    another kernel -- on random head-major Q | K | V, at any size: the oracle runs in fp64 over chunks of sequences, so the bench's 27,136
    titles (grid caps, persistent loops, the last partial round of heads) take seconds on the host."""
    rng = np.random.default_rng(seed)
    qkv_u = f32_to_bf16(rng.normal(0, 0.8, size=(n_seq, H, 3, S * DK)).astype(np.float32))
    dg_u = np.full((n_seq * S, NR_KP), 0x7FC0, dtype=np.uint16)                 # padding columns hold NaN: never read as data
    dg_u[:, :NR_D] = f32_to_bf16(rng.normal(0, 0.05, size=(n_seq * S, NR_D)).astype(np.float32))
    aw = rng.random(size=(n_seq, S)).astype(np.float32)
    aw /= aw.sum(1, keepdims=True)
    go = rng.normal(0, 1.0, size=(n_seq, NR_D)).astype(np.float32)
    key_len = None
    if with_key_len:
        key_len = rng.integers(1, S + 1, size=n_seq).astype(np.int32)
        key_len[0] = S
    hl = be.dev(key_len) if key_len is not None else None
    got_h = be.empty((n_seq * S, NR_LDG), np.uint16)
    kc.ck(be, be.lib.nr_attn_bwd_hm(be.ptr(be.dev(qkv_u.reshape(-1))), be.ptr(be.dev(dg_u)), NR_KP, be.ptr(be.dev(aw)), be.ptr(be.dev(go)),
                                    be.ptr(got_h), be.ptr(hl), n_seq, S, p_drop, seed, be.stream))
    be.sync()
    out = be.np(got_h)
    mask2 = kc.export_mask(be, n_seq * S * NR_D, p_drop, seed, 2).reshape(n_seq, S, NR_D) if p_drop > 0 else None
    worst = [0.0, 0.0, 0.0]
    for lo in range(0, n_seq, chunk):
        hi = min(lo + chunk, n_seq)
        n = hi - lo
        q, k, v = hm_split(qkv_u[lo:hi], n)
        dC = bf16_to_f32(dg_u[lo * S:hi * S, :NR_D]).astype(np.float64).reshape(n, S, NR_D) + aw[lo:hi, :, None].astype(np.float64) * go[lo:hi, None, :]
        if mask2 is not None:
            dC = dC * mask2[lo:hi] * np.float32(1.0 / (1.0 - p_drop))
        g = bf16_round(dC.astype(np.float32)).astype(np.float64).reshape(n, S, H, DK).transpose(0, 2, 1, 3)
        e = np.exp(np.minimum(q @ np.swapaxes(k, -1, -2) / np.sqrt(np.float32(DK)).astype(np.float64), 80.0))
        if key_len is not None:
            e = e * (np.arange(S)[None, :] < key_len[lo:hi, None])[:, None, None, :]
        attn = e / (e.sum(-1, keepdims=True) + 1e-8)
        dattn = g @ np.swapaxes(v, -1, -2)
        dv = np.swapaxes(attn, -1, -2) @ g
        dS = attn * (dattn - (attn * dattn).sum(-1, keepdims=True)) / np.sqrt(DK)
        dq = dS @ k
        dkk = np.swapaxes(dS, -1, -2) @ q
        mg = lambda t: t.transpose(0, 2, 1, 3).reshape(n * S, NR_D)
        for i, (name, ref) in enumerate((('dQ', dq), ('dK', dkk), ('dV', dv))):
            blk = out[lo * S:hi * S, i * NR_KP:(i + 1) * NR_KP]
            worst[i] = max(worst[i], kc.close_bf16(bf16_to_f32(blk[:, :NR_D]), mg(ref), f'attn_bwd_hm {name} seqs {lo}..{hi}', rms_max=7.5e-3))
            assert not blk[:, NR_D:].any()
    return worst
