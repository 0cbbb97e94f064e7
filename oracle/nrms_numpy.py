"""NumPy restatement of the reference's NRMS hot path (forward AND backward).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Every function cites the
reference file:line (relative to /root/reference) whose arithmetic it restates.
Arithmetic runs in the dtype of the inputs (use float64 for the tolerance
anchor, float32 to mimic the reference's own precision).

Parameter dictionaries use the reference's state_dict key names
(SURVEY.md section 8 b6), e.g. ``news_encoder.word_embedding.weight``.
"""
import numpy as np

# --------------------------------------------------------------------------
# L0 primitives
# --------------------------------------------------------------------------


def linear(x, W, b):
    """nn.Linear: y = x W^T + b  (used at src/model/general/attention/multihead_self.py:53-58,
    src/model/general/attention/additive.py:35)."""
    return x @ W.T + b


def sdpa(Q, K, V, d_k):
    """ScaledDotProductAttention.forward, src/model/general/attention/multihead_self.py:15-23.

    scores = exp(Q K^T / sqrt(d_k));  attn = scores / (rowsum(scores) + 1e-8);  ctx = attn V.
    NOTE: no max-subtraction and a +1e-8 in the denominator -- not F.softmax.
    Shapes: [..., S, d].  Returns (context, attn).
    """
    scores = np.exp((Q @ np.swapaxes(K, -1, -2)) / np.sqrt(np.asarray(d_k, dtype=Q.dtype)))
    attn = scores / (scores.sum(axis=-1, keepdims=True) + np.asarray(1e-8, dtype=Q.dtype))
    return attn @ V, attn


def mhsa(x, Wq, bq, Wk, bk, Wv, bv, heads):
    """MultiHeadSelfAttention.forward with K=V=Q=x and length=None,
    src/model/general/attention/multihead_self.py:46-75.  x: [B,S,D] -> [B,S,D].
    No output projection; heads are concatenated (lines 74-76)."""
    B, S, D = x.shape
    dk = D // heads

    def split(t):  # view(B,-1,H,dk).transpose(1,2)
        return t.reshape(B, S, heads, dk).transpose(0, 2, 1, 3)

    q, k, v = split(linear(x, Wq, bq)), split(linear(x, Wk, bk)), split(linear(x, Wv, bv))
    ctx, attn = sdpa(q, k, v, dk)
    out = ctx.transpose(0, 2, 1, 3).reshape(B, S, heads * dk)
    return out, (q, k, v, attn)


def additive(x, W, b, qv):
    """AdditiveAttention.forward, src/model/general/attention/additive.py:27-53.
    temp = tanh(x W^T + b); w = softmax(temp . qv, dim=1); out = sum_i w_i x_i.
    x: [B,S,D] -> ([B,D], w [B,S], temp [B,S,Q])."""
    temp = np.tanh(linear(x, W, b))
    s = temp @ qv
    s = s - s.max(axis=1, keepdims=True)
    e = np.exp(s)
    w = e / e.sum(axis=1, keepdims=True)
    out = np.einsum('bs,bsd->bd', w, x)
    return out, w, temp


def dot_score(cand, user):
    """DotProductClickPredictor.forward, src/model/general/click_predictor/dot_product.py:8-19.
    cand [B,C,D], user [B,D] -> [B,C]."""
    return np.einsum('bcd,bd->bc', cand, user)


# --------------------------------------------------------------------------
# L1 encoders
# --------------------------------------------------------------------------

def _p(params, prefix, name):
    return params[prefix + name]


def news_encoder(ids, params, heads, p_drop=0.0, mask1=None, mask2=None, prefix='news_encoder.'):
    """NRMS NewsEncoder.forward, src/model/NRMS/news_encoder.py:27-48.
    ids: int [T,L].  Dropout (lines 38-45) is applied only when masks are given:
    maskN are 0/1 keep-masks of shape [T,L,D]; kept values are scaled by 1/(1-p)
    (F.dropout semantics).  Row 0 of the table is an ordinary row in forward
    (SURVEY.md 5.9 #4).  Returns (vec [T,D], cache for backward)."""
    table = _p(params, prefix, 'word_embedding.weight')
    x = table[ids]                                   # :38  nn.Embedding
    scale = np.asarray(1.0 / (1.0 - p_drop), dtype=table.dtype)
    if mask1 is not None:
        x = x * mask1 * scale                        # :38-40 F.dropout
    m = prefix + 'multihead_self_attention.'
    y, (q, k, v, attn) = mhsa(x, params[m + 'W_Q.weight'], params[m + 'W_Q.bias'],
                              params[m + 'W_K.weight'], params[m + 'W_K.bias'],
                              params[m + 'W_V.weight'], params[m + 'W_V.bias'], heads)  # :42
    c = y
    if mask2 is not None:
        c = y * mask2 * scale                        # :43-45 F.dropout
    a = prefix + 'additive_attention.'
    out, w, temp = additive(c, params[a + 'linear.weight'], params[a + 'linear.bias'],
                            params[a + 'attention_query_vector'])                     # :47
    cache = dict(ids=ids, x=x, q=q, k=k, v=v, attn=attn, c=c, w=w, temp=temp,
                 mask1=mask1, mask2=mask2, scale=scale)
    return out, cache


def user_encoder(x, params, heads, prefix='user_encoder.'):
    """NRMS UserEncoder.forward, src/model/NRMS/user_encoder.py:15-26: MHSA then additive, no dropout."""
    m = prefix + 'multihead_self_attention.'
    y, (q, k, v, attn) = mhsa(x, params[m + 'W_Q.weight'], params[m + 'W_Q.bias'],
                              params[m + 'W_K.weight'], params[m + 'W_K.bias'],
                              params[m + 'W_V.weight'], params[m + 'W_V.bias'], heads)
    a = prefix + 'additive_attention.'
    out, w, temp = additive(y, params[a + 'linear.weight'], params[a + 'linear.bias'],
                            params[a + 'attention_query_vector'])
    cache = dict(x=x, q=q, k=k, v=v, attn=attn, c=y, w=w, temp=temp)
    return out, cache


# --------------------------------------------------------------------------
# L2 model
# --------------------------------------------------------------------------

def nrms_forward(cand_ids, clicked_ids, params, heads, p_drop=0.0, masks=None):
    """NRMS.forward, src/model/NRMS/__init__.py:19-48.
    cand_ids int [B,C,L], clicked_ids int [B,N,L] (the reference receives them as
    lists of per-position dicts, lines 21-33; stacking along dim=1 is lines 38-42).
    masks: optional dict(cand1,cand2,click1,click2) of keep-masks.
    Returns (logits [B,C], cache)."""
    B, C, L = cand_ids.shape
    N = clicked_ids.shape[1]
    mk = masks or {}
    cv, cc = news_encoder(cand_ids.reshape(B * C, L), params, heads, p_drop,
                          mk.get('cand1'), mk.get('cand2'))
    hv, hc = news_encoder(clicked_ids.reshape(B * N, L), params, heads, p_drop,
                          mk.get('click1'), mk.get('click2'))
    cand_vec = cv.reshape(B, C, -1)
    click_vec = hv.reshape(B, N, -1)
    user_vec, uc = user_encoder(click_vec, params, heads)      # :44
    logits = dot_score(cand_vec, user_vec)                     # :46
    cache = dict(cc=cc, hc=hc, uc=uc, cand_vec=cand_vec, click_vec=click_vec, user_vec=user_vec)
    return logits, cache


def cross_entropy_target0(logits):
    """nn.CrossEntropyLoss()(y_pred, zeros) -- src/train.py:126,205-206 (positive is candidate 0).
    Returns (mean loss, dlogits)."""
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    loss = -logp[:, 0].mean()
    d = np.exp(logp)
    d[:, 0] -= 1.0
    return loss, d / logits.shape[0]


# --------------------------------------------------------------------------
# Backward (autograd of the reference, restated by hand; pinned against torch
# autograd of the imported reference in tests/test_oracle_golden.py)
# --------------------------------------------------------------------------

def additive_bwd(g_out, x, w, temp, W, qv):
    """Backward of additive(): returns (dx, dW, db, dqv)."""
    # out = sum_i w_i x_i
    dx = w[:, :, None] * g_out[:, None, :]
    dw = np.einsum('bd,bsd->bs', g_out, x)
    ds = w * (dw - (w * dw).sum(axis=1, keepdims=True))       # softmax backward
    dqv = np.einsum('bs,bsq->q', ds, temp)
    dtemp = ds[:, :, None] * qv[None, None, :]
    dpre = dtemp * (1.0 - temp * temp)                        # tanh'
    dW = np.einsum('bsq,bsd->qd', dpre, x)
    db = dpre.sum(axis=(0, 1))
    dx = dx + dpre @ W
    return dx, dW, db, dqv


def mhsa_bwd(g_y, x, q, k, v, attn, Wq, Wk, Wv, heads):
    """Backward of mhsa(): returns (dx, dWq, dbq, dWk, dbk, dWv, dbv)."""
    B, S, D = x.shape
    dk = D // heads
    inv = np.asarray(1.0 / np.sqrt(dk), dtype=x.dtype)
    g = g_y.reshape(B, S, heads, dk).transpose(0, 2, 1, 3)     # [B,H,S,dk]
    dattn = g @ np.swapaxes(v, -1, -2)                         # [B,H,S,S]
    dv = np.swapaxes(attn, -1, -2) @ g
    # attn = E/(r+eps), E = exp(s):  dS = attn * (dattn - sum_j attn*dattn)
    dS = attn * (dattn - (attn * dattn).sum(axis=-1, keepdims=True))
    dq = (dS @ k) * inv
    dkk = (np.swapaxes(dS, -1, -2) @ q) * inv

    def merge(t):
        return t.transpose(0, 2, 1, 3).reshape(B * S, D)

    dq2, dk2, dv2 = merge(dq), merge(dkk), merge(dv)
    x2 = x.reshape(B * S, D)
    dWq, dWk, dWv = dq2.T @ x2, dk2.T @ x2, dv2.T @ x2
    dbq, dbk, dbv = dq2.sum(0), dk2.sum(0), dv2.sum(0)
    dx = (dq2 @ Wq + dk2 @ Wk + dv2 @ Wv).reshape(B, S, D)
    return dx, dWq, dbq, dWk, dbk, dWv, dbv


def _acc(grads, key, val):
    grads[key] = grads.get(key, 0) + val


def news_encoder_bwd(g_out, cache, params, heads, grads, prefix='news_encoder.'):
    """Backward of news_encoder(); accumulates into `grads` (dict keyed like params).
    The embedding gradient is a scatter-add over token ids with row 0
    (padding_idx, src/model/NRMS/news_encoder.py:15-20) left untouched."""
    a = prefix + 'additive_attention.'
    m = prefix + 'multihead_self_attention.'
    dc, dWa, dba, dqv = additive_bwd(g_out, cache['c'], cache['w'], cache['temp'],
                                     params[a + 'linear.weight'], params[a + 'attention_query_vector'])
    _acc(grads, a + 'linear.weight', dWa)
    _acc(grads, a + 'linear.bias', dba)
    _acc(grads, a + 'attention_query_vector', dqv)
    if cache['mask2'] is not None:
        dc = dc * cache['mask2'] * cache['scale']
    dx, dWq, dbq, dWk, dbk, dWv, dbv = mhsa_bwd(dc, cache['x'], cache['q'], cache['k'], cache['v'],
                                                cache['attn'], params[m + 'W_Q.weight'],
                                                params[m + 'W_K.weight'], params[m + 'W_V.weight'], heads)
    for n, val in (('W_Q.weight', dWq), ('W_Q.bias', dbq), ('W_K.weight', dWk),
                   ('W_K.bias', dbk), ('W_V.weight', dWv), ('W_V.bias', dbv)):
        _acc(grads, m + n, val)
    if cache['mask1'] is not None:
        dx = dx * cache['mask1'] * cache['scale']
    table = params[prefix + 'word_embedding.weight']
    dT = grads.get(prefix + 'word_embedding.weight')
    if dT is None:
        dT = np.zeros_like(table)
    ids = cache['ids'].reshape(-1)
    np.add.at(dT, ids, dx.reshape(-1, dx.shape[-1]))
    dT[0] = 0                                                  # padding_idx=0: no gradient
    grads[prefix + 'word_embedding.weight'] = dT


def user_encoder_bwd(g_out, cache, params, heads, grads, prefix='user_encoder.'):
    a = prefix + 'additive_attention.'
    m = prefix + 'multihead_self_attention.'
    dc, dWa, dba, dqv = additive_bwd(g_out, cache['c'], cache['w'], cache['temp'],
                                     params[a + 'linear.weight'], params[a + 'attention_query_vector'])
    _acc(grads, a + 'linear.weight', dWa)
    _acc(grads, a + 'linear.bias', dba)
    _acc(grads, a + 'attention_query_vector', dqv)
    dx, dWq, dbq, dWk, dbk, dWv, dbv = mhsa_bwd(dc, cache['x'], cache['q'], cache['k'], cache['v'],
                                                cache['attn'], params[m + 'W_Q.weight'],
                                                params[m + 'W_K.weight'], params[m + 'W_V.weight'], heads)
    for n, val in (('W_Q.weight', dWq), ('W_Q.bias', dbq), ('W_K.weight', dWk),
                   ('W_K.bias', dbk), ('W_V.weight', dWv), ('W_V.bias', dbv)):
        _acc(grads, m + n, val)
    return dx


def nrms_backward(dlogits, cache, params, heads):
    """Backward of nrms_forward(): returns dict of gradients keyed like params."""
    grads = {}
    cand_vec, user_vec = cache['cand_vec'], cache['user_vec']
    B, C, D = cand_vec.shape
    d_cand = dlogits[:, :, None] * user_vec[:, None, :]
    d_user = np.einsum('bc,bcd->bd', dlogits, cand_vec)
    d_click = user_encoder_bwd(d_user, cache['uc'], params, heads, grads)
    news_encoder_bwd(d_cand.reshape(B * C, D), cache['cc'], params, heads, grads)
    news_encoder_bwd(d_click.reshape(-1, D), cache['hc'], params, heads, grads)
    return grads


# --------------------------------------------------------------------------
# Parameter helpers
# --------------------------------------------------------------------------

def nrms_param_shapes(num_words=70976, d=300, qdim=200):
    """state_dict keys/shapes of the reference NRMS."""
    sh = {'news_encoder.word_embedding.weight': (num_words, d)}
    for enc in ('news_encoder.', 'user_encoder.'):
        for w in ('W_Q', 'W_K', 'W_V'):
            sh[enc + f'multihead_self_attention.{w}.weight'] = (d, d)
            sh[enc + f'multihead_self_attention.{w}.bias'] = (d,)
        sh[enc + 'additive_attention.attention_query_vector'] = (qdim,)
        sh[enc + 'additive_attention.linear.weight'] = (qdim, d)
        sh[enc + 'additive_attention.linear.bias'] = (qdim,)
    return sh


def random_nrms_params(rng, num_words=70976, d=300, qdim=200, dtype=np.float32, emb_std=1.0):
    """Random parameters with the reference's init scales (xavier-uniform W_QKV,
    multihead_self.py:41-44; query vector U(-0.1,0.1), additive.py:19-20)."""
    out = {}
    for k, s in nrms_param_shapes(num_words, d, qdim).items():
        if k.endswith('word_embedding.weight'):
            v = rng.normal(0, emb_std, s)
        elif k.endswith('attention_query_vector'):
            v = rng.uniform(-0.1, 0.1, s)
        elif len(s) == 2:
            lim = np.sqrt(6.0 / (s[0] + s[1]))
            v = rng.uniform(-lim, lim, s)
        else:
            lim = 1.0 / np.sqrt(d)
            v = rng.uniform(-lim, lim, s)
        out[k] = v.astype(dtype)
    return out
