"""PyTorch-CPU restatement of the reference's NAML path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Built from primitive ops only (matmul, tanh, softmax, relu): the convolution is restated as the three-tap
token-window matmul it is, not as nn.Conv2d.  state_dict-compatible with the reference (same sub-module and
parameter names), pinned against the imported reference in tests/test_oracle_golden.py.

Dropout: the reference draws Bernoulli masks from torch's global RNG (F.dropout).  For parity tests against
the HIP engine the same keep-masks must be used on both sides, so every encoder takes optional explicit
keep-masks (1.0 = keep); with masks=None and training=True it falls back to F.dropout like the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .nrms_torch import OracleAdditive


def _bf16_ste(t):
    """Round to bf16 with a straight-through gradient: the value the engine's MFMA operands carry (see q_operands below)."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _dropout(x, p, training, keep):
    if keep is not None:
        return x * keep.to(x.dtype) / (1.0 - p)
    return F.dropout(x, p=p, training=training)


class OracleConv(nn.Module):
    """nn.Conv2d(1, F, (window, D), padding=((window-1)/2, 0)) applied to [B,1,S,D] and squeezed
    (src/model/NAML/news_encoder.py:15-17,27-28; src/model/LSTUR/news_encoder.py:24-28,62-63):
    y[b,s,f] = bias[f] + sum_w sum_d x[b, s + w - pad, d] * weight[f,0,w,d] with zero rows outside [0,S).
    Returns [B,S,F] (the reference's [B,F,S] transposed -- it transposes right after, :36)."""

    def __init__(self, num_filters, window, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_filters, 1, window, d))
        self.bias = nn.Parameter(torch.empty(num_filters))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        bound = 1.0 / (window * d) ** 0.5
        nn.init.uniform_(self.bias, -bound, bound)
        self.window = window
        # Test knob, off by default (= the reference's arithmetic).  When True the two conv operands (token matrix, filter bank)
        # are rounded to bf16 exactly where the engine rounds them, everything else stays fp32.  Needed for GRADIENT parity at
        # tiny batch sizes only: relu'(y) flips wherever |y| is below the operand-rounding noise (~1e-3), and with a handful of
        # candidate tokens carrying most of the loss gradient a single flipped element moves a filter's gradient by >10 %.
        self.q_operands = False

    def forward(self, x):
        B, S, D = x.shape
        pad = (self.window - 1) // 2
        W = self.weight
        if self.q_operands:
            x, W = _bf16_ste(x), _bf16_ste(W)
        xp = F.pad(x, (0, 0, pad, pad))
        win = torch.cat([xp[:, w:w + S] for w in range(self.window)], dim=-1)          # [B,S,window*D]
        return win @ W.view(W.shape[0], -1).t() + self.bias


class OracleTextEncoder(nn.Module):
    """TextEncoder, src/model/NAML/news_encoder.py:9-37."""

    def __init__(self, word_embedding, d, num_filters, window, qdim, p):
        super().__init__()
        self.word_embedding = word_embedding
        self.CNN = OracleConv(num_filters, window, d)
        self.additive_attention = OracleAdditive(qdim, num_filters)
        self.p = p

    def forward(self, text, keep1=None, keep2=None):
        x = _dropout(self.word_embedding(text), self.p, self.training, keep1)           # :23-25
        y = _dropout(F.relu(self.CNN(x)), self.p, self.training, keep2)                 # :27-32
        return self.additive_attention(y)                                                # :35-36


class OracleElementEncoder(nn.Module):
    """ElementEncoder, src/model/NAML/news_encoder.py:40-47."""

    def __init__(self, embedding, din, dout):
        super().__init__()
        self.embedding = embedding
        self.linear = nn.Linear(din, dout)

    def forward(self, element):
        return F.relu(self.linear(self.embedding(element)))


class OracleNAMLNewsEncoder(nn.Module):
    """NewsEncoder, src/model/NAML/news_encoder.py:50-115 with dataset_attributes = category, subcategory, title, abstract.
    View order (title, abstract, category, subcategory) is fixed here; the reference's order depends on set iteration
     and the final additive attention is permutation invariant."""

    def __init__(self, num_words, d, num_categories, dcat, num_filters, window, qdim, p,
                 attrs=('title', 'abstract', 'category', 'subcategory')):
        """attrs = config.dataset_attributes['news']: the encoders that exist (:63-81); one view needs no final attention (:82-84)."""
        super().__init__()
        we = nn.Embedding(num_words, d, padding_idx=0)
        self.text_encoders = nn.ModuleDict({n: OracleTextEncoder(we, d, num_filters, window, qdim, p) for n in ('title', 'abstract') if n in attrs})
        ce = nn.Embedding(num_categories, dcat, padding_idx=0)
        self.element_encoders = nn.ModuleDict({n: OracleElementEncoder(ce, dcat, num_filters) for n in ('category', 'subcategory') if n in attrs})
        if len(self.text_encoders) + len(self.element_encoders) > 1:
            self.final_attention = OracleAdditive(qdim, num_filters)

    def forward(self, news, keep=None):
        keep = keep or {}
        vecs = [enc(news[n], keep.get(n + '1'), keep.get(n + '2')) for n, enc in self.text_encoders.items()]
        vecs += [enc(news[n]) for n, enc in self.element_encoders.items()]
        if len(vecs) == 1:                                                               # :108-110
            return vecs[0]
        return self.final_attention(torch.stack(vecs, dim=1))                            # :111-114


class OracleNAMLUserEncoder(nn.Module):
    """UserEncoder, src/model/NAML/user_encoder.py:5-19."""

    def __init__(self, qdim, num_filters):
        super().__init__()
        self.additive_attention = OracleAdditive(qdim, num_filters)

    def forward(self, x):
        return self.additive_attention(x)


class OracleNAML(nn.Module):
    """NAML, src/model/NAML/__init__.py:7-93."""

    def __init__(self, num_words=70976, d=300, num_categories=275, dcat=100, num_filters=300, window=3, qdim=200, p_drop=0.2,
                 attrs=('title', 'abstract', 'category', 'subcategory')):
        super().__init__()
        self.news_encoder = OracleNAMLNewsEncoder(num_words, d, num_categories, dcat, num_filters, window, qdim, p_drop, attrs)
        self.user_encoder = OracleNAMLUserEncoder(qdim, num_filters)

    def forward(self, candidate_news, clicked_news, keeps=None):
        """keeps: optional list (one per position, candidates first) of keep-mask dicts for the dropout sites."""
        k = keeps or [None] * (len(candidate_news) + len(clicked_news))
        C = len(candidate_news)
        cand = torch.stack([self.news_encoder(x, k[i]) for i, x in enumerate(candidate_news)], dim=1)      # :44-45
        clicked = torch.stack([self.news_encoder(x, k[C + i]) for i, x in enumerate(clicked_news)], dim=1)  # :47-48
        user = self.user_encoder(clicked)
        return torch.bmm(cand, user.unsqueeze(-1)).squeeze(-1)

    def get_news_vector(self, news):
        return self.news_encoder(news)

    def get_user_vector(self, clicked_news_vector):
        return self.user_encoder(clicked_news_vector)

    def get_prediction(self, news_vector, user_vector):
        return torch.bmm(news_vector.unsqueeze(0), user_vector.view(1, -1, 1)).view(-1)


def random_naml_params(seed, num_words, d, num_categories, dcat, num_filters, window, qdim, emb_std=0.5):
    """Seeded parameters under the reference's state_dict keys (shared tables appear under both keys)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    un = lambda *s, a=0.1: (torch.rand(*s, generator=g) * 2 - 1) * a
    p = {}
    we = rn(num_words, d, std=emb_std)
    ce = rn(num_categories, dcat, std=emb_std)
    we[0] = 0          # nn.Embedding(padding_idx=0) initialises row 0 to zero (pretrained_word_embedding=None path) and it never
    ce[0] = 0          # receives a gradient.  (With a random row 0 every all-pad title is one identical non-trivial token stream and a
                       # ReLU pre-activation that happens to sit within bf16 noise of 0 flips for all of them at once.)
    for n in ('title', 'abstract'):
        pre = f'news_encoder.text_encoders.{n}.'
        p[pre + 'word_embedding.weight'] = we
        p[pre + 'CNN.weight'] = rn(num_filters, 1, window, d, std=(1.0 / (window * d)) ** 0.5)
        p[pre + 'CNN.bias'] = un(num_filters, a=0.05)
        p[pre + 'additive_attention.linear.weight'] = rn(qdim, num_filters, std=(1.0 / num_filters) ** 0.5)
        p[pre + 'additive_attention.linear.bias'] = un(qdim, a=0.05)
        p[pre + 'additive_attention.attention_query_vector'] = un(qdim)
    for n in ('category', 'subcategory'):
        pre = f'news_encoder.element_encoders.{n}.'
        p[pre + 'embedding.weight'] = ce
        p[pre + 'linear.weight'] = rn(num_filters, dcat, std=(1.0 / dcat) ** 0.5)
        p[pre + 'linear.bias'] = un(num_filters, a=0.05)
    for pre in ('news_encoder.final_attention.', 'user_encoder.additive_attention.'):
        p[pre + 'linear.weight'] = rn(qdim, num_filters, std=(1.0 / num_filters) ** 0.5)
        p[pre + 'linear.bias'] = un(qdim, a=0.05)
        p[pre + 'attention_query_vector'] = un(qdim)
    return p
