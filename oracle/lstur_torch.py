"""PyTorch-CPU restatement of the reference's LSTUR path (TEST INFRASTRUCTURE, see oracle/__init__.py).

The GRU over the packed click history is restated as the explicit per-step gate recurrence (torch.nn.GRU's
published equations, gate order r, z, n) instead of nn.GRU + pack_padded_sequence; state_dict-compatible with
the reference (user_encoder.gru.weight_ih_l0 ...), pinned against the imported reference in
tests/test_oracle_golden.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .nrms_torch import OracleAdditive
from .naml_torch import OracleConv, _dropout


class OracleLSTURNewsEncoder(nn.Module):
    """NewsEncoder, src/model/LSTUR/news_encoder.py:9-76: cat[category_emb, subcategory_emb, conv-title-additive]."""

    def __init__(self, num_words, d, num_categories, num_filters, window, qdim, p):
        super().__init__()
        self.word_embedding = nn.Embedding(num_words, d, padding_idx=0)
        self.category_embedding = nn.Embedding(num_categories, num_filters, padding_idx=0)
        self.title_CNN = OracleConv(num_filters, window, d)
        self.title_attention = OracleAdditive(qdim, num_filters)
        self.p = p

    def forward(self, news, keep=None):
        keep = keep or {}
        cat = self.category_embedding(news['category'])                                                # :52
        sub = self.category_embedding(news['subcategory'])                                             # :54-55
        x = _dropout(self.word_embedding(news['title']), self.p, self.training, keep.get('title1'))    # :58-60
        y = _dropout(F.relu(self.title_CNN(x)), self.p, self.training, keep.get('title2'))             # :62-67
        return torch.cat([cat, sub, self.title_attention(y)], dim=1)                                   # :69-75


class OracleGRUParams(nn.Module):
    """Parameter holder with nn.GRU's names and shapes (1 layer, gate order r, z, n)."""

    def __init__(self, din, dh):
        super().__init__()
        k = 1.0 / dh ** 0.5
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * dh, din).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * dh, dh).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * dh).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * dh).uniform_(-k, k))
        self.dh = dh


class OracleLSTURUserEncoder(nn.Module):
    """UserEncoder, src/model/LSTUR/user_encoder.py:6-45.

    pack_padded_sequence(x, length, batch_first, enforce_sorted=False) + GRU returns, for sample b, the hidden state
    after consuming x[b, 0:length[b]] -- the FIRST length[b] slots of the (left-padded) history.
    'ini': h0 = user row, returns h_last.  'con': h0 = 0 with hidden size 1.5 F, returns cat(h_last, user)."""

    def __init__(self, num_filters, method='ini'):
        super().__init__()
        self.method = method
        dh = num_filters * 3 if method == 'ini' else int(num_filters * 1.5)
        self.gru = OracleGRUParams(num_filters * 3, dh)

    def forward(self, user, clicked_news_length, x):
        length = clicked_news_length.clone()
        length[length == 0] = 1                                                                         # :27
        g = self.gru
        dh = g.dh
        B, N, _ = x.shape
        h = user if self.method == 'ini' else x.new_zeros(B, dh)
        gi_all = x @ g.weight_ih_l0.t() + g.bias_ih_l0                                                  # [B,N,3dh]
        for t in range(int(length.max())):
            gi = gi_all[:, t]
            gh = h @ g.weight_hh_l0.t() + g.bias_hh_l0
            r = torch.sigmoid(gi[:, :dh] + gh[:, :dh])
            z = torch.sigmoid(gi[:, dh:2 * dh] + gh[:, dh:2 * dh])
            n = torch.tanh(gi[:, 2 * dh:] + r * gh[:, 2 * dh:])
            hn = (1 - z) * n + z * h
            live = (length > t).to(h.dtype).unsqueeze(1).to(h.device)
            h = live * hn + (1 - live) * h
        return h if self.method == 'ini' else torch.cat((h, user), dim=1)


class OracleLSTUR(nn.Module):
    """LSTUR, src/model/LSTUR/__init__.py:11-120."""

    def __init__(self, num_words=70976, d=300, num_categories=275, num_users=50001, num_filters=300, window=3, qdim=200,
                 p_drop=0.2, masking_probability=0.5, method='ini'):
        super().__init__()
        self.news_encoder = OracleLSTURNewsEncoder(num_words, d, num_categories, num_filters, window, qdim, p_drop)
        self.user_encoder = OracleLSTURUserEncoder(num_filters, method)
        self.user_embedding = nn.Embedding(num_users, num_filters * 3 if method == 'ini' else int(num_filters * 1.5), padding_idx=0)
        self.pm = masking_probability

    def forward(self, user, clicked_news_length, candidate_news, clicked_news, keeps=None, user_keep=None):
        k = keeps or [None] * (len(candidate_news) + len(clicked_news))
        C = len(candidate_news)
        cand = torch.stack([self.news_encoder(x, k[i]) for i, x in enumerate(candidate_news)], dim=1)       # :69-70
        u = self.user_embedding(user)
        if user_keep is not None:                     # F.dropout2d on [1,B,D] = whole-row masking, x 1/(1-p)  (:74-77)
            u = u * user_keep.to(u.dtype).unsqueeze(1) / (1.0 - self.pm)
        elif self.training:
            u = F.dropout1d(u.unsqueeze(0), p=self.pm, training=True).squeeze(0)
        clicked = torch.stack([self.news_encoder(x, k[C + i]) for i, x in enumerate(clicked_news)], dim=1)   # :79-80
        uv = self.user_encoder(u, clicked_news_length, clicked)
        return torch.bmm(cand, uv.unsqueeze(-1)).squeeze(-1)

    def get_news_vector(self, news):
        return self.news_encoder(news)

    def get_user_vector(self, user, clicked_news_length, clicked_news_vector):
        return self.user_encoder(self.user_embedding(user), clicked_news_length, clicked_news_vector)      # :103-105

    def get_prediction(self, news_vector, user_vector):
        return torch.bmm(news_vector.unsqueeze(0), user_vector.view(1, -1, 1)).view(-1)


def random_lstur_params(seed, num_words, d, num_categories, num_users, num_filters, window, qdim, method='ini', emb_std=0.5):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    un = lambda *s, a=0.1: (torch.rand(*s, generator=g) * 2 - 1) * a
    din = num_filters * 3
    dh = din if method == 'ini' else int(num_filters * 1.5)
    k = 1.0 / dh ** 0.5
    p = {
        'news_encoder.word_embedding.weight': rn(num_words, d, std=emb_std),
        'news_encoder.category_embedding.weight': rn(num_categories, num_filters, std=emb_std),
        'news_encoder.title_CNN.weight': rn(num_filters, 1, window, d, std=(1.0 / (window * d)) ** 0.5),
        'news_encoder.title_CNN.bias': un(num_filters, a=0.05),
        'news_encoder.title_attention.linear.weight': rn(qdim, num_filters, std=(1.0 / num_filters) ** 0.5),
        'news_encoder.title_attention.linear.bias': un(qdim, a=0.05),
        'news_encoder.title_attention.attention_query_vector': un(qdim),
        'user_encoder.gru.weight_ih_l0': un(3 * dh, din, a=k),
        'user_encoder.gru.weight_hh_l0': un(3 * dh, dh, a=k),
        'user_encoder.gru.bias_ih_l0': un(3 * dh, a=k),
        'user_encoder.gru.bias_hh_l0': un(3 * dh, a=k),
        'user_embedding.weight': rn(num_users, dh, std=emb_std),
    }
    for k in ('news_encoder.word_embedding.weight', 'news_encoder.category_embedding.weight', 'user_embedding.weight'):
        p[k][0] = 0        # padding_idx=0 rows are zero-initialised by nn.Embedding (pretrained_word_embedding=None path)
    return p
