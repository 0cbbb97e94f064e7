"""PyTorch-CPU restatement of the reference's NRMS path (TEST INFRASTRUCTURE).

Purpose: (1) an autograd-capable fp32 checker that can travel to the GPU box
(the real reference at /root/reference cannot), (2) the `cpu_baseline` leg of
bench.py -- it reproduces the reference's *op sequence* on CPU PyTorch,
including the per-position Python loop of 1+K+N news-encoder calls
(src/model/NRMS/__init__.py:38-42), so its timing is representative of the
reference's CPU PyTorch path.  Pinned against the imported reference in
tests/test_oracle_golden.py (state_dict-compatible: load the same tensors,
get the same outputs).
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


class OracleMHSA(nn.Module):
    """src/model/general/attention/multihead_self.py:26-75 (length=None path)."""

    def __init__(self, d_model, heads):
        super().__init__()
        assert d_model % heads == 0
        self.h, self.dk = heads, d_model // heads
        self.W_Q = nn.Linear(d_model, d_model)
        self.W_K = nn.Linear(d_model, d_model)
        self.W_V = nn.Linear(d_model, d_model)
        for lin in (self.W_Q, self.W_K, self.W_V):
            nn.init.xavier_uniform_(lin.weight, gain=1)

    def forward(self, x):
        B, S, _ = x.shape
        q = self.W_Q(x).view(B, S, self.h, self.dk).transpose(1, 2)
        k = self.W_K(x).view(B, S, self.h, self.dk).transpose(1, 2)
        v = self.W_V(x).view(B, S, self.h, self.dk).transpose(1, 2)
        e = torch.exp(q @ k.transpose(-1, -2) / math.sqrt(self.dk))      # :16-17
        a = e / (e.sum(dim=-1, keepdim=True) + 1e-8)                      # :20
        return (a @ v).transpose(1, 2).contiguous().view(B, S, self.h * self.dk)


class OracleAdditive(nn.Module):
    """src/model/general/attention/additive.py:6-53."""

    def __init__(self, qdim, d):
        super().__init__()
        self.linear = nn.Linear(d, qdim)
        self.attention_query_vector = nn.Parameter(torch.empty(qdim).uniform_(-0.1, 0.1))

    def forward(self, x):
        t = torch.tanh(self.linear(x))
        w = F.softmax(t @ self.attention_query_vector, dim=1)
        return torch.bmm(w.unsqueeze(1), x).squeeze(1)


def _dropout(x, p, training, keep):
    if keep is not None:
        return x * keep.to(x.dtype) / (1.0 - p)
    return F.dropout(x, p=p, training=training)


class OracleNewsEncoder(nn.Module):
    """src/model/NRMS/news_encoder.py:10-48."""

    def __init__(self, num_words, d, heads, qdim, p_drop):
        super().__init__()
        self.word_embedding = nn.Embedding(num_words, d, padding_idx=0)
        self.multihead_self_attention = OracleMHSA(d, heads)
        self.additive_attention = OracleAdditive(qdim, d)
        self.p = p_drop

    def forward(self, title, keep1=None, keep2=None):
        """keep1 / keep2: optional explicit keep-masks (1.0 = keep) of the two dropout sites (news_encoder.py:38-40, :43-45) -- for parity
        with the engine the same masks must be used on both sides (tests/test_trajectory_gpu.py); None: F.dropout like the reference."""
        x = _dropout(self.word_embedding(title), self.p, self.training, keep1)
        y = _dropout(self.multihead_self_attention(x), self.p, self.training, keep2)
        return self.additive_attention(y)


class OracleUserEncoder(nn.Module):
    """src/model/NRMS/user_encoder.py:6-26."""

    def __init__(self, d, heads, qdim):
        super().__init__()
        self.multihead_self_attention = OracleMHSA(d, heads)
        self.additive_attention = OracleAdditive(qdim, d)

    def forward(self, x):
        return self.additive_attention(self.multihead_self_attention(x))


class OracleNRMS(nn.Module):
    """src/model/NRMS/__init__.py:7-84 (same sub-module names => same state_dict keys)."""

    def __init__(self, num_words=70976, d=300, heads=15, qdim=200, p_drop=0.2):
        super().__init__()
        self.news_encoder = OracleNewsEncoder(num_words, d, heads, qdim, p_drop)
        self.user_encoder = OracleUserEncoder(d, heads, qdim)

    def forward(self, candidate_news, clicked_news, keeps=None):
        """keeps: optional list (one per position, candidates first) of {'title1': mask, 'title2': mask} keep-masks for the two dropout sites."""
        # the per-position loop of the reference (:38-42) is kept on purpose
        k = keeps or [{}] * (len(candidate_news) + len(clicked_news))
        C = len(candidate_news)
        cand = torch.stack([self.news_encoder(x['title'], k[j].get('title1'), k[j].get('title2')) for j, x in enumerate(candidate_news)], dim=1)
        clicked = torch.stack([self.news_encoder(x['title'], k[C + j].get('title1'), k[C + j].get('title2')) for j, x in enumerate(clicked_news)], dim=1)
        user = self.user_encoder(clicked)
        return torch.bmm(cand, user.unsqueeze(-1)).squeeze(-1)          # dot_product.py:17-18

    def get_news_vector(self, news):
        return self.news_encoder(news['title'])

    def get_user_vector(self, clicked_news_vector):
        return self.user_encoder(clicked_news_vector)

    def get_prediction(self, news_vector, user_vector):
        return torch.bmm(news_vector.unsqueeze(0), user_vector.view(1, -1, 1)).view(-1)
