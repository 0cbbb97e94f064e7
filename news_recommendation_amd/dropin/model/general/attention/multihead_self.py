"""MultiHeadSelfAttention -- interface of src/model/general/attention/multihead_self.py:26-75."""
import torch.nn as nn

from news_recommendation_amd import ops


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, d_model, num_attention_heads):
        super().__init__()
        self.d_model = d_model
        self.num_attention_heads = num_attention_heads
        assert d_model % num_attention_heads == 0          # multihead_self.py:31
        self.d_k = d_model // num_attention_heads
        self.d_v = d_model // num_attention_heads
        self.W_Q = nn.Linear(d_model, d_model)
        self.W_K = nn.Linear(d_model, d_model)
        self.W_V = nn.Linear(d_model, d_model)
        self._initialize_weights()

    def _initialize_weights(self):                          # multihead_self.py:40-44
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)

    def forward(self, Q, K=None, V=None, length=None):
        """Q: [batch, S, d_model] -> [batch, S, d_model] (heads concatenated, no output projection).
        Only the self-attention form the reference's models use (K = V = Q, no length mask) is implemented."""
        if K is not None or V is not None or length is not None:
            raise NotImplementedError("only K=V=Q with length=None is used by NRMS (SURVEY.md 5.9 #3) and implemented")
        return ops.mhsa_dense(Q, self)
