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
        """Q: [batch, S, d_model] -> [batch, S, d_model] (heads concatenated, no output projection).  ``length`` (int tensor [batch]): keys at
        positions >= length[b] are masked for every query (multihead_self.py:60-70).  K = V = Q (every reference model) at d_model 300 /
        15 heads / S <= 50 runs the tuned kernels; any other geometry (d_k <= 32, S <= 64) and the cross-attention form (K or V a different
        tensor of Q's shape, :46-58) run the general-geometry path (ops_generic.py)."""
        if (K is not None and K is not Q) or (V is not None and V is not Q):
            from news_recommendation_amd import ops_generic
            import torch
            K = Q if K is None else K
            V = Q if V is None else V
            if K.shape != Q.shape or V.shape != Q.shape:
                raise NotImplementedError("MultiHeadSelfAttention: K and V must have Q's shape (the reference's mask is square, multihead_self.py:61-66)")
            ops._require_cuda(Q, "MultiHeadSelfAttention input")
            f = lambda t: t.to(device=Q.device, dtype=torch.float32)
            return ops_generic.mhsa(f(Q), f(K), f(V), self, length)
        return ops.mhsa_dense(Q, self, length)
