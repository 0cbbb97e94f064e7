"""AdditiveAttention -- interface of src/model/general/attention/additive.py:6-53."""
import torch
import torch.nn as nn

from news_recommendation_amd import ops


class AdditiveAttention(nn.Module):
    def __init__(self, query_vector_dim, candidate_vector_dim, writer=None, tag=None, names=None):
        super().__init__()
        self.linear = nn.Linear(candidate_vector_dim, query_vector_dim)
        self.attention_query_vector = nn.Parameter(torch.empty(query_vector_dim).uniform_(-0.1, 0.1))
        # For tensorboard (additive.py:21-25)
        self.writer = writer
        self.tag = tag
        self.names = names
        self.local_step = 1

    def forward(self, candidate_vector):
        """candidate_vector: [batch, S, D] -> [batch, D].  With a ``writer`` the batch mean of the attention weights is logged every tenth call
        (additive.py:40-49)."""
        if self.writer is None:
            return ops.additive_dense(candidate_vector, self)
        target, candidate_weights = ops.additive_dense(candidate_vector, self, return_weights=True)
        assert candidate_weights.size(1) == len(self.names)
        if self.local_step % 10 == 0:
            self.writer.add_scalars(self.tag, {x: y for x, y in zip(self.names, candidate_weights.mean(dim=0))}, self.local_step)
        self.local_step += 1
        return target
