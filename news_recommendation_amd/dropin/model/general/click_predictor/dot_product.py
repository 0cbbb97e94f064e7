"""DotProductClickPredictor -- interface of src/model/general/click_predictor/dot_product.py:4-19."""
import torch

from news_recommendation_amd import ops


class DotProductClickPredictor(torch.nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, candidate_news_vector, user_vector):
        """[batch, C, X], [batch, X] -> [batch, C]."""
        return ops.dot_score(candidate_news_vector, user_vector)
