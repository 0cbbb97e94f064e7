"""NRMS -- interface of src/model/NRMS/__init__.py:7-84."""
import torch

from news_recommendation_amd import ops
from .news_encoder import NewsEncoder
from .user_encoder import UserEncoder
from ..general.click_predictor.dot_product import DotProductClickPredictor


class NRMS(torch.nn.Module):
    """Input 1 + K candidate news and a list of user clicked news, produce the click logits."""

    def __init__(self, config, pretrained_word_embedding=None):
        super().__init__()
        self.config = config
        self.news_encoder = NewsEncoder(config, pretrained_word_embedding)
        self.user_encoder = UserEncoder(config)
        self.click_predictor = DotProductClickPredictor()

    def forward(self, candidate_news, clicked_news):
        """candidate_news: list[1+K] of {"title": int64 [B, L]}, clicked_news: list[N] of the same
        (the DataLoader's default collation, train.py:202-203) -> [B, 1+K].

        The reference encodes the 1+K+N positions one by one (__init__.py:38-42); here all B*(1+K+N) titles are
        stacked into ONE id matrix and encoded by one kernel chain."""
        w = self.news_encoder.word_embedding.weight
        ops._require_cuda(w, "NRMS parameters")
        # the 1+K+N per-position CPU tensors cross PCIe as two pinned, asynchronous copies (ops.stack_to_device)
        cand = ops.stack_to_device([x["title"] for x in candidate_news], w.device, w.shape[0], "title token id")       # [B, C, L]
        click = ops.stack_to_device([x["title"] for x in clicked_news], w.device, w.shape[0], "title token id")        # [B, N, L]
        return self.forward_ids(cand, click)

    def forward_ids(self, cand, click, loss=False, target=None):
        """Same as forward() on already-stacked id tensors: cand int64 [B, C, L], click int64 [B, N, L]
        (host or device resident)."""
        dev = self.news_encoder.word_embedding.weight.device
        B, C, L = cand.shape
        N = click.shape[1]
        V = self.news_encoder.word_embedding.weight.shape[0]
        ops.check_ids(cand, V, "title token id")
        ops.check_ids(click, V, "title token id")
        ids = torch.cat([cand.reshape(B * C, L), click.reshape(B * N, L)], dim=0).to(dev, non_blocking=True)
        return self.forward_stacked(ids, B, C, loss=loss, target=target)

    def forward_stacked(self, ids, B, C, loss=False, target=None):
        """The engine's own batch layout (news_recommendation_amd.data_fast.TrainData builds it with one gather): ids int64 [B*C + B*N, L] on the
        device -- the candidates' titles impression-major, then the history titles.  loss=False: the click logits [B, C] of forward();
        loss=True: the training loop's scalar `criterion(y_pred, y)` (CrossEntropyLoss, mean; y = target int64 [B] on the device, None = class 0
        as train.py:205 builds it) from the fused scorer + loss kernels (ops.dot_score_ce)."""
        N = ids.shape[0] // B - C
        vec = self.news_encoder.encode_ids(ids)                               # [B*(C+N), D]
        cand_rows, click_rows = ops.split_rows(vec, B * C)           # slices whose consumers write their gradients into one buffer
        candidate_news_vector = cand_rows.view(B, C, -1)
        clicked_news_vector = click_rows.view(B, N, -1)
        user_vector = self.user_encoder(clicked_news_vector)
        if loss:
            return ops.dot_score_ce(candidate_news_vector, user_vector, target)
        return self.click_predictor(candidate_news_vector, user_vector)

    def get_news_vector(self, news):
        """{"title": [batch, L]} -> [batch, D]   (evaluate.py:198)."""
        return self.news_encoder(news)

    def get_user_vector(self, clicked_news_vector):
        """[batch, N, D] -> [batch, D]   (evaluate.py:226-230)."""
        return self.user_encoder(clicked_news_vector)

    def get_prediction(self, news_vector, user_vector):
        """[C, D], [D] -> [C]   (evaluate.py:257)."""
        return self.click_predictor(news_vector.unsqueeze(dim=0), user_vector.unsqueeze(dim=0)).squeeze(dim=0)
