"""LSTUR -- interface of src/model/LSTUR/__init__.py:11-120."""
import torch
import torch.nn as nn

from news_recommendation_amd import ops, ops_gru
from .news_encoder import NewsEncoder
from .user_encoder import UserEncoder
from ..general.click_predictor.dot_product import DotProductClickPredictor

ATTRS = ('title', 'category', 'subcategory')


class LSTUR(torch.nn.Module):
    """Input 1 + K candidate news and a list of user clicked news, produce the click logits."""

    def __init__(self, config, pretrained_word_embedding=None):
        super().__init__()
        self.config = config
        self.news_encoder = NewsEncoder(config, pretrained_word_embedding)
        self.user_encoder = UserEncoder(config)
        self.click_predictor = DotProductClickPredictor()
        assert int(config.num_filters * 1.5) == config.num_filters * 1.5
        self.user_embedding = nn.Embedding(
            config.num_users,
            config.num_filters * 3 if config.long_short_term_method == 'ini' else int(config.num_filters * 1.5),
            padding_idx=0)
        self.last_user_keep = None        # the whole-row keep mask drawn by the last training forward (for parity tests)

    def _user_rows(self, user, masked):
        dev = self.user_embedding.weight.device
        ops.check_ids(user, self.user_embedding.weight.shape[0], "user id")
        ids = ops.to_device_async(user, dev)
        scale = None
        if masked:
            # F.dropout2d on the [1, B, D] user tensor (:74-77) = one Bernoulli draw per sample, survivors scaled by 1/(1-p)
            p = float(self.config.masking_probability)
            if user.is_cuda and 0.0 < p < 1.0:
                # ids already resident on the device (forward_ids / the engine's trainer): the draw stays there too -- the engine's counter-based
                # generator (site 3, one element per sample), which follows the device step counter like every other mask, so that the
                # step can be captured into a HIP graph (graph.py) and replayed with a fresh mask
                keep = torch.empty(ids.shape[0], dtype=torch.float32, device=dev)
                ops._call('nr_dropout_mask[user]', ops._lib().nr_dropout_mask, keep.data_ptr(), ids.shape[0], p, ops.new_seed(), 3, ops._stream())
                self.last_user_keep = keep
                scale = keep / (1.0 - p)
            else:
                keep = (torch.rand(ids.shape[0]) >= p).to(torch.float32)
                self.last_user_keep = keep
                scale = ops.to_device_async(keep / (1.0 - p), dev)        # a blocking copy would drain the stream every step
        return ops_gru.user_rows(ids, self.user_embedding.weight, scale)

    def forward(self, user, clicked_news_length, candidate_news, clicked_news):
        """user: [B], clicked_news_length: [B], candidate_news: list[1+K] of {"category": [B], "subcategory": [B], "title": [B, L]},
        clicked_news: list[N] of the same (train.py:183-185) -> [B, 1+K]."""
        ne = self.news_encoder
        dev = self.user_embedding.weight.device
        ops._require_cuda(self.user_embedding.weight, "LSTUR parameters")
        cand = {k: ops.stack_to_device([x[k] for x in candidate_news], dev, ne.table_rows(k), f"{k} id") for k in ATTRS}
        click = {k: ops.stack_to_device([x[k] for x in clicked_news], dev, ne.table_rows(k), f"{k} id") for k in ATTRS}
        return self.forward_ids(user, clicked_news_length, cand, click)

    def forward_ids(self, user, clicked_news_length, cand, click, loss=False, target=None):
        B, C = cand['category'].shape
        N = click['category'].shape[1]

        def flat(k):
            a, b = cand[k], click[k]
            return self.news_encoder.to_device(k, torch.cat([a.reshape(B * C, *a.shape[2:]), b.reshape(B * N, *b.shape[2:])], dim=0))
        return self.forward_stacked(user, clicked_news_length, {k: flat(k) for k in ATTRS}, B, C, loss=loss, target=target)

    def forward_stacked(self, user, clicked_news_length, ids, B, C, loss=False, target=None):
        """The engine's own batch layout (data_fast.TrainData builds it with one gather per attribute): ids = {attr: int64 [B*C + B*N, ...]} on
        the device -- the candidates impression-major, then the history.  loss=True: the training loop's scalar `criterion(y_pred, y)`
        (CrossEntropyLoss, mean; target None = class 0, train.py:186) from the fused scorer + loss kernels instead of the logits."""
        N = ids['category'].shape[0] // B - C
        vec = self.news_encoder.encode(ids['title'], ids['category'], ids['subcategory'])       # one kernel chain for all B*(C+N) news
        cand_rows, click_rows = ops.split_rows(vec, B * C)           # slices whose consumers write their gradients into one buffer
        candidate_news_vector = cand_rows.view(B, C, -1)
        clicked_news_vector = click_rows.view(B, N, -1)
        user_row = self._user_rows(user, self.training)
        user_vector = self.user_encoder(user_row, clicked_news_length, clicked_news_vector)
        if loss:
            return ops.dot_score_ce(candidate_news_vector, user_vector, target)
        return self.click_predictor(candidate_news_vector, user_vector)

    def get_news_vector(self, news):
        return self.news_encoder(news)

    def get_user_vector(self, user, clicked_news_length, clicked_news_vector):
        """[B], [B], [B, N, 3F] -> [B, 3F]   (evaluate.py:226-230; no masking at evaluation time, :103-105)."""
        return self.user_encoder(self._user_rows(user, False), clicked_news_length, clicked_news_vector)

    def get_user_vector_rows(self, user, clicked_news_length, news_vectors, clicked_rows):
        """get_user_vector for histories given as row indices into the news-vector matrix (engine-side batched evaluation only)."""
        return self.user_encoder.forward_rows(self._user_rows(user, False), clicked_news_length, news_vectors, clicked_rows)

    def get_prediction(self, news_vector, user_vector):
        return self.click_predictor(news_vector.unsqueeze(dim=0), user_vector.unsqueeze(dim=0)).squeeze(dim=0)
