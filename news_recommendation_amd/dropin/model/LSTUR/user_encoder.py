"""LSTUR UserEncoder -- interface of src/model/LSTUR/user_encoder.py:6-45."""
import torch
import torch.nn as nn

from news_recommendation_amd import ops_gru


class UserEncoder(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        assert int(config.num_filters * 1.5) == config.num_filters * 1.5
        # parameter holder with nn.GRU's names (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0); the recurrence runs in csrc/k_gru.h
        self.gru = nn.GRU(config.num_filters * 3,
                          config.num_filters * 3 if config.long_short_term_method == 'ini' else int(config.num_filters * 1.5))

    def forward(self, user, clicked_news_length, clicked_news_vector):
        """user: [B, 3F] ('ini') or [B, 1.5F] ('con'); clicked_news_length: [B] (CPU, as the reference requires for packing);
        clicked_news_vector: [B, N, 3F] -> [B, 3F].  pack_padded_sequence semantics: the first length[b] slots are consumed."""
        if clicked_news_length.is_cuda:
            clicked_news_length.clamp_(min=1)                        # same effect; a boolean-mask assignment would synchronise the stream
        else:
            clicked_news_length[clicked_news_length == 0] = 1        # in place, like the reference (:27)
        if self.config.long_short_term_method == 'ini':
            return ops_gru.gru_last_state(clicked_news_vector, user, clicked_news_length, self.gru)
        last_hidden = ops_gru.gru_last_state(clicked_news_vector, None, clicked_news_length, self.gru)
        return torch.cat((last_hidden, user), dim=1)

    @torch.no_grad()
    def forward_rows(self, user, clicked_news_length, news_vectors, clicked_rows):
        """Evaluation-only form of forward() for histories given as ROW INDICES into a matrix of news vectors (news_vectors f32 [R, 3F] on
        the GPU, clicked_rows integer [B, N]): same result as forward(user, length, news_vectors[clicked_rows]) without materialising the
        [B, N, 3F] block (news_recommendation_amd.evaluate_fast, phase B)."""
        lengths = clicked_news_length.clone()
        lengths[lengths == 0] = 1
        if self.config.long_short_term_method == 'ini':
            return ops_gru.gru_last_state_rows(news_vectors, clicked_rows, user, lengths.numpy(), self.gru)
        last_hidden = ops_gru.gru_last_state_rows(news_vectors, clicked_rows, None, lengths.numpy(), self.gru)
        return torch.cat((last_hidden, user), dim=1)
