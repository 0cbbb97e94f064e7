"""LSTUR NewsEncoder -- interface of src/model/LSTUR/news_encoder.py:9-76."""
import torch
import torch.nn as nn

from news_recommendation_amd import ops, ops_conv
from ..general.attention.additive import AdditiveAttention


class NewsEncoder(torch.nn.Module):
    def __init__(self, config, pretrained_word_embedding):
        super().__init__()
        self.config = config
        ops_conv.check_conv_dims(config.word_embedding_dim, config.num_filters, config.window_size, config.query_vector_dim)
        if pretrained_word_embedding is None:
            self.word_embedding = nn.Embedding(config.num_words, config.word_embedding_dim, padding_idx=0)
        else:
            self.word_embedding = nn.Embedding.from_pretrained(pretrained_word_embedding, freeze=False, padding_idx=0)
        self.category_embedding = nn.Embedding(config.num_categories, config.num_filters, padding_idx=0)
        assert config.window_size >= 1 and config.window_size % 2 == 1
        self.title_CNN = nn.Conv2d(1, config.num_filters, (config.window_size, config.word_embedding_dim),
                                   padding=(int((config.window_size - 1) / 2), 0))
        self.title_attention = AdditiveAttention(config.query_vector_dim, config.num_filters)

    def encode(self, title, category, subcategory):
        """int64 device tensors [T, L], [T], [T] -> [T, 3 * num_filters] = [category row | subcategory row | title vector]."""
        return ops_conv.lstur_news(title, category, subcategory, self.word_embedding.weight, self.category_embedding.weight,
                                   self.title_CNN, self.title_attention, self.config.dropout_probability, self.training)

    def table_rows(self, key):
        """Rows of the embedding table attribute `key` indexes."""
        return (self.word_embedding if key == 'title' else self.category_embedding).weight.shape[0]

    def to_device(self, key, ids):
        """Host or device id tensor of attribute `key` -> contiguous device tensor; out-of-table ids raise IndexError like nn.Embedding
        (host tensors always, device tensors with NR_CHECK_IDS=1: ops.check_ids)."""
        ops.check_ids(ids, self.table_rows(key), f"{key} id")
        return ids.to(self.word_embedding.weight.device, non_blocking=True).contiguous()

    def forward(self, news):
        """news: {"category": [B], "subcategory": [B], "title": [B, L]} (CPU or GPU) -> [B, 3 * num_filters]."""
        mv = lambda k: self.to_device(k, news[k])
        return self.encode(mv('title'), mv('category'), mv('subcategory'))
