"""NAML NewsEncoder -- interface of src/model/NAML/news_encoder.py:9-115 (TextEncoder, ElementEncoder, NewsEncoder)."""
import torch
import torch.nn as nn

from news_recommendation_amd import ops, ops_conv
from ..general.attention.additive import AdditiveAttention


class TextEncoder(torch.nn.Module):
    """src/model/NAML/news_encoder.py:9-37.  Parameter holder with the reference's names; the math runs fused inside
    NewsEncoder.encode (conv3 + pooling kernels), so calling it alone encodes one text view."""

    def __init__(self, word_embedding, word_embedding_dim, num_filters, window_size, query_vector_dim, dropout_probability):
        super().__init__()
        ops_conv.check_conv_dims(word_embedding_dim, num_filters, window_size, query_vector_dim)
        self.word_embedding = word_embedding
        self.dropout_probability = dropout_probability
        self.CNN = nn.Conv2d(1, num_filters, (window_size, word_embedding_dim), padding=(int((window_size - 1) / 2), 0))
        self.additive_attention = AdditiveAttention(query_vector_dim, num_filters)

    def forward(self, text):
        """text: int64 [batch, num_words_text] -> [batch, num_filters]."""
        dev = self.word_embedding.weight.device
        return ops_conv.text_only(text.to(dev, non_blocking=True), self.word_embedding.weight, self.CNN, self.additive_attention,
                                  self.dropout_probability, self.training)


class ElementEncoder(torch.nn.Module):
    """src/model/NAML/news_encoder.py:40-47 (parameter holder; evaluated per category row inside NewsEncoder.encode)."""

    def __init__(self, embedding, linear_input_dim, linear_output_dim):
        super().__init__()
        self.embedding = embedding
        self.linear = nn.Linear(linear_input_dim, linear_output_dim)

    def forward(self, element):
        """element: int64 [batch] -> [batch, linear_output_dim] = relu(linear(embedding(element)))   (:46-47)."""
        dev = self.linear.weight.device
        return ops_conv.element_only(element.to(dev, non_blocking=True), self.embedding, self.linear)


class NewsEncoder(torch.nn.Module):
    def __init__(self, config, pretrained_word_embedding):
        super().__init__()
        self.config = config
        attrs = set(config.dataset_attributes['news'])
        if attrs != {'category', 'subcategory', 'title', 'abstract'}:
            raise NotImplementedError("the fused NAML news encoder implements the reference's NAMLConfig view set "
                                      "(category, subcategory, title, abstract); got " + str(sorted(attrs)))
        if pretrained_word_embedding is None:
            word_embedding = nn.Embedding(config.num_words, config.word_embedding_dim, padding_idx=0)
        else:
            word_embedding = nn.Embedding.from_pretrained(pretrained_word_embedding, freeze=False, padding_idx=0)
        self.text_encoders = nn.ModuleDict({
            name: TextEncoder(word_embedding, config.word_embedding_dim, config.num_filters, config.window_size,
                              config.query_vector_dim, config.dropout_probability)
            for name in ('title', 'abstract')})
        category_embedding = nn.Embedding(config.num_categories, config.category_embedding_dim, padding_idx=0)
        self.element_encoders = nn.ModuleDict({
            name: ElementEncoder(category_embedding, config.category_embedding_dim, config.num_filters)
            for name in ('category', 'subcategory')})
        self.final_attention = AdditiveAttention(config.query_vector_dim, config.num_filters)

    def encode(self, title, abstract, category, subcategory):
        """int64 device tensors [T, Lt], [T, La], [T], [T] -> (news vectors f32 [T, F], their bf16 ctx-row copy)."""
        te, ee = self.text_encoders, self.element_encoders
        return ops_conv.naml_news(title, abstract, category, subcategory, te['title'].word_embedding.weight,
                                  ee['category'].embedding.weight, te['title'], te['abstract'], ee['category'], ee['subcategory'],
                                  self.final_attention, self.config.dropout_probability, self.training)

    def table_rows(self, key):
        """Rows of the embedding table attribute `key` indexes."""
        return (self.text_encoders['title'].word_embedding if key in ('title', 'abstract') else self.element_encoders['category'].embedding).weight.shape[0]

    def to_device(self, key, ids):
        """Host or device id tensor of attribute `key` -> contiguous device tensor; ids outside the embedding table raise IndexError like
        nn.Embedding (host tensors always, device tensors with NR_CHECK_IDS=1: ops.check_ids)."""
        ops.check_ids(ids, self.table_rows(key), f"{key} id")
        return ids.to(self.final_attention.linear.weight.device, non_blocking=True).contiguous()

    def forward(self, news):
        """news: {"category": [B], "subcategory": [B], "title": [B, Lt], "abstract": [B, La]} (CPU or GPU) -> [B, num_filters]."""
        mv = lambda k: self.to_device(k, news[k])
        return self.encode(mv('title'), mv('abstract'), mv('category'), mv('subcategory'))[0]
