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


ALL_ATTRS = ('title', 'abstract', 'category', 'subcategory')


class NewsEncoder(torch.nn.Module):
    """src/model/NAML/news_encoder.py:50-115.  ``config.dataset_attributes['news']`` selects the views (any non-empty subset of title,
    abstract, category, subcategory: :63-81).  With all four -- the reference's NAMLConfig -- the views are produced and pooled by one
    fused kernel chain (ops_conv.naml_news); a subset composes the same hand-written kernels view by view (TextEncoder /
    ElementEncoder / AdditiveAttention on their own), exactly as the reference's forward does (:100-114)."""

    def __init__(self, config, pretrained_word_embedding):
        super().__init__()
        self.config = config
        attrs = set(config.dataset_attributes['news'])
        assert len(attrs) > 0                                                # news_encoder.py:61
        unknown = attrs - set(ALL_ATTRS) - {'title_entities', 'abstract_entities'}
        if unknown:
            raise NotImplementedError(f"NAML news attributes {sorted(unknown)} are not views of the reference's NAML encoder")
        self.attrs = tuple(a for a in ALL_ATTRS if a in attrs)               # fixed order (the reference iterates a set: SURVEY 5.9 #12)
        # the fused kernel chain is the tuned geometry's; any other word_embedding_dim / num_filters / window_size / query_vector_dim composes the
        # views one by one on the general-geometry kernels, exactly as the reference's forward does (:100-114)
        self.fused = self.attrs == ALL_ATTRS and ops_conv.conv_tuned(config.word_embedding_dim, config.num_filters, config.window_size,
                                                                     config.query_vector_dim)
        if pretrained_word_embedding is None:
            word_embedding = nn.Embedding(config.num_words, config.word_embedding_dim, padding_idx=0)
        else:
            word_embedding = nn.Embedding.from_pretrained(pretrained_word_embedding, freeze=False, padding_idx=0)
        self._word_embedding = [word_embedding]                              # kept out of the module tree when no text view uses it
        self.text_encoders = nn.ModuleDict({
            name: TextEncoder(word_embedding, config.word_embedding_dim, config.num_filters, config.window_size,
                              config.query_vector_dim, config.dropout_probability)
            for name in ('title', 'abstract') if name in attrs})
        category_embedding = nn.Embedding(config.num_categories, config.category_embedding_dim, padding_idx=0)
        self._category_embedding = [category_embedding]
        self.element_encoders = nn.ModuleDict({
            name: ElementEncoder(category_embedding, config.category_embedding_dim, config.num_filters)
            for name in ('category', 'subcategory') if name in attrs})
        # the reference counts EVERY listed attribute here, title_entities / abstract_entities included (news_encoder.py:82): the module (its
        # state_dict keys and its share of the init RNG stream) exists for ['title', 'title_entities'] although forward pools a single view
        if len(config.dataset_attributes['news']) > 1:
            self.final_attention = AdditiveAttention(config.query_vector_dim, config.num_filters)

    def _device(self):
        return next(self.parameters()).device

    def encode(self, title, abstract, category, subcategory):
        """All four views, fused: int64 device tensors [T, Lt], [T, La], [T], [T] -> (news vectors f32 [T, F], their bf16 ctx-row copy)."""
        te, ee = self.text_encoders, self.element_encoders
        return ops_conv.naml_news(title, abstract, category, subcategory, te['title'].word_embedding.weight,
                                  ee['category'].embedding.weight, te['title'], te['abstract'], ee['category'], ee['subcategory'],
                                  self.final_attention, self.config.dropout_probability, self.training)

    def encode_views(self, news):
        """news: {attr: device id tensor} for self.attrs -> (news vectors f32 [T, F], bf16 ctx-row copy or None)."""
        if self.fused:
            return self.encode(news['title'], news['abstract'], news['category'], news['subcategory'])
        vectors = [enc(news[name]) for name, enc in self.text_encoders.items()]
        vectors += [enc(news[name]) for name, enc in self.element_encoders.items()]
        if len(vectors) == 1:
            return vectors[0], None
        return self.final_attention(torch.stack(vectors, dim=1)), None

    def table_rows(self, key):
        """Rows of the embedding table attribute `key` indexes."""
        return (self._word_embedding[0] if key in ('title', 'abstract') else self._category_embedding[0]).weight.shape[0]

    def to_device(self, key, ids):
        """Host or device id tensor of attribute `key` -> contiguous device tensor; ids outside the embedding table raise IndexError like
        nn.Embedding (host tensors always, device tensors with NR_CHECK_IDS=1: ops.check_ids)."""
        ops.check_ids(ids, self.table_rows(key), f"{key} id")
        return ids.to(self._device(), non_blocking=True).contiguous()

    def forward(self, news):
        """news: {"category": [B], "subcategory": [B], "title": [B, Lt], "abstract": [B, La]} restricted to the configured attributes
        (CPU or GPU) -> [B, num_filters]."""
        return self.encode_views({k: self.to_device(k, news[k]) for k in self.attrs})[0]
