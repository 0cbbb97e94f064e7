"""The reference's configuration surface (src/config.py:10-69: BaseConfig, NRMSConfig, NAMLConfig, LSTURConfig) with the
reference's attribute names and default values, for runs that do not point at a reference checkout
(``train_fast.py`` / ``evaluate_fast.py`` accept ``--reference`` to use the reference's own ``config.py`` instead).
Only the knobs the three supported models read are listed."""


class BaseConfig:
    num_epochs = 2
    num_batches_show_loss = 100
    num_batches_validate = 1000
    batch_size = 128
    learning_rate = 0.0001
    num_workers = 4
    num_clicked_news_a_user = 50
    num_words_title = 20
    num_words_abstract = 50
    negative_sampling_ratio = 2
    dropout_probability = 0.2
    num_words = 1 + 70975
    num_categories = 1 + 274
    num_users = 1 + 50000
    word_embedding_dim = 300
    category_embedding_dim = 100
    query_vector_dim = 200


class NRMSConfig(BaseConfig):
    dataset_attributes = {"news": ['title'], "record": []}
    num_attention_heads = 15


class NAMLConfig(BaseConfig):
    dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}
    num_filters = 300
    window_size = 3


class LSTURConfig(BaseConfig):
    dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}
    num_filters = 300
    window_size = 3
    long_short_term_method = 'ini'
    masking_probability = 0.5
