"""data_tools.py (dependency-free parse_behaviors / parse_news) against the reference's OWN functions (src/data_preprocess.py:22-81,
84-242): byte-identical output files on the committed raw-MIND fixture (tests/golden/data_tools, written by
oracle/make_golden_data_tools.py from the imported reference), and -- when the reference checkout is present -- on a fresh, larger
random tree.  Tokenisation is injected into the reference (nltk is absent), so it is the one thing these tests do NOT pin."""
import filecmp
import os
import random
import shutil

import pytest

from news_recommendation_amd import data_tools, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'data_tools')


def run_ours(root):
    tr, te = os.path.join(root, 'data', 'train'), os.path.join(root, 'data', 'test')
    random.seed(0)
    n_users = data_tools.parse_behaviors(os.path.join(tr, 'behaviors.tsv'), os.path.join(tr, 'behaviors_parsed.tsv'), os.path.join(tr, 'user2int.tsv'),
                                         log=lambda *_: None)
    maps = [os.path.join(tr, n) for n in ('category2int.tsv', 'word2int.tsv', 'entity2int.tsv')]
    sizes = data_tools.parse_news(os.path.join(tr, 'news.tsv'), os.path.join(tr, 'news_parsed.tsv'), *maps, mode='train', log=lambda *_: None)
    data_tools.parse_news(os.path.join(te, 'news.tsv'), os.path.join(te, 'news_parsed.tsv'), *maps, mode='test', log=lambda *_: None)
    return n_users, sizes


def assert_same_outputs(root):
    for sub, name in (('train', 'behaviors_parsed.tsv'), ('train', 'user2int.tsv'), ('train', 'category2int.tsv'), ('train', 'word2int.tsv'),
                      ('train', 'entity2int.tsv'), ('train', 'news_parsed.tsv'), ('test', 'news_parsed.tsv')):
        a, b = os.path.join(root, 'data', sub, name), os.path.join(root, 'data', sub, 'ref_' + name)
        assert filecmp.cmp(a, b, shallow=False), f"{sub}/{name} differs from the reference's output"


def test_golden_fixture_byte_identical(tmp_path):
    work = tmp_path / 'tree'
    shutil.copytree(GOLD, work)
    n_users, (ncat, nword, nent) = run_ours(str(work))
    assert (n_users, ncat, nword, nent) == (8, 10, 141, 11)
    assert_same_outputs(str(work))


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason="reference checkout not present on this box")
def test_fresh_tree_vs_imported_reference(tmp_path):
    from oracle.make_golden_data_tools import run_reference
    root = str(tmp_path)
    synth.write_raw_mind(root, n_news=120, n_users=25, n_behaviors=200, seed=11, splits=('train', 'test'))
    run_reference(root)
    run_ours(root)
    assert_same_outputs(root)


def test_parsed_files_feed_the_engine_readers(tmp_path):
    """The tree the tools write is what data_fast.TrainData / evaluate_fast.build_plan (and the reference's dataset.py) read."""
    import torch
    from news_recommendation_amd import default_config, evaluate_fast
    from news_recommendation_amd.data_fast import TrainData
    root = str(tmp_path)
    synth.write_raw_mind(root, n_news=60, n_users=10, n_behaviors=80, seed=2)
    sizes = data_tools.preprocess_tree(root)
    cfg = default_config.NAMLConfig
    data = TrainData(os.path.join(root, 'data/train/behaviors_parsed.tsv'), os.path.join(root, 'data/train/news_parsed.tsv'), cfg, torch.device('cpu'))
    assert len(data) > 0 and data.news['title'].shape[1] == 20 and data.news['abstract'].shape[1] == 50
    assert int(data.news['title'].max()) <= sizes[1] and int(data.news['category'].max()) <= sizes[0]
    b = data.batch(torch.arange(min(4, len(data))))
    from news_recommendation_amd.data_fast import split_batch
    cand, click = split_batch(b)
    assert b['ids']['title'].shape == (b['B'] * 53, 20) and cand['title'].shape[1:] == (3, 20) and click['title'].shape[1:] == (50, 20)
    plan = evaluate_fast.build_plan(os.path.join(root, 'data/val'), cfg.dataset_attributes['news'], 50,
                                    user2int_path=os.path.join(root, 'data/train/user2int.tsv'))
    assert len(plan.imp_user_row) == 80 and plan.cand_ptr[-1] == len(plan.cand_idx)


def test_tokenize_treebank_rules():
    t = data_tools.tokenize
    assert t("don't stop, it's 5,000 dollars!") == ['do', "n't", 'stop', ',', 'it', "'s", '5,000', 'dollars', '!']
    assert t('he said "hello world." then left.') == ['he', 'said', '``', 'hello', 'world', '.', "''", 'then', 'left', '.']
    assert t('mr. smith went to the u.s. in 2019') == ['mr.', 'smith', 'went', 'to', 'the', 'u.s.', 'in', '2019']
    assert t('') == [] and t('   ') == []
    assert t('(really) -- cannot') == ['(', 'really', ')', '--', 'can', 'not']


def test_tokenizer_reproduces_nltk_published_doctest_vectors():
    """f4: the dependency-free tokeniser against NLTK's OWN published known-answer vectors (tokenize.doctest 'Tokenizing some test strings',
    the word_tokenize / TreebankWordTokenizer docstrings; provenance in the fixture): contractions, currency, percent, times, quotes,
    brackets, 'cannot', commas inside and after numbers, sentence-final periods of a multi-sentence text."""
    import json
    from news_recommendation_amd.data_tools import tokenize
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_tools', 'nltk_doctest_vectors.json')) as f:
        vec = json.load(f)
    assert len(vec['word_tokenize']) >= 13
    for text, want in vec['word_tokenize']:
        assert tokenize(text) == want, text
