"""TEST INFRASTRUCTURE: golden outputs of the reference's OWN preprocessing functions (src/data_preprocess.py:22-81 parse_behaviors,
:84-242 parse_news) on a tiny synthetic raw-MIND tree, for tests/test_data_tools_cpu.py.

The reference module is imported from /root/reference/src with its two absent third-party dependencies replaced: ``swifter`` by a
pandas accessor whose ``apply`` is ``DataFrame.apply`` (that is all swifter does besides parallelising), and
``nltk.tokenize.word_tokenize`` by ``news_recommendation_amd.data_tools.tokenize`` -- so the golden files pin everything EXCEPT
tokenisation (see data_tools.py).  Run:  python oracle/make_golden_data_tools.py   (writes tests/golden/data_tools/)."""
import os
import random
import shutil
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/src'


def import_reference_preprocess():
    import pandas as pd
    from news_recommendation_amd import data_tools
    if 'swifter' not in sys.modules:
        sw = types.ModuleType('swifter')

        @pd.api.extensions.register_dataframe_accessor('swifter')
        class _Swifter:
            def __init__(self, df):
                self._df = df

            def apply(self, fn, axis=0):
                return self._df.apply(fn, axis=axis)
        sys.modules['swifter'] = sw
    nltk = types.ModuleType('nltk')
    tok = types.ModuleType('nltk.tokenize')
    tok.word_tokenize = data_tools.tokenize
    nltk.tokenize = tok
    sys.modules['nltk'], sys.modules['nltk.tokenize'] = nltk, tok
    os.environ.setdefault('MODEL_NAME', 'NRMS')
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import data_preprocess
    return data_preprocess


def run_reference(root):
    dp = import_reference_preprocess()
    tr, te = os.path.join(root, 'data', 'train'), os.path.join(root, 'data', 'test')
    random.seed(0)
    dp.parse_behaviors(os.path.join(tr, 'behaviors.tsv'), os.path.join(tr, 'ref_behaviors_parsed.tsv'), os.path.join(tr, 'ref_user2int.tsv'))
    maps = [os.path.join(tr, f'ref_{n}') for n in ('category2int.tsv', 'word2int.tsv', 'entity2int.tsv')]
    dp.parse_news(os.path.join(tr, 'news.tsv'), os.path.join(tr, 'ref_news_parsed.tsv'), *maps, mode='train')
    dp.parse_news(os.path.join(te, 'news.tsv'), os.path.join(te, 'ref_news_parsed.tsv'), *maps, mode='test')


if __name__ == '__main__':
    from news_recommendation_amd import synth
    out = os.path.join(ROOT, 'tests', 'golden', 'data_tools')
    shutil.rmtree(out, ignore_errors=True)
    synth.write_raw_mind(out, n_news=24, n_users=8, n_behaviors=30, seed=5, splits=('train', 'test'))
    run_reference(out)
    print('wrote', out)
