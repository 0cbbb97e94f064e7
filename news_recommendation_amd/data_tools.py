"""Dependency-free restatement of the reference's preprocessing: raw MIND files -> the parsed files that
``src/dataset.py`` / ``src/evaluate.py`` (and data_fast.py / evaluate_fast.py) read.

    python -m news_recommendation_amd.data_tools --root RUN_DIR [--glove glove.840B.300d.txt] [--set negative_sampling_ratio=2 ...]

Mirrors ``src/data_preprocess.py``: ``parse_behaviors`` (:22-81), ``parse_news`` (:84-242), ``generate_word_embedding`` (:245-284) and the
``__main__`` flow (:310-360; same file names under ./data/{train,val,test}).  The reference needs ``swifter`` and ``nltk`` (both absent
here, and the MIND download): this module needs pandas / numpy only.

What is pinned and what is not (tests/test_data_tools_cpu.py):
  * everything except tokenisation -- user / category / word / entity numbering by first appearance, frequency thresholds, the balanced
    (1 positive + K shuffled negatives) sampling with ``random.shuffle`` (same draws for the same ``random.seed``), truncation at
    ``num_words_title`` / ``num_words_abstract`` INCLUDING out-of-vocabulary positions, the entity map, the written TSV formats -- is
    checked against the reference's OWN functions, imported with ``swifter`` stubbed and ``nltk.tokenize.word_tokenize`` bound to
    ``tokenize`` below (plus committed golden files for boxes without the reference checkout);
  * ``tokenize`` restates the published Penn-Treebank rules behind ``nltk.tokenize.word_tokenize`` (NLTK 3.x, unpinned in the
    reference's requirements.txt:5, not installed here).  NLTK first splits sentences with the Punkt model (a trained artefact that
    cannot be restated); here a sentence ends at ``.``, ``?`` or ``!`` followed by whitespace unless the token is a known abbreviation.
    The Treebank word rules are PINNED against NLTK's own published known-answer vectors (13 doctest examples of
    nltk/test/tokenize.doctest and the word_tokenize / TreebankWordTokenizer docstrings, tests/golden/data_tools/nltk_doctest_vectors.json,
    tests/test_data_tools_cpu.py); the sentence split remains a restated heuristic, so vocabularies built by this module are
    self-consistent (train / val / test use the same function) but may differ from ones built with NLTK on texts whose sentence
    boundaries Punkt places differently (abbreviations it has learnt, ellipses).
"""
import argparse
import csv
import json
import random
import re
from os import path

import numpy as np
import pandas as pd

# ---- tokeniser: Penn Treebank rules (Robert MacIntyre's sed script, as restated in NLTK's TreebankWordTokenizer / NLTKWordTokenizer) -----
_STARTING_QUOTES = [(re.compile(r'^\"'), r'``'), (re.compile(r'(``)'), r' \1 '), (re.compile(r"([ \(\[{<])(\"|\'{2})"), r'\1 `` '),
                    (re.compile(r"(?i)(\')(?!re|ve|ll|m|t|s|d|n)(\w)\b", re.U), r'\1 \2')]
_PUNCTUATION = [(re.compile(r'([^\.])(\.)([\]\)}>"\']*)\s*$', re.U), r'\1 \2 \3 '), (re.compile(r'([:,])([^\d])'), r' \1 \2'),
                (re.compile(r'([:,])$'), r' \1 '), (re.compile(r'\.{2,}', re.U), r' \g<0> '), (re.compile(r'[;@#$%&]'), r' \g<0> '),
                (re.compile(r'([^\.])(\.)([\]\)}>"\']*)\s*$'), r'\1 \2\3 '), (re.compile(r'[?!]'), r' \g<0> '),
                (re.compile(r"([^'])' "), r"\1 ' "), (re.compile(r'[*]', re.U), r' \g<0> ')]
_PARENS = (re.compile(r'[\]\[\(\)\{\}\<\>]'), r' \g<0> ')
_DASHES = (re.compile(r'--'), r' -- ')
_ENDING_QUOTES = [(re.compile(r"([»”’])", re.U), r' \1 '), (re.compile(r"''"), " '' "), (re.compile(r'"'), " '' "),
                  (re.compile(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r'\1 \2 '),
                  (re.compile(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r'\1 \2 ')]
_CONTRACTIONS = [re.compile(p) for p in (r'(?i)\b(can)(?#X)(not)\b', r"(?i)\b(d)(?#X)('ye)\b", r'(?i)\b(gim)(?#X)(me)\b', r'(?i)\b(gon)(?#X)(na)\b',
                                         r'(?i)\b(got)(?#X)(ta)\b', r'(?i)\b(lem)(?#X)(me)\b', r"(?i)\b(more)(?#X)('n)\b",
                                         r'(?i)\b(wan)(?#X)(na)(?=\s)', r"(?i) ('t)(?#X)(is)\b", r"(?i) ('t)(?#X)(was)\b")]
_ABBREV = {'mr.', 'mrs.', 'ms.', 'dr.', 'prof.', 'sr.', 'jr.', 'st.', 'vs.', 'inc.', 'ltd.', 'co.', 'corp.', 'gov.', 'sen.', 'rep.', 'gen.',
           'col.', 'lt.', 'u.s.', 'u.k.', 'u.n.', 'a.m.', 'p.m.', 'no.', 'jan.', 'feb.', 'aug.', 'sept.', 'sep.', 'oct.', 'nov.', 'dec.', 'e.g.', 'i.e.'}
_SENT_END = re.compile(r'(\S*[.?!]["\')\]]*)(\s+)')


def split_sentences(text):
    """Approximation of the Punkt split (see the module docstring): break after ``.?!`` + whitespace unless the word is an abbreviation."""
    out, start = [], 0
    for m in _SENT_END.finditer(text):
        word = m.group(1).lower().rstrip('"\')]')
        if word in _ABBREV or re.fullmatch(r'[a-z]\.', word):
            continue
        out.append(text[start:m.end(1)])
        start = m.end()
    if start < len(text):
        out.append(text[start:])
    return [s for s in out if s.strip()]


def _treebank(text):
    for rx, sub in _STARTING_QUOTES:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION:
        text = rx.sub(sub, text)
    text = _PARENS[0].sub(_PARENS[1], text)
    text = _DASHES[0].sub(_DASHES[1], text)
    text = ' ' + text + ' '
    for rx, sub in _ENDING_QUOTES:
        text = rx.sub(sub, text)
    for rx in _CONTRACTIONS:
        text = rx.sub(r' \1 \2 ', text)
    return text.split()


def tokenize(text):
    """Stand-in for ``nltk.tokenize.word_tokenize`` (data_preprocess.py:10,133,143,165,170)."""
    return [tok for sent in split_sentences(text) for tok in _treebank(sent)]


# ---- parse_behaviors (data_preprocess.py:22-81) -------------------------------------------------------------------------------------------
def parse_behaviors(source, target, user2int_path, negative_sampling_ratio=2, log=print):
    """Training behaviours -> balanced samples.  Users are numbered 1.. in order of first appearance (:39-42); every impression yields
    as many samples as it has positives for which K negatives are still left, negatives drawn without replacement from
    ``random.shuffle`` of the impression's negatives (:52-66); impressions that yield none disappear (:68-69)."""
    log(f"Parse {source}")
    beh = pd.read_table(source, header=None, names=['impression_id', 'user', 'time', 'clicked_news', 'impressions'])
    beh['clicked_news'] = beh['clicked_news'].fillna(' ')
    user2int = {}
    for u in beh['user'].tolist():
        if u not in user2int:
            user2int[u] = len(user2int) + 1
    pd.DataFrame(user2int.items(), columns=['user', 'int']).to_csv(user2int_path, sep='\t', index=False)
    log(f'Please modify `num_users` in `src/config.py` into 1 + {len(user2int)}')
    rows = []
    for user, clicked, imps in zip(beh['user'].tolist(), beh['clicked_news'].tolist(), beh['impressions'].tolist()):
        imps = imps.split()
        positive = iter([x for x in imps if x.endswith('1')])
        negative = [x for x in imps if x.endswith('0')]
        random.shuffle(negative)
        negative = iter(negative)
        try:
            while True:
                pair = [next(positive)]
                for _ in range(negative_sampling_ratio):
                    pair.append(next(negative))           # a positive without K remaining negatives is dropped with its partial pair
                rows.append((user2int[user], clicked, ' '.join(e.split('-')[0] for e in pair), ' '.join(e.split('-')[1] for e in pair)))
        except StopIteration:
            pass
    pd.DataFrame(rows, columns=['user', 'clicked_news', 'candidate_news', 'clicked']).to_csv(target, sep='\t', index=False)
    return len(user2int)


# ---- parse_news (data_preprocess.py:84-242) ---------------------------------------------------------------------------------------------------
def _read_news(source):
    news = pd.read_table(source, header=None, usecols=[0, 1, 2, 3, 4, 6, 7], quoting=csv.QUOTE_NONE,
                         names=['id', 'category', 'subcategory', 'title', 'abstract', 'title_entities', 'abstract_entities'])
    news['title_entities'] = news['title_entities'].fillna('[]')
    news['abstract_entities'] = news['abstract_entities'].fillna('[]')
    return news.fillna(' ')


def parse_news(source, target, category2int_path, word2int_path, entity2int_path, mode, num_words_title=20, num_words_abstract=50,
               word_freq_threshold=1, entity_freq_threshold=2, entity_confidence_threshold=0.5, tokenizer=tokenize, log=print):
    """mode 'train': build the category / word / entity numberings (by first appearance; words with frequency >= word_freq_threshold,
    entities with confidence-weighted occurrence count >= entity_freq_threshold, :155-203) and write them; mode 'test': load them.  Both:
    write ``target`` with the id-encoded news (:108-153)."""
    log(f"Parse {source}")
    news = _read_news(source)
    records = list(news.itertuples(index=False))
    if mode == 'train':
        category2int, word2freq, entity2freq = {}, {}, {}
        for r in records:
            for c in (r.category, r.subcategory):
                if c not in category2int:
                    category2int[c] = len(category2int) + 1
            for text in (r.title, r.abstract):
                for w in tokenizer(text.lower()):
                    word2freq[w] = word2freq.get(w, 0) + 1
            for field in (r.title_entities, r.abstract_entities):
                for e in json.loads(field):
                    times = len(e['OccurrenceOffsets']) * e['Confidence']
                    if times > 0:
                        entity2freq[e['WikidataId']] = entity2freq.get(e['WikidataId'], 0) + times
        word2int = {}
        for k, v in word2freq.items():
            if v >= word_freq_threshold:
                word2int[k] = len(word2int) + 1
        entity2int = {}
        for k, v in entity2freq.items():
            if v >= entity_freq_threshold:
                entity2int[k] = len(entity2int) + 1
    elif mode == 'test':
        category2int = dict(pd.read_table(category2int_path).values.tolist())
        word2int = dict(pd.read_table(word2int_path, na_filter=False).values.tolist())        # "nan" is a valid word (:211-213)
        entity2int = dict(pd.read_table(entity2int_path).values.tolist())
    else:
        raise ValueError("mode must be 'train' or 'test'")

    def encode(text, limit, local_entity_map):
        words, ents = [0] * limit, [0] * limit
        for i, w in enumerate(tokenizer(text.lower())):
            if i >= limit:                    # the reference runs into IndexError here and keeps what it has (:132-149)
                break
            if w in word2int:
                words[i] = word2int[w]
                if w in local_entity_map:
                    ents[i] = local_entity_map[w]
        return words, ents

    out = []
    for r in records:
        local_entity_map = {}                 # lower-cased single word of a surface form -> entity id (:119-130)
        for field in (r.title_entities, r.abstract_entities):
            for e in json.loads(field):
                if e['Confidence'] > entity_confidence_threshold and e['WikidataId'] in entity2int:
                    for x in ' '.join(e['SurfaceForms']).lower().split():
                        local_entity_map[x] = entity2int[e['WikidataId']]
        tw, te = encode(r.title, num_words_title, local_entity_map)
        aw, ae = encode(r.abstract, num_words_abstract, local_entity_map)
        out.append((r.id, category2int.get(r.category, 0), category2int.get(r.subcategory, 0), tw, aw, te, ae))
    pd.DataFrame(out, columns=['id', 'category', 'subcategory', 'title', 'abstract', 'title_entities', 'abstract_entities']).to_csv(
        target, sep='\t', index=False)
    if mode == 'train':
        pd.DataFrame(category2int.items(), columns=['category', 'int']).to_csv(category2int_path, sep='\t', index=False)
        log(f'Please modify `num_categories` in `src/config.py` into 1 + {len(category2int)}')
        pd.DataFrame(word2int.items(), columns=['word', 'int']).to_csv(word2int_path, sep='\t', index=False)
        log(f'Please modify `num_words` in `src/config.py` into 1 + {len(word2int)}')
        pd.DataFrame(entity2int.items(), columns=['entity', 'int']).to_csv(entity2int_path, sep='\t', index=False)
        log(f'Please modify `num_entities` in `src/config.py` into 1 + {len(entity2int)}')
    return len(category2int), len(word2int), len(entity2int)


# ---- generate_word_embedding (data_preprocess.py:245-284) ----------------------------------------------------------------------------------
def generate_word_embedding(source, target, word2int_path, word_embedding_dim=300, log=print):
    """Rows of the pretrained (GloVe text format) file for the vocabulary's words; every index without one -- INCLUDING index 0, the
    padding word -- is drawn from N(0, 1) (``np.random.normal``), SURVEY.md 5.9 #4."""
    word2int = pd.read_table(word2int_path, na_filter=False, index_col='word')
    src = pd.read_table(source, index_col=0, sep=' ', header=None, quoting=csv.QUOTE_NONE, names=range(word_embedding_dim))
    src.index.rename('word', inplace=True)
    merged = word2int.merge(src, how='inner', left_index=True, right_index=True)
    merged.set_index('int', inplace=True)
    missed_index = np.setdiff1d(np.arange(len(word2int) + 1), merged.index.values)
    missed = pd.DataFrame(data=np.random.normal(size=(len(missed_index), word_embedding_dim)))
    missed['int'] = missed_index
    missed.set_index('int', inplace=True)
    final = pd.concat([merged, missed]).sort_index()
    np.save(target, final.values)
    log(f'Rate of word missed in pretrained embedding: {(len(missed_index) - 1) / len(word2int):.4f}')
    return final.values.shape


def preprocess_tree(root, glove=None, **knobs):
    """The reference's ``__main__`` flow (:310-360) under ``root``/data/{train,val,test} (entity embeddings are only used by DKN, which
    is out of scope)."""
    tr, va, te = (path.join(root, 'data', d) for d in ('train', 'val', 'test'))
    k = int(knobs.pop('negative_sampling_ratio', 2))
    parse_behaviors(path.join(tr, 'behaviors.tsv'), path.join(tr, 'behaviors_parsed.tsv'), path.join(tr, 'user2int.tsv'), k)
    maps = [path.join(tr, f) for f in ('category2int.tsv', 'word2int.tsv', 'entity2int.tsv')]
    sizes = parse_news(path.join(tr, 'news.tsv'), path.join(tr, 'news_parsed.tsv'), *maps, mode='train', **knobs)
    if glove:
        generate_word_embedding(glove, path.join(tr, 'pretrained_word_embedding.npy'), maps[1])
    for d in (va, te):
        if path.exists(path.join(d, 'news.tsv')):
            parse_news(path.join(d, 'news.tsv'), path.join(d, 'news_parsed.tsv'), *maps, mode='test', **knobs)
    return sizes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--root', default='.')
    ap.add_argument('--glove', default=None)
    ap.add_argument('--set', nargs='*', default=[], metavar='KNOB=VALUE')
    a = ap.parse_args(argv)
    knobs = {}
    for kv in a.set:
        key, v = kv.split('=', 1)
        knobs[key] = float(v) if '.' in v else int(v)
    preprocess_tree(a.root, a.glove, **knobs)


if __name__ == '__main__':
    main()
