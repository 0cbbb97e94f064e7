"""Synthetic MIND-small-shaped inputs.  No dataset files are needed: the generator
emits the id tensors the hot path consumes (and, for the drop-in launcher, can be written out in the
reference's on-disk formats later)."""
import numpy as np

MIND_SMALL = dict(num_words=70976, num_news=65238, num_clicked=50, title_len=20, neg_k=2)

# Dataset shapes.  'small' = the sizes hard-coded in the reference's src/config.py:29-33 (MIND-small after its
# preprocessing).  'large' = MIND-large: ~161 k news, 711,222 training users (src/model/LSTUR/__init__.py:38-42 sizes user_embedding from
# num_users: 711,223 x 900 fp32 = 2.56 GB); the reference gives no MIND-large vocabulary, 1 + 130,000 words is this project's knob
# (156 MB fp32 table, still inside the 256 MB Infinity Cache).  'xlarge' only differs in the vocabulary: 1 + 250,000 words = 300 MB,
# the point at which the embedding gather and the table's gradient exchange leave the cache.
SHAPES = {
    'small': dict(num_words=1 + 70975, num_news=65238, num_users=1 + 50000, num_categories=1 + 274, eval_impressions=73152),
    'large': dict(num_words=1 + 130000, num_news=161013, num_users=1 + 711222, num_categories=1 + 274, eval_impressions=376471),
    'xlarge': dict(num_words=1 + 250000, num_news=161013, num_users=1 + 711222, num_categories=1 + 274, eval_impressions=376471),
}


def zipf_ids(rng, shape, num_words, s=1.05):
    """Zipf(s)-distributed token ids over 1..num_words-1 (rank-frequency like natural text)."""
    # inverse-CDF sampling on a truncated zeta via the continuous approximation
    u = rng.random(size=shape)
    n = num_words - 1
    if abs(s - 1.0) < 1e-6:
        r = np.exp(u * np.log(n))
    else:
        r = ((n ** (1 - s) - 1) * u + 1) ** (1 / (1 - s))
    return np.clip(r.astype(np.int64), 1, n)


def news_titles(rng, n_news, title_len=20, num_words=70976, dist='zipf'):
    """[n_news, title_len] int64: 4..title_len real tokens (mean ~11), right-padded with 0."""
    if dist == 'zipf':
        ids = zipf_ids(rng, (n_news, title_len), num_words)
    else:
        ids = rng.integers(1, num_words, size=(n_news, title_len))
    lens = np.clip(rng.normal(11, 4, size=n_news).round().astype(np.int64), 4, title_len)
    ids[np.arange(title_len)[None, :] >= lens[:, None]] = 0
    return ids.astype(np.int64)


def news_abstracts(rng, n_news, abstract_len=50, num_words=70976):
    """[n_news, abstract_len] int64: 8..abstract_len real tokens (mean ~35), right-padded with 0."""
    ids = zipf_ids(rng, (n_news, abstract_len), num_words)
    lens = np.clip(rng.normal(35, 10, size=n_news).round().astype(np.int64), 8, abstract_len)
    ids[np.arange(abstract_len)[None, :] >= lens[:, None]] = 0
    return ids.astype(np.int64)


def history_lengths(rng, n, num_clicked=50):
    """clipped lognormal history lengths in [0, num_clicked], mean ~32."""
    return np.clip(rng.lognormal(3.6, 0.8, size=n).round().astype(np.int64), 0, num_clicked)


def train_batch(rng, news, B, num_clicked=50, neg_k=2):
    """One train-shaped batch: candidate news indices [B, 1+K] (positive first, data_preprocess.py:63-66) and
    LEFT-padded history indices [B, N] with -1 for padding (dataset.py:79-83)."""
    n_news = news.shape[0]
    cand = rng.integers(0, n_news, size=(B, 1 + neg_k))
    hl = history_lengths(rng, B, num_clicked)
    hist = rng.integers(0, n_news, size=(B, num_clicked))
    hist[np.arange(num_clicked)[None, :] < (num_clicked - hl)[:, None]] = -1
    return cand, hist


def batch_token_ids(news, cand, hist):
    """news-index batches -> token id tensors [B, C, L], [B, N, L]; padded history slots are all-zero titles
    (training-time semantics)."""
    L = news.shape[1]
    pad = np.zeros((1, L), dtype=np.int64)
    tab = np.concatenate([news, pad], axis=0)
    return tab[cand], tab[np.where(hist < 0, news.shape[0], hist)]


def eval_impressions(rng, n_news, n_impr, num_clicked=50, mean_cands=37):
    """Eval-shaped impressions: per impression a left-padded history (news indices, -1 = PADDED_NEWS) and a ragged
    candidate list (lognormal length in [2, 300], mean ~37)."""
    hl = history_lengths(rng, n_impr, num_clicked)
    hist = rng.integers(0, n_news, size=(n_impr, num_clicked))
    hist[np.arange(num_clicked)[None, :] < (num_clicked - hl)[:, None]] = -1
    lens = np.clip(rng.lognormal(np.log(mean_cands) - 0.32, 0.8, size=n_impr).round().astype(np.int64), 2, 300)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cands = rng.integers(0, n_news, size=int(ptr[-1])).astype(np.int32)
    return hist, cands, ptr


def teacher_labels(rng, scores, ptr):
    """y ~ Bernoulli(sigmoid(2 z - 1.5)) on per-impression z-scored teacher scores, forced >= 1 positive and
    >= 1 negative."""
    labels = np.zeros(len(scores), dtype=np.int64)
    for i in range(len(ptr) - 1):
        s = scores[ptr[i]:ptr[i + 1]]
        z = (s - s.mean()) / (s.std() + 1e-9)
        y = (rng.random(len(s)) < 1 / (1 + np.exp(-(2 * z - 1.5)))).astype(np.int64)
        if y.sum() == 0:
            y[np.argmax(z)] = 1
        if y.sum() == len(y):
            y[np.argmin(z)] = 0
        labels[ptr[i]:ptr[i + 1]] = y
    return labels


def write_reference_dataset(root, n_news=300, n_users=40, n_train=256, n_val_impr=60, num_words=500, seed=0,
                            title_len=20, num_clicked=50, neg_k=2, learnable=False):
    """Write a tiny synthetic data tree in the reference's on-disk formats so that its UNCHANGED train.py /
    evaluate.py can run from ``root`` as cwd (file schemas: src/dataset.py:27-37, src/evaluate.py:57-65,87-93,
    135-143; writers: src/data_preprocess.py:45-49,77-81,205).  Returns the directory.
    learnable: clicks follow a rule a model can learn (default: uniformly random clicks, AUC 0.5 whatever is trained) -- every news has a
    topic (= its category; 70 % of its real tokens come from the topic's slice of the vocabulary), every user a preferred topic; 80 % of a
    history and the clicked training candidate are of that topic, and an evaluation candidate is labelled 1 with probability 0.85 / 0.08
    when its topic is / is not the user's."""
    import os
    import pandas as pd
    rng = np.random.default_rng(seed)
    titles = news_titles(rng, n_news, title_len, num_words)
    abstracts = news_titles(rng, n_news, 50, num_words)
    nid = [f'N{i + 1}' for i in range(n_news)]
    n_topics = 9
    if learnable:
        topic = rng.integers(0, n_topics, n_news)
        width = max((num_words - 1) // n_topics, 1)
        for arr in (titles, abstracts):
            own = 1 + topic[:, None] * width + rng.integers(0, width, size=arr.shape)
            arr[...] = np.where((arr != 0) & (rng.random(arr.shape) < 0.7), own, arr)
        by_topic = [np.flatnonzero(topic == t) for t in range(n_topics)]
        user_topic = rng.integers(0, n_topics, n_users + 1)
    news = pd.DataFrame({
        'id': nid, 'category': (topic + 1) if learnable else rng.integers(1, 10, n_news), 'subcategory': rng.integers(1, 30, n_news),
        'title': [str(list(map(int, t))) for t in titles],
        'abstract': [str(list(map(int, t))) for t in abstracts],
        'title_entities': [str([0] * title_len)] * n_news, 'abstract_entities': [str([0] * 50)] * n_news})
    users = [f'U{i + 1}' for i in range(n_users)]
    for split in ('train', 'val', 'test'):
        os.makedirs(os.path.join(root, 'data', split), exist_ok=True)
        news.to_csv(os.path.join(root, 'data', split, 'news_parsed.tsv'), sep='\t', index=False)
    pd.DataFrame({'user': users, 'int': np.arange(1, n_users + 1)}).to_csv(
        os.path.join(root, 'data', 'train', 'user2int.tsv'), sep='\t', index=False)

    def of_topic(t, k):
        pool = by_topic[t]
        return pool[rng.integers(0, len(pool), size=k)] if len(pool) else rng.integers(0, n_news, size=k)

    def hist_str(k, u=None):
        if k <= 0:
            return ' '
        if not learnable:
            return ' '.join(rng.choice(nid, size=k))
        idx = np.where(rng.random(k) < 0.8, of_topic(user_topic[u], k), rng.integers(0, n_news, size=k))
        return ' '.join(nid[j] for j in idx)
    rows = []
    for _ in range(n_train):
        k = int(history_lengths(rng, 1, num_clicked)[0])
        u = int(rng.integers(1, n_users + 1))
        h = hist_str(k, u)
        if learnable:
            cand = [nid[int(of_topic(user_topic[u], 1)[0])]] + [nid[j] for j in rng.integers(0, n_news, size=neg_k)]
        else:
            cand = list(rng.choice(nid, size=1 + neg_k))
        rows.append({'user': u, 'clicked_news': h, 'candidate_news': ' '.join(cand), 'clicked': '1 ' + ' '.join(['0'] * neg_k)})
    pd.DataFrame(rows).to_csv(os.path.join(root, 'data', 'train', 'behaviors_parsed.tsv'), sep='\t', index=False)
    for split in ('val', 'test'):
        with open(os.path.join(root, 'data', split, 'behaviors.tsv'), 'w') as f:
            for i in range(n_val_impr):
                k = int(history_lengths(rng, 1, num_clicked)[0])
                c = int(rng.integers(2, 12))
                if not learnable:
                    lab = rng.integers(0, 2, c)
                    lab[0], lab[1] = 1, 0
                    imp = ' '.join(f'{n}-{l}' for n, l in zip(rng.choice(nid, size=c, replace=False), lab))     # an impression lists a news once
                    f.write(f"{i + 1}\t{users[int(rng.integers(0, n_users))]}\t11/11/2019 9:00:00 AM\t{hist_str(k)}\t{imp}\n")
                    continue
                u = int(rng.integers(0, n_users))
                picks = rng.choice(n_news, size=c, replace=False)
                lab = (rng.random(c) < np.where(topic[picks] == user_topic[u + 1], 0.85, 0.08)).astype(np.int64)
                if lab.sum() == 0:
                    lab[0] = 1
                if lab.sum() == c:
                    lab[1] = 0
                imp = ' '.join(f'{nid[n]}-{l}' for n, l in zip(picks, lab))
                f.write(f"{i + 1}\t{users[u]}\t11/11/2019 9:00:00 AM\t{hist_str(k, u + 1)}\t{imp}\n")
    return root


_WORDS = ("the of and to in a is that for it as was with be by on not he this are or his from at which but have an had they you were "
          "their one all we can her has there been if more when will would who so no out up said what its about than into them only "
          "new some could time these two may first then do any like my now over such our man me even most made after also did many off "
          "before must well back through years much where your way down should because long each just state people those too how "
          "election storm market player game police school city court team health fire president trade season coach film music").split()


def write_raw_mind(root, n_news=40, n_users=12, n_behaviors=50, seed=0, splits=('train', 'val', 'test')):
    """A tiny tree of RAW MIND-format files (news.tsv: id, category, subcategory, title, abstract, url, title_entities,
    abstract_entities; behaviors.tsv: impression id, user, time, clicked news, impressions) for the preprocessing tools
    (data_tools.py; schemas as read by src/data_preprocess.py:33-37,99-107).  Titles carry punctuation, quotes, contractions and
    digits so that tokenisation is exercised; entities carry confidences around the 0.5 threshold."""
    import json
    import os
    rng = np.random.default_rng(seed)
    cats = ['news', 'sports', 'finance', 'health']
    subs = ['newsus', 'football_nfl', 'markets', 'wellness', 'newspolitics', 'baseball_mlb']
    ents = [f'Q{100 + i}' for i in range(12)]

    def sentence(nw):
        ws = [str(rng.choice(_WORDS)) for _ in range(nw)]
        if rng.random() < 0.3:
            ws[int(rng.integers(0, nw))] += ','
        if rng.random() < 0.2:
            ws[int(rng.integers(0, nw))] = f'"{ws[0]}"'
        if rng.random() < 0.2:
            ws.append("don't")
        if rng.random() < 0.2:
            ws.append(str(int(rng.integers(1, 2020))))
        s = ' '.join(ws).capitalize()
        return s + str(rng.choice(['.', '?', '!', '', ' ...']))

    def entities(text):
        out = []
        for w in set(text.lower().replace(',', ' ').replace('.', ' ').split()):
            if rng.random() < 0.15:
                out.append({"Label": w, "Type": "O", "WikidataId": str(rng.choice(ents)), "Confidence": float(rng.choice([0.3, 0.6, 0.9, 1.0])),
                            "OccurrenceOffsets": [0] * int(rng.integers(0, 3)), "SurfaceForms": [w]})
        return json.dumps(out)
    nid = [f'N{1000 + i}' for i in range(n_news)]
    rows = []
    for i in range(n_news):
        title = sentence(int(rng.integers(3, 26)))
        abstract = ' '.join(sentence(int(rng.integers(4, 30))) for _ in range(int(rng.integers(0, 4))))
        rows.append((nid[i], str(rng.choice(cats)), str(rng.choice(subs)), title, abstract, f'https://example.invalid/{i}', entities(title),
                     entities(abstract) if abstract and rng.random() < 0.8 else ''))
    users = [f'U{i + 1}' for i in range(n_users)]
    for split in splits:
        d = os.path.join(root, 'data', split)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'news.tsv'), 'w') as f:
            for r in rows:
                f.write('\t'.join(r) + '\n')
        with open(os.path.join(d, 'behaviors.tsv'), 'w') as f:
            for i in range(n_behaviors):
                k = int(rng.integers(0, 8))
                hist = ' '.join(rng.choice(nid, size=k)) if k else ''
                c = int(rng.integers(2, 9))
                lab = (rng.random(c) < 0.3).astype(int)
                imp = ' '.join(f'{n}-{l}' for n, l in zip(rng.choice(nid, size=c, replace=False), lab))
                f.write(f"{i + 1}\t{users[int(rng.integers(0, n_users))]}\t11/11/2019 9:00:00 AM\t{hist}\t{imp}\n")
    return root
