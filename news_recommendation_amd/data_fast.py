"""GPU-resident training data: the reference's ``BaseDataset`` (src/dataset.py:17-85) builds 53 dicts of
small tensors per sample in Python and the default collate restacks them per batch in 4 worker processes -- at ~85 k
impressions/s the engine would starve behind it.  Here the two training files are parsed ONCE:

  news_parsed.tsv       -> one int64 device tensor per attribute, ``[n_news + 1, ...]``; the extra last row is the all-zero
                           padding news of dataset.py:44-60
  behaviors_parsed.tsv  -> index arrays: candidates ``[M, 1+K]`` (positive first, data_preprocess.py:63-66), LEFT-padded history
                           ``[M, N]`` (dataset.py:79-83), user id ``[M]``, clicked_news_length ``[M]``

A batch is then one index gather per attribute on the device, already in the order the news encoder consumes (candidates, then history), and
feeds ``model.forward_stacked`` (what ``forward_ids`` and the reference-format ``forward(list-of-dicts)`` reduce to).  Same sample semantics as BaseDataset.__getitem__.
"""
from ast import literal_eval

import numpy as np
import pandas as pd
import torch

TEXT_ATTRS = ('title', 'abstract', 'title_entities', 'abstract_entities')


class TrainData:
    def __init__(self, behaviors_path, news_path, config, device, rank=0, world=1):
        attrs = list(config.dataset_attributes['news'])
        N = config.num_clicked_news_a_user
        news = pd.read_table(news_path, index_col='id', usecols=['id'] + attrs,
                             converters={a: literal_eval for a in set(attrs) & set(TEXT_ATTRS)})
        nid2row = {n: i for i, n in enumerate(news.index)}
        self.n_news = len(news)
        self.news = {}
        for a in attrs:
            t = np.asarray(news[a].tolist(), dtype=np.int64)
            pad = np.zeros((1,) + t.shape[1:], dtype=np.int64)
            self.news[a] = torch.from_numpy(np.concatenate([t, pad])).to(device)
        beh = pd.read_table(behaviors_path)
        self.n_total = len(beh)                                      # samples over ALL ranks (iteration counts must not depend on the shard)
        beh = beh.iloc[rank::world]                                  # data parallel: every rank owns a strided shard of the samples
        cand, hist, length = [], [], []
        for c, h in zip(beh['candidate_news'].tolist(), beh['clicked_news'].tolist()):
            cand.append([nid2row[x] for x in c.split()])
            clicked = [nid2row[x] for x in str(h).split()[:N]] if isinstance(h, str) else []
            length.append(len(clicked))
            hist.append([self.n_news] * (N - len(clicked)) + clicked)
        self.cand = torch.tensor(cand, dtype=torch.int64, device=device)
        self.hist = torch.tensor(hist, dtype=torch.int64, device=device)
        self.length = torch.tensor(length, dtype=torch.int64)          # stays on the host (LSTUR packs with CPU lengths)
        self.user = torch.tensor(beh['user'].tolist(), dtype=torch.int64, device=device) if 'user' in beh else None
        self.device = device

    def __len__(self):
        return self.cand.shape[0]

    def batches(self, batch_size, generator=None):
        """One shuffled pass, drop_last (DataLoader(shuffle=True, drop_last=True), train.py:118-124)."""
        perm = torch.randperm(len(self), generator=generator)
        for i in range(0, len(self) - batch_size + 1, batch_size):
            yield self.batch(perm[i:i + batch_size])

    def batch(self, idx):
        """The engine's batch layout (model.forward_stacked): per attribute ONE gather of the B*C candidate rows (impression-major) followed by
        the B*N history rows -- the order the news encoder consumes them in, so that no concatenation follows."""
        di = idx.to(self.device)
        c, h = self.cand[di], self.hist[di]
        rows = torch.cat([c.reshape(-1), h.reshape(-1)])
        b = {'ids': gather_rows(self.news, rows), 'B': c.shape[0], 'C': c.shape[1]}
        if self.user is not None:
            b['user'] = self.user[di]
        b['length'] = self.length[idx]
        return b


def gather_rows(tables, rows):
    """{attr: table[rows]} with the TEXT attributes gathered into ONE allocation, back to back in the order of `tables` (title tokens, then
    abstract tokens): the embedding backward sorts all token ids of a step as one stream (ops_conv.sort_tokens_async), and finds it there without
    a concatenation."""
    text = [a for a in tables if a in TEXT_ATTRS and tables[a].dim() == 2]
    out = {}
    if len(text) > 1:
        n = rows.shape[0]
        buf = torch.empty(sum(n * tables[a].shape[1] for a in text), dtype=tables[text[0]].dtype, device=rows.device)
        lo = 0
        for a in text:
            w = tables[a].shape[1]
            out[a] = torch.index_select(tables[a], 0, rows, out=buf[lo:lo + n * w].view(n, w))
            lo += n * w
    for a, t in tables.items():
        if a not in out:
            out[a] = t[rows]
    return {a: out[a] for a in tables}


def pack_text_streams(ids):
    """The same layout for id tensors that already exist: the text attributes of `ids` copied into one allocation, back to back."""
    text = [a for a in ids if a in TEXT_ATTRS and ids[a].dim() == 2]
    if len(text) < 2:
        return ids
    buf = torch.cat([ids[a].reshape(-1) for a in text])
    out, lo = dict(ids), 0
    for a in text:
        out[a] = buf[lo:lo + ids[a].numel()].view(ids[a].shape)
        lo += ids[a].numel()
    return out


def split_batch(b):
    """(cand, click) dicts of [B, C, ...] / [B, N, ...] views of a stacked batch (the layout forward_ids takes)."""
    B, C = b['B'], b['C']
    cand = {a: t[:B * C].view(B, C, *t.shape[1:]) for a, t in b['ids'].items()}
    click = {a: t[B * C:].view(B, -1, *t.shape[1:]) for a, t in b['ids'].items()}
    return cand, click


def forward_batch(model, b, loss=False, target=None):
    """Logits [B, 1+K] -- or, loss=True, the scalar cross entropy against `target` (None = class 0, train.py:205) -- of a TrainData batch through
    the model's stacked-id entry point."""
    name = type(model).__name__
    if name == 'NRMS':
        return model.forward_stacked(b['ids']['title'], b['B'], b['C'], loss=loss, target=target)
    if name == 'LSTUR':
        return model.forward_stacked(b['user'], b['length'].clone(), b['ids'], b['B'], b['C'], loss=loss, target=target)
    return model.forward_stacked(b['ids'], b['B'], b['C'], loss=loss, target=target)
