#!/usr/bin/env python
"""A/B of the embedding scatter's span (positions per wave; NR_SCATTER_SPAN is read once per process): the NAML and NRMS token streams of a
B = 512 step (Zipf ids, padding id 0), HIP events over 20 launches.  python tools/scatter_ab.py -> one JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from news_recommendation_amd import _capi, ops, synth   # noqa: E402

lib = _capi.load()
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
out = {"span": os.environ.get('NR_SCATTER_SPAN', 'default(256)')}
for name, streams in (('NRMS', [(27136, 20)]), ('NAML', [(27136, 20), (27136, 50)])):
    ids = np.concatenate([(synth.news_titles(rng, n, L, 70976) if L == 20 else synth.news_abstracts(rng, n, L, 70976)).reshape(-1) for n, L in streams])
    t = torch.from_numpy(ids).to(dev)
    ids_sorted, perm = ops.sort_ids(t, 70976)
    dx = (torch.randn(ids.size, 320, device=dev) * 0.01).to(torch.bfloat16).view(torch.int16)
    g = torch.zeros(70976, 300, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _capi.check(lib, lib.nr_embed_scatter_sorted(ids_sorted.data_ptr(), perm.data_ptr(), dx.data_ptr(), 320, g.data_ptr(), 70976, ids.size, 0.2, 1234, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _capi.check(lib, lib.nr_embed_scatter_sorted(ids_sorted.data_ptr(), perm.data_ptr(), dx.data_ptr(), 320, g.data_ptr(), 70976, ids.size, 0.2, 1234, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    live = int((ids != 0).sum())
    out[name] = {"us": round(us, 1), "tokens": int(ids.size), "non_padding": live, "row_read_GBs": round(live * 640 / us / 1e3, 1)}
print(json.dumps(out))
