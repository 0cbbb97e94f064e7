#!/usr/bin/env python
"""Where does the engine's small AUC deficit in the statistical training-parity leg come from?  (TEST / ANALYSIS INFRASTRUCTURE, CPU only.)

tests/test_training_parity_gpu.py measures the engine's trained NRMS models ~3.6e-3 AUC below the reference's (8 seeds a side, z = -2.3).
Hypothesis: it is the bf16 rounding of operands (tokens, weights, Q / K / V, attention probabilities, ctx rows, and in the backward dctx, dqkv,
dpre, dX), not a defect.  This script trains the ORACLE (fp32 torch restatement, pinned to the reference) on the fixture's task with exactly
those tensors rounded to bf16 -- straight-through in the forward, rounded again on the way back -- with torch's own dropout and
torch.optim.Adam, and prints the held-out AUCs next to the fixture's fp32 reference runs.

    python tools/bf16_bias_experiment.py [--seeds 8] [--procs 3] [--mode fwd|bwd|both]
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    seed, mode, threads = args
    import torch
    import torch.nn.functional as F
    from oracle import train_parity as tp
    from oracle.nrms_torch import OracleNRMS
    torch.set_num_threads(threads)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_parity', 'nrms.npz'))
    task = tp.task_from_arrays(z)
    st0 = tp.init_state(task["num_words"])

    class Q(torch.autograd.Function):            # bf16 rounding of a tensor; the gradient passing back through it is rounded too (mode-dependent)
        @staticmethod
        def forward(ctx, x):
            return x.bfloat16().float() if mode in ('fwd', 'both') else x

        @staticmethod
        def backward(ctx, g):
            return g.bfloat16().float() if mode in ('bwd', 'both') else g
    q = Q.apply

    def mhsa(m, x):
        B, S, _ = x.shape
        lin = lambda L: q(F.linear(x, q(L.weight), L.bias)).view(B, S, m.h, m.dk).transpose(1, 2)
        qq, kk, vv = lin(m.W_Q), lin(m.W_K), lin(m.W_V)
        e = torch.exp(qq @ kk.transpose(-1, -2) / math.sqrt(m.dk))
        a = q(e / (e.sum(dim=-1, keepdim=True) + 1e-8))
        return (a @ vv).transpose(1, 2).contiguous().view(B, S, m.h * m.dk)

    def additive(m, x):
        t = torch.tanh(q(F.linear(x, q(m.linear.weight), m.linear.bias)))
        w = F.softmax(t @ m.attention_query_vector, dim=1)
        return torch.bmm(w.unsqueeze(1), x).squeeze(1)

    def news(m, title):
        x = q(F.dropout(m.word_embedding(title), p=m.p, training=m.training))
        y = q(F.dropout(mhsa(m.multihead_self_attention, x), p=m.p, training=m.training))
        return additive(m.additive_attention, y)

    def user(m, x):
        return additive(m.additive_attention, q(mhsa(m.multihead_self_attention, q(x))))

    def fwd(model, cand, click):
        c = torch.stack([news(model.news_encoder, x['title']) for x in cand], dim=1)
        h = torch.stack([news(model.news_encoder, x['title']) for x in click], dim=1)
        return torch.bmm(c, user(model.user_encoder, h).unsqueeze(-1)).squeeze(-1)

    m = OracleNRMS(task["num_words"], 300, 15, 200, 0.2)
    m.load_state_dict(st0)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    torch.manual_seed(seed)
    y = torch.zeros(task["B"], dtype=torch.long)
    for i in range(task["steps"]):
        loss = crit(fwd(m, tp.as_lists(task["cand_ids"][i]), tp.as_lists(task["click_ids"][i])), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    met = tp.eval_metrics(task, tp.oracle_eval_scores(task, {k: v.detach() for k, v in m.state_dict().items()}))
    return seed, [float(x) for x in met]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--procs', type=int, default=3)
    ap.add_argument('--threads', type=int, default=2)
    ap.add_argument('--mode', default='both')
    a = ap.parse_args()
    with mp.get_context('spawn').Pool(a.procs) as pool:
        res = dict(pool.imap_unordered(run, [(s, a.mode, a.threads) for s in range(a.seeds)]))
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_parity', 'nrms.npz'))
    auc = np.array([res[s][0] for s in sorted(res)])
    ref = z['ref_metrics'][:, 0]
    se = math.sqrt(auc.var(ddof=1) / len(auc) + ref.var(ddof=1) / len(ref))
    print(json.dumps({"mode": a.mode, "quantised_oracle_auc": [round(float(x), 4) for x in auc], "mean": float(auc.mean()), "reference_mean": float(ref.mean()),
                      "diff": float(auc.mean() - ref.mean()), "stderr": se, "z": float((auc.mean() - ref.mean()) / se)}))


if __name__ == '__main__':
    main()
