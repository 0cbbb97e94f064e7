#!/usr/bin/env python
"""Benchmark of the news-recommendation hot path on MI355X: training impressions/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W [--model NRMS|NAML|LSTUR] [--shape small|large|xlarge]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of MIND-shaped synthetic impressions resident in HBM: forward (embedding gather ->
news encoder for 53 news/impression -> user encoder -> dot-product scorer), cross-entropy, backward, gradient exchange over RCCL
(N > 1) and the Adam update -- everything src/train.py:182-233 does per batch.

Workloads (BASELINE.json configs): N = 1 defaults to configs[1] -- NRMS, bf16 operands / fp32 accumulate, batch 512, MIND-small shape
(title_len 20, 50 clicked news, d 300, vocabulary 70,976).  N > 1 defaults to the MIND-LARGE shape the multi-GPU configs name
(configs[3] NRMS DP, configs[4] with --model LSTUR): 161,013 news, vocabulary 1 + 130,000 (knob: --vocab), 711,223 users, batch 512
per GPU (weak scaling).  --model NAML is configs[2].  --shape overrides the default.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline        dominant hand-written kernel: algorithmic FLOPs per launch / HIP-event duration vs the dense bf16 MFMA peak
                  (MI355X_MICROARCH.md: 2.5 PFLOP/s); traffic = HBM bytes per launch from the committed PMC passes (profiles/traffic.json,
                  keyed on a hash of the kernel source so that it cannot go stale silently)
  gather_roofline the embedding gather (north_star) at two points: the workload's table (Infinity-Cache resident) and a > 256 MB table
                  with uniform ids (HBM bound): algorithmic bytes / duration vs 8 TB/s
  value_dropin    the same step through the REAL drop-in boundary: model(candidate_news, clicked_news) on the DataLoader's CPU
                  list-of-dicts (train.py:166-203), pinned H2D copy included
  score_eval      eval-shaped scoring throughput, phases A-C of src/evaluate.py:185-272 on the batched driver (evaluate_fast.run_plan)
  cpu_baseline    the reference's own modules (kind "reference", when /root/reference/src exists) or the oracle's torch port (kind
                  "port") timed on this host's cores in a GPU-hidden child process: train step, eval forward, evaluate()
  parity          AUC / nDCG@10 of engine vs oracle on n = 1000 AND n = 5000 eval impressions, for three weight states (N = 1 only)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s
CRITERION_STEP = os.environ.get('NR_BENCH_CRITERION', '0') == '1'      # A/B: the round-5 step (forward_ids + torch CrossEntropyLoss) instead of train_fast's
BYTES_PER_TOKEN_F32 = 1200   # one fp32 embedding row
REFERENCE_SRC = '/root/reference/src'


def make_cfg(model, shape, vocab=0):
    """The reference's BaseConfig / <Model>Config knobs (src/config.py:10-69) at the requested dataset shape."""
    from news_recommendation_amd import default_config, synth
    sh = dict(synth.SHAPES[shape])
    if vocab:
        sh['num_words'] = vocab
    base = getattr(default_config, f'{model}Config')
    cfg = type(f'{model}Config', (base,), {k: sh[k] for k in ('num_words', 'num_users', 'num_categories')})
    cfg.num_news = sh['num_news']
    cfg.eval_impressions = sh['eval_impressions']
    cfg.shape = shape
    return cfg


# ----------------------------------------------------------------------------------------------------------------------
# synthetic news table + batches (ids resident in HBM)
# ----------------------------------------------------------------------------------------------------------------------
def news_table(cfg, seed, n_news):
    from news_recommendation_amd import synth
    rng = np.random.default_rng(seed)
    return {'title': synth.news_titles(rng, n_news, cfg.num_words_title, cfg.num_words),
            'abstract': synth.news_abstracts(rng, n_news, cfg.num_words_abstract, cfg.num_words),
            'category': rng.integers(1, cfg.num_categories, size=n_news).astype(np.int64),
            'subcategory': rng.integers(1, cfg.num_categories, size=n_news).astype(np.int64)}


def take(news, attr, idx):
    """news-index batch -> attribute tensor; padded history slots (-1) are all-zero news (dataset.py:44-60,79-83)."""
    a = news[attr]
    pad = np.zeros((1,) + a.shape[1:], dtype=np.int64)
    return np.concatenate([a, pad])[np.where(idx < 0, a.shape[0], idx)]


class Workload:
    def __init__(self, name, cfg):
        self.name, self.cfg = name, cfg
        self.attrs = {'NRMS': ('title',), 'NAML': ('title', 'abstract', 'category', 'subcategory'),
                      'LSTUR': ('title', 'category', 'subcategory')}[name]
        self._news = None

    def news(self):
        if self._news is None:
            self._news = news_table(self.cfg, 0, self.cfg.num_news)
        return self._news

    def make_model(self, seed=0, pretrained=None):
        torch.manual_seed(seed)
        import importlib
        cls = getattr(importlib.import_module(f'news_recommendation_amd.dropin.model.{self.name}'), self.name)
        return cls(self.cfg, pretrained)

    def make_oracle(self):
        c = self.cfg
        if self.name == 'NRMS':
            from oracle.nrms_torch import OracleNRMS
            return OracleNRMS(c.num_words, 300, 15, 200, c.dropout_probability)
        if self.name == 'NAML':
            from oracle.naml_torch import OracleNAML
            return OracleNAML(c.num_words, 300, c.num_categories, 100, 300, 3, 200, c.dropout_probability)
        from oracle.lstur_torch import OracleLSTUR
        return OracleLSTUR(c.num_words, 300, c.num_categories, c.num_users, 300, 3, 200, c.dropout_probability, 0.5, 'ini')

    def make_optimizer(self, model):
        from news_recommendation_amd.optim import EngineAdam
        return EngineAdam(model, lr=self.cfg.learning_rate, row_sparse=('user_embedding.weight',) if self.name == 'LSTUR' else ())

    def batches(self, rank, n_batches, B, device):
        """Stacked id tensors on `device` ('cpu' gives the raw material of the list-of-dicts form)."""
        from news_recommendation_amd import synth
        c, news = self.cfg, self.news()
        rng = np.random.default_rng(1000 + rank)
        out = []
        for _ in range(n_batches):
            cand, hist = synth.train_batch(rng, news['title'], B, c.num_clicked_news_a_user, c.negative_sampling_ratio)
            b = {'cand': {k: torch.from_numpy(take(news, k, cand)).to(device) for k in self.attrs},
                 'click': {k: torch.from_numpy(take(news, k, hist)).to(device) for k in self.attrs}}
            if self.name == 'LSTUR':
                b['user'] = torch.from_numpy(rng.integers(1, c.num_users, size=B).astype(np.int64)).to(device)
                b['length'] = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))          # CPU, as the reference requires
            # the engine's own batch layout (data_fast.TrainData.batch, model.forward_stacked): candidates' rows, then the history rows
            C, N = cand.shape[1], hist.shape[1]
            b['ids'] = {k: torch.cat([b['cand'][k].reshape(B * C, *b['cand'][k].shape[2:]), b['click'][k].reshape(B * N, *b['click'][k].shape[2:])])
                        for k in self.attrs}
            from news_recommendation_amd.data_fast import pack_text_streams
            b['ids'] = pack_text_streams(b['ids'])           # title and abstract tokens back to back, as TrainData.batch lays them out
            b['B'], b['C'] = B, C
            out.append(b)
        return out

    def loss(self, model, b, crit=None, target=None):
        """The training loop's `criterion(y_pred, y)` (train.py:205-206) for one batch.  Default: what news_recommendation_amd.train_fast runs -- the
        stacked-id entry point with the fused scorer + cross-entropy kernels (y = class 0); NR_BENCH_CRITERION=1 (A/B): forward_ids + a
        torch.nn.CrossEntropyLoss on the logits, the round-5 form of this step."""
        if CRITERION_STEP:
            return crit(self.forward(model, b), target)
        if self.name == 'NRMS':
            return model.forward_stacked(b['ids']['title'], b['B'], b['C'], loss=True)
        if self.name == 'NAML':
            return model.forward_stacked(b['ids'], b['B'], b['C'], loss=True)
        return model.forward_stacked(b['user'], b['length'].clone(), b['ids'], b['B'], b['C'], loss=True)

    def forward(self, model, b):
        if self.name == 'NRMS':
            return model.forward_ids(b['cand']['title'], b['click']['title'])
        if self.name == 'NAML':
            return model.forward_ids(b['cand'], b['click'])
        return model.forward_ids(b['user'], b['length'].clone(), b['cand'], b['click'])

    def as_dataloader_batch(self, b):
        """What train.py:166 gets from the DataLoader: candidate_news = list[1+K] of {attr: CPU tensor [B, ...]}, clicked_news =
        list[N] of the same (default collate; pin_memory=True, train.py:118-124)."""
        C, N = b['cand']['title'].shape[1], b['click']['title'].shape[1]
        pin = (lambda t: t.pin_memory()) if torch.cuda.is_available() else (lambda t: t)
        mb = {'candidate_news': [{k: pin(b['cand'][k][:, j].contiguous()) for k in self.attrs} for j in range(C)],
              'clicked_news': [{k: pin(b['click'][k][:, j].contiguous()) for k in self.attrs} for j in range(N)]}
        if self.name == 'LSTUR':
            mb['user'], mb['clicked_news_length'] = b['user'], b['length']
        return mb

    def forward_dropin(self, model, mb):
        if self.name == 'LSTUR':            # train.py:183-185
            return model(mb['user'], mb['clicked_news_length'].clone(), mb['candidate_news'], mb['clicked_news'])
        return model(mb['candidate_news'], mb['clicked_news'])      # train.py:202-203

    def hbm_bytes(self, B):
        """Algorithmic HBM bytes per launch of the kernels that are bandwidth / instruction bound, not dense contractions (the
        attention core is reported against HBM, never against the MFMA peak): bf16 saves read + bf16 results written."""
        tok = B * 53 * 20
        tok_a = B * 53 * 50
        return {
            'nr_qkv_proj_fwd[S=20]': tok * (1200 + 8 + 3 * 300 * 2 + 300 * 2),       # table row + id in; Q, K, V and the masked token row (dW operand) out
            'nr_dx_gemm[S=20]': tok * (3 * 300 * 2 + 300 * 2),                       # dQ | dK | dV in, dX out
            'nr_gemm_tn_dWqkv[S=20]': tok * (3 * 300 * 2 + 300 * 2),                 # dqkv and the token rows in (the 1.2 MB result is noise)
            'nr_gemm_tn_dWa[S=20]': tok * (200 * 2 + 300 * 2),
            'nr_additive_bwd[S=20]': tok * (300 * 2 + 200 * 2 + 300 * 2),            # ctx in; dpre, dctx out (1,696 B with padding, 1,600 without)
            'nr_additive_bwd[title]': tok * (300 * 2 + 200 * 2 + 300 * 2),
            'nr_additive_bwd[abstract]': tok_a * (300 * 2 + 200 * 2 + 300 * 2),
            'nr_additive_fwd[title]': tok * 300 * 2 + B * 53 * (300 * 2 + 20 * 4),    # act rows in; pooled vector + weights out
            'nr_additive_fwd[abstract]': tok_a * 300 * 2 + B * 53 * (300 * 2 + 50 * 4),
            'nr_conv3_fwd[title]': tok * (1200 + 8 + 300 * 2 + 300 * 2),             # table row + id in; activation + masked token row out
            'nr_conv3_fwd[abstract]': tok_a * (1200 + 8 + 300 * 2 + 300 * 2),
            'nr_conv3_dgrad[title]': tok * (300 * 2 + 300 * 2), 'nr_conv3_dgrad[abstract]': tok_a * (300 * 2 + 300 * 2),
            'nr_gemm_tn_dWconv[title]': tok * (300 * 2 + 300 * 2), 'nr_gemm_tn_dWconv[abstract]': tok_a * (300 * 2 + 300 * 2),
            'nr_attn_fwd[S=20]': tok * (3 * 300 * 2 + 300 * 2),                      # Q, K, V in; ctx out
            'nr_attn_pool_fwd[S=20]': tok * (3 * 300 * 2 + 300 * 2) + B * 53 * (300 + 20) * 4,   # + pooled vectors and attention weights out
            'nr_attn_bwd[S=20]': tok * (3 * 300 * 2 + 300 * 2 + 3 * 300 * 2),        # Q, K, V, dctx in; dQ, dK, dV out
            'nr_embed_scatter_sorted[S=20]': tok * (300 * 2 + 16),                   # dX rows + (id, position) in; table rows reduced in registers
        }

    def flops(self, B):
        """Algorithmic FLOPs per launch of the kernels we know how to price (SURVEY 8 d6 / DESIGN.md)."""
        T = B * 53
        conv = lambda S: T * 2 * S * 900 * 300
        mhsa20 = T * (2 * 20 * 300 * 900 + 2 * 2 * 15 * 20 * 20 * 20)
        pool20 = T * 2 * 20 * 300 * 200
        return {
            'nr_mhsa_fwd[S=20]': mhsa20, 'nr_news_fwd[S=20]': mhsa20 + pool20,
            'nr_qkv_proj_fwd[S=20]': T * 2 * 20 * 300 * 900, 'nr_attn_fwd[S=20]': T * 2 * 2 * 15 * 20 * 20 * 20,
            'nr_mhsa_fwd[S=50]': B * (2 * 50 * 300 * 900 + 2 * 2 * 15 * 50 * 50 * 20),
            'nr_attn_bwd[S=20]': T * 15 * 6 * 2 * 20 * 20 * 20,
            'nr_attn_bwd[S=50]': B * 15 * 6 * 2 * 50 * 50 * 20,
            'nr_additive_fwd[S=20]': pool20, 'nr_additive_bwd[S=20]': 3 * pool20 // 2 + pool20,      # projection recomputed, dctx = dpre Wa, (dWa is its own GEMM)
            'nr_attn_pool_fwd[S=20]': T * 2 * 2 * 15 * 20 * 20 * 20 + pool20,
            'nr_dx_gemm[S=20]': T * 2 * 20 * 900 * 300, 'nr_gemm_tn_dWqkv[S=20]': T * 2 * 20 * 900 * 300, 'nr_gemm_tn_dWa[S=20]': pool20,
            'nr_gemm_tn_dWconv[title]': conv(20), 'nr_gemm_tn_dWconv[abstract]': conv(50),
            'nr_additive_fwd[title]': pool20, 'nr_additive_bwd[title]': pool20,
            'nr_additive_fwd[abstract]': T * 2 * 50 * 300 * 200, 'nr_additive_bwd[abstract]': T * 2 * 50 * 300 * 200,
            'nr_conv3_fwd[title]': conv(20), 'nr_conv3_dgrad[title]': conv(20),
            'nr_conv3_fwd[abstract]': conv(50), 'nr_conv3_dgrad[abstract]': conv(50),
            # one entry per SWEEP (ops_gru issues the recurrence from one C call: a persistent launch on MI355X): N forward steps, N + 1 backward calls
            'nr_gru_fwd_seq': self.cfg.num_clicked_news_a_user * 2 * B * 900 * 2700,
            'nr_gru_bwd_seq': (self.cfg.num_clicked_news_a_user + 1) * 2 * B * 2700 * 900,
        }


def price_kernel(name, rec, flops, hbm, traffic=None, busy=None):
    """One kernel against BOTH roofs (the record must be comparable from round to round whichever kernel happens to be the
    arg-max): rec = (launches, average us, total us) from HIP events on the launch stream; algorithmic flops / bytes per launch from
    Workload.flops / hbm_bytes (None where no figure is tabulated); PMC traffic / MFMA-busy from the committed rocprofv3 passes when they were
    taken on these kernel sources."""
    n, avg, tot = rec
    o = {"kernel": name, "avg_us": avg, "launches": n}
    if name in flops:
        o["flop_per_launch"] = flops[name]
        o["frac_mfma"] = flops[name] / (avg * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF
    if name in hbm:
        o["bytes_per_launch"] = hbm[name]
        o["frac_hbm"] = hbm[name] / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS
    o["traffic"] = (traffic or {}).get(name)
    if o["traffic"] is not None and avg > 0:
        o["frac_hbm_traffic"] = o["traffic"] / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS
    if busy and name in busy:
        o["mfma_busy_frac"] = busy[name]
    return o


def committed_counters(model, shape, B):
    """(per-kernel PMC HBM traffic, per-kernel MFMA-busy fraction) from profiles/traffic.json / profiles/mfma_busy.json -- only entries measured
    on the kernel sources this library was built from (source hash) and, for traffic, on this workload."""
    traffic, busy = {}, {}
    h = kernel_source_hash()
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            for k, v in json.load(f).items():
                if isinstance(v, dict) and v.get("source_hash") == h and v.get("workload") == f"{model}/{shape}/B{B}":
                    traffic[k] = v["bytes"]
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(ROOT, 'profiles', 'mfma_busy.json')) as f:
            for k, v in json.load(f).items():
                if isinstance(v, dict) and v.get("source_hash") == h:
                    busy[k] = v["mfma_busy_frac"]
    except (OSError, ValueError):
        pass
    return traffic, busy


# ----------------------------------------------------------------------------------------------------------------------
# measurement legs besides the headline
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, shape, vocab, budget=24.0):
    """The reference's CPU PyTorch path timed on this host in a child process that cannot see the GPUs."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_baseline.py'), '--model', wl.name, '--shape', shape, '--budget', str(budget)]
    if vocab:
        cmd += ['--vocab', str(vocab)]
    if os.path.isdir(REFERENCE_SRC):
        cmd += ['--reference', REFERENCE_SRC]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        return json.loads(line)
    except Exception as e:        # noqa: BLE001 -- a failed baseline leg must not lose the measured GPU line
        return {"value": None, "unit": "impressions/s", "cores": None, "kind": "port", "sample": f"cpu baseline child failed: {e!r}"}


def eval_set(wl, n_news, n_impr, seed=3, news_seed=2):
    from news_recommendation_amd import synth
    rng = np.random.default_rng(seed)
    news = {k: v for k, v in news_table(wl.cfg, news_seed, n_news).items() if k in wl.attrs}
    hist, cands, ptr = synth.eval_impressions(rng, n_news, n_impr)
    users = rng.integers(1, wl.cfg.num_users, size=n_impr).astype(np.int64)
    return news, hist, cands, ptr, users


def engine_scores(wl, model, device, es):
    """Phases A-C of src/evaluate.py:185-260 on the engine: news vectors, user vectors (PADDED_NEWS = zero vector), CSR dot products."""
    from news_recommendation_amd import ops
    news, hist, cands, ptr, users = es
    n_news, n_impr = news['title'].shape[0], hist.shape[0]
    tn = {k: torch.from_numpy(v) for k, v in news.items()}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        nv = torch.cat([model.get_news_vector({k: v[i:i + 2048] for k, v in tn.items()}) for i in range(0, n_news, 2048)])
        nvp = torch.cat([nv, torch.zeros(1, nv.shape[1], device=device)])
        hidx = torch.from_numpy(np.where(hist < 0, n_news, hist)).to(device)
        if wl.name == 'LSTUR':
            lengths = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))
            uv = torch.cat([model.get_user_vector(torch.from_numpy(users[i:i + 1024]), lengths[i:i + 1024].clone(), nvp[hidx[i:i + 1024]])
                            for i in range(0, n_impr, 1024)])
        else:
            uv = torch.cat([model.get_user_vector(nvp[hidx[i:i + 1024]]) for i in range(0, n_impr, 1024)])
        sc = ops.score_csr(nv, uv, torch.from_numpy(cands).to(device), torch.from_numpy(ptr).to(device),
                           torch.arange(n_impr, dtype=torch.int32, device=device)).cpu().numpy()
    model.train(was_training)
    return sc


def oracle_scores(wl, state_dict, es):
    news, hist, cands, ptr, users = es
    n_news, n_impr = news['title'].shape[0], hist.shape[0]
    ref = wl.make_oracle()
    ref.load_state_dict(state_dict)
    ref.eval()
    tn = {k: torch.from_numpy(v) for k, v in news.items()}
    with torch.no_grad():
        nv = torch.cat([ref.get_news_vector({k: v[i:i + 1024] for k, v in tn.items()}) for i in range(0, n_news, 1024)])
        nvp = torch.cat([nv, torch.zeros(1, nv.shape[1])])                     # PADDED_NEWS = zero vector (evaluate.py:203)
        hidx = torch.from_numpy(np.where(hist < 0, n_news, hist))
        if wl.name == 'LSTUR':
            lengths = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))
            uv = torch.cat([ref.get_user_vector(torch.from_numpy(users[i:i + 256]), lengths[i:i + 256].clone(), nvp[hidx[i:i + 256]])
                            for i in range(0, n_impr, 256)])
        else:
            uv = torch.cat([ref.get_user_vector(nvp[hidx[i:i + 256]]) for i in range(0, n_impr, 256)])
    return np.concatenate([(nv[cands[ptr[i]:ptr[i + 1]]] @ uv[i]).numpy() for i in range(n_impr)])


def parity_eval(wl, states, device, n_news=4000, n_impr=5000):
    """AUC / nDCG@10 of the engine vs the CPU oracle on the same synthetic eval-shaped impressions and the same weights, reported for
    the first 1,000 impressions (BASELINE configs[0]) and for all 5,000, for every weight state in `states`
    ({name: engine model}).  Labels come from a seeded teacher on the ORACLE's scores; the budget is 1e-3."""
    from news_recommendation_amd import synth
    from oracle import metrics
    es = eval_set(wl, n_news, n_impr)
    ptr = es[3]
    out = {"tolerance": 1e-3, "n_news": n_news, "states": {}}
    worst = {1000: [0.0, 0.0], n_impr: [0.0, 0.0]}
    for name, model in states.items():
        sc = engine_scores(wl, model, device, es)
        sc_ref = oracle_scores(wl, {k: v.detach().cpu() for k, v in model.state_dict().items()}, es)
        labels = synth.teacher_labels(np.random.default_rng(4), sc_ref.astype(np.float64), ptr)
        st = {"max_abs_logit_err": float(np.abs(sc - sc_ref).max()), "rms_logit_err": float(np.sqrt(np.mean((sc - sc_ref) ** 2))),
              "mean_logit_err": float(np.mean(sc - sc_ref)), "logit_scale": float(np.abs(sc_ref).max())}
        for n in (1000, n_impr):
            split = lambda a: [a[ptr[i]:ptr[i + 1]] for i in range(n)]
            auc_r, _, _, nd_r = metrics.evaluate_impressions(split(labels), split(sc_ref))
            auc_e, _, _, nd_e = metrics.evaluate_impressions(split(labels), split(sc))
            st[f"n{n}"] = {"auc_oracle": auc_r, "auc_engine": auc_e, "ndcg10_oracle": nd_r, "ndcg10_engine": nd_e,
                           "abs_diff_auc": abs(auc_r - auc_e), "abs_diff_ndcg10": abs(nd_r - nd_e)}
            worst[n][0] = max(worst[n][0], abs(auc_r - auc_e))
            worst[n][1] = max(worst[n][1], abs(nd_r - nd_e))
        out["states"][name] = st
    out["worst_abs_diff_auc_n1000"], out["worst_abs_diff_ndcg10_n1000"] = worst[1000]
    out[f"worst_abs_diff_auc_n{n_impr}"], out[f"worst_abs_diff_ndcg10_n{n_impr}"] = worst[n_impr]
    out["within_tolerance"] = bool(max(worst[1000] + worst[n_impr]) < 1e-3)
    return out


def parity_seeds(wl, states, device, seeds=8, n_news=2000, n_impr=1000):
    """The 1e-3 claim as a statistic, not a single draw: for every weight state, `seeds` independent evaluation sets of BASELINE configs[0]'s
    size (n = 1,000 eval-shaped impressions; each seed draws its own news table, histories, candidate lists and teacher labels); per state
    the max AND the mean of |dAUC| and |dnDCG@10| over the seeds; `within_tolerance` is judged on the max over all states and seeds."""
    from news_recommendation_amd import synth
    from oracle import metrics
    out = {"seeds": seeds, "n_impr": n_impr, "n_news": n_news, "tolerance": 1e-3, "states": {}}
    gmax = [0.0, 0.0]
    for name, model in states.items():
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        da, dn = [], []
        for s_ in range(seeds):
            es = eval_set(wl, n_news, n_impr, seed=100 + 7 * s_, news_seed=200 + s_)
            ptr = es[3]
            sc = engine_scores(wl, model, device, es)
            sc_ref = oracle_scores(wl, sd, es)
            labels = synth.teacher_labels(np.random.default_rng(300 + s_), sc_ref.astype(np.float64), ptr)
            split = lambda a: [a[ptr[i]:ptr[i + 1]] for i in range(n_impr)]
            auc_r, _, _, nd_r = metrics.evaluate_impressions(split(labels), split(sc_ref))
            auc_e, _, _, nd_e = metrics.evaluate_impressions(split(labels), split(sc))
            da.append(abs(auc_r - auc_e))
            dn.append(abs(nd_r - nd_e))
        out["states"][name] = {"abs_diff_auc": {"max": max(da), "mean": float(np.mean(da)), "all": da},
                               "abs_diff_ndcg10": {"max": max(dn), "mean": float(np.mean(dn)), "all": dn}}
        gmax = [max(gmax[0], max(da)), max(gmax[1], max(dn))]
    out["max_abs_diff_auc"], out["max_abs_diff_ndcg10"] = gmax
    out["mean_abs_diff_auc"] = float(np.mean([v["abs_diff_auc"]["mean"] for v in out["states"].values()]))
    out["mean_abs_diff_ndcg10"] = float(np.mean([v["abs_diff_ndcg10"]["mean"] for v in out["states"].values()]))
    out["within_tolerance"] = bool(max(gmax) < 1e-3)
    return out


def train_parity(device, steps=200, B=16, lr=1e-3, engine_seeds=(0, 1), oracle=True, oracle_seeds=(0, 1)):
    """Statistical training parity: the engine and the CPU oracle train NRMS for `steps` steps with dropout ON from
    the same initial weights on the same teacher-labelled batches (oracle/train_parity.py: the reference's loop, src/train.py:202-233, with
    torch.optim.Adam and torch's dropout; the engine with EngineAdam and its own counter-based dropout), then both trained models rank the
    same held-out eval-shaped impressions.  The two cannot agree bit for bit (different dropout streams, bf16 vs fp32): what is compared is
    AUC / nDCG@10 of the trained models, next to the engine's own seed-to-seed spread.  lr = 1e-3 (10 x src/config.py:19) so that
    200 small steps move AUC from 0.51 to ~0.75 (teacher: 0.87); vocabulary 6,000 keeps the oracle's dense table gradients cheap."""
    from news_recommendation_amd import ops
    from news_recommendation_amd.optim import EngineAdam
    from oracle import train_parity as tp
    t0 = time.perf_counter()
    task = tp.make_task(steps=steps, B=B)
    st0 = tp.init_state(task["num_words"])
    cfg = make_cfg('NRMS', 'small', vocab=task["num_words"])
    wl = Workload('NRMS', cfg)
    crit = torch.nn.CrossEntropyLoss()
    target = torch.zeros(B, dtype=torch.long, device=device)
    cand_d = torch.from_numpy(task["cand_ids"]).to(device)
    click_d = torch.from_numpy(task["click_ids"]).to(device)
    es = ({'title': task["titles"]}, task["eval_hist"], task["eval_cands"], task["eval_ptr"], None)

    def engine_run(seed):
        m = wl.make_model().to(device)
        m.load_state_dict(st0)
        m.train()
        opt = EngineAdam(m, lr=lr)
        torch.manual_seed(1000 + seed)                   # ops.new_seed() draws the kernels' dropout seeds from torch's CPU generator
        losses = []
        for i in range(steps):
            loss = crit(m.forward_ids(cand_d[i], click_d[i]), target)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        sc = engine_scores(wl, m, device, es)
        return tp.eval_metrics(task, sc), float(torch.stack(losses[-10:]).mean())
    out = {"steps": steps, "batch": B, "lr": lr, "dropout": cfg.dropout_probability, "vocab": task["num_words"], "eval_impressions": len(task["eval_ptr"]) - 1,
           "auc_teacher": float(tp.eval_metrics(task, task["teacher_scores"])[0])}
    runs = [engine_run(s_) for s_ in engine_seeds]
    out["engine"] = [{"auc": float(r[0][0]), "ndcg10": float(r[0][3]), "last10_loss": r[1]} for r in runs]
    out["engine_seed_spread_auc"] = float(max(r["auc"] for r in out["engine"]) - min(r["auc"] for r in out["engine"]))
    ops.invalidate_packed()
    if oracle:
        init_m = tp.eval_metrics(task, tp.oracle_eval_scores(task, st0))
        out["auc_init"] = float(init_m[0])
        # the oracle on TWO dropout streams as well: its own seed-to-seed spread is the yardstick of the comparison, and it is in the record
        oruns = []
        for os_ in oracle_seeds:
            trained, losses = tp.train_oracle(task, st0, lr=lr, p_drop=cfg.dropout_probability, torch_seed=os_)
            om = tp.eval_metrics(task, tp.oracle_eval_scores(task, trained))
            oruns.append({"auc": float(om[0]), "ndcg10": float(om[3]), "last10_loss": float(np.mean(losses[-10:])), "torch_seed": os_})
        out["oracle_runs"] = oruns
        out["oracle"] = {k: float(np.mean([r[k] for r in oruns])) for k in ("auc", "ndcg10", "last10_loss")}
        out["oracle_seed_spread_auc"] = float(max(r["auc"] for r in oruns) - min(r["auc"] for r in oruns))
        out["oracle_seed_spread_ndcg10"] = float(max(r["ndcg10"] for r in oruns) - min(r["ndcg10"] for r in oruns))
        ea = float(np.mean([r["auc"] for r in out["engine"]]))
        en = float(np.mean([r["ndcg10"] for r in out["engine"]]))
        out["abs_diff_auc"] = abs(ea - out["oracle"]["auc"])
        out["abs_diff_ndcg10"] = abs(en - out["oracle"]["ndcg10"])
        # two trainings with different dropout streams differ by the seed-to-seed spread: the means of two runs each may differ by about the
        # larger of the two recorded spreads (floor 5e-3: two draws under-estimate a spread)
        out["tolerance_auc"] = float(max(5e-3, 1.5 * max(out["engine_seed_spread_auc"], out["oracle_seed_spread_auc"])))
        out["within_noise"] = bool(out["abs_diff_auc"] < out["tolerance_auc"] and ea > out["auc_init"] + 0.1)
    out["seconds"] = time.perf_counter() - t0
    return out


def train_parity_naml(device, steps=100, B=16, lr=1e-3, engine_seeds=(0, 1), oracle_seeds=(0,)):
    """The NAML leg of the statistical training parity (see train_parity): conv text encoders over titles and abstracts, the category views,
    three levels of additive attention -- engine (bf16 operands, counter dropout, EngineAdam) vs OracleNAML (torch dropout, torch.optim.Adam)
    from the same initial weights on the same teacher-labelled impressions, compared by the AUC / nDCG@10 of the two trained models on held-out
    impressions.  Unlike the NRMS task this one does not start at chance (two random NAML models already agree on which candidates share a
    category with the history: AUC_init ~0.7), so what is asserted is the gain over AUC_init and the agreement of the trained models."""
    from news_recommendation_amd import ops
    from news_recommendation_amd.optim import EngineAdam
    from oracle import train_parity as tp
    t0 = time.perf_counter()
    task = tp.make_task_naml(steps=steps, B=B)
    st0 = tp.init_state_naml(task["num_words"], task["num_categories"])
    cfg = make_cfg('NAML', 'small', vocab=task["num_words"])
    cfg.num_categories = task["num_categories"]
    wl = Workload('NAML', cfg)
    crit = torch.nn.CrossEntropyLoss()
    target = torch.zeros(B, dtype=torch.long, device=device)
    to_dev = lambda d: {k: torch.from_numpy(v).to(device) for k, v in d.items()}
    batches = [tuple(to_dev(x) for x in tp.naml_batch(task, i)) for i in range(steps)]
    es = (task["news"], task["eval_hist"], task["eval_cands"], task["eval_ptr"], None)

    def engine_run(seed):
        m = wl.make_model().to(device)
        m.load_state_dict(st0)
        m.train()
        opt = EngineAdam(m, lr=lr)
        torch.manual_seed(1000 + seed)
        losses = []
        for cand, click in batches:
            loss = crit(m.forward_ids(cand, click), target)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        return tp.eval_metrics(task, engine_scores(wl, m, device, es)), float(torch.stack(losses[-10:]).mean())
    out = {"model": "NAML", "steps": steps, "batch": B, "lr": lr, "dropout": cfg.dropout_probability, "vocab": task["num_words"],
           "eval_impressions": len(task["eval_ptr"]) - 1, "auc_teacher": float(tp.eval_metrics(task, task["teacher_scores"])[0]),
           "auc_init": float(tp.eval_metrics(task, tp.oracle_eval_scores_naml(task, st0))[0])}
    runs = [engine_run(s_) for s_ in engine_seeds]
    out["engine"] = [{"auc": float(r[0][0]), "ndcg10": float(r[0][3]), "last10_loss": r[1]} for r in runs]
    out["engine_seed_spread_auc"] = float(max(r["auc"] for r in out["engine"]) - min(r["auc"] for r in out["engine"]))
    ops.invalidate_packed()
    oruns = []
    for os_ in oracle_seeds:
        trained, losses = tp.train_oracle_naml(task, st0, lr=lr, p_drop=cfg.dropout_probability, torch_seed=os_)
        om = tp.eval_metrics(task, tp.oracle_eval_scores_naml(task, trained))
        oruns.append({"auc": float(om[0]), "ndcg10": float(om[3]), "last10_loss": float(np.mean(losses[-10:])), "torch_seed": os_})
    out["oracle_runs"] = oruns
    out["oracle"] = {k: float(np.mean([r[k] for r in oruns])) for k in ("auc", "ndcg10", "last10_loss")}
    ea = float(np.mean([r["auc"] for r in out["engine"]]))
    out["abs_diff_auc"] = abs(ea - out["oracle"]["auc"])
    out["abs_diff_ndcg10"] = abs(float(np.mean([r["ndcg10"] for r in out["engine"]])) - out["oracle"]["ndcg10"])
    # one oracle stream only (a CPU NAML training costs minutes of GPU-box time): the yardstick is the engine's own seed-to-seed spread, floor 1e-2
    out["tolerance_auc"] = float(max(1e-2, 2.0 * out["engine_seed_spread_auc"]))
    out["within_noise"] = bool(out["abs_diff_auc"] < out["tolerance_auc"])
    out["seconds"] = time.perf_counter() - t0
    return out


def train_parity_lstur(device, steps=100, B=16, lr=1e-3, engine_seeds=(0, 1), oracle_seeds=(0, 1)):
    """The LSTUR leg of the statistical training parity: title CNN + category views, the GRU over the click history initialised from the
    per-user row, whole-row user masking (p = 0.5) and dropout on -- engine (persistent GRU sweeps on MI355X, bf16 operands, row-sparse lazy
    Adam for the user table) vs OracleLSTUR (torch.optim.Adam) from the same initial weights on the same teacher-labelled impressions.  The
    task is noisier than the other two (every step masks half of 16 user rows): both sides run two seeds and the tolerance is 2 x the larger
    recorded seed spread (floor 1.5e-2)."""
    from news_recommendation_amd import ops, ops_gru
    from news_recommendation_amd.optim import EngineAdam
    from oracle import train_parity as tp
    t0 = time.perf_counter()
    task = tp.make_task_lstur(steps=steps, B=B)
    st0 = tp.init_state_lstur(task["num_words"], task["num_categories"], task["num_users"])
    cfg = make_cfg('LSTUR', 'small', vocab=task["num_words"])
    cfg.num_categories, cfg.num_users = task["num_categories"], task["num_users"]
    wl = Workload('LSTUR', cfg)
    crit = torch.nn.CrossEntropyLoss()
    target = torch.zeros(B, dtype=torch.long, device=device)
    to_dev = lambda d: {k: torch.from_numpy(v).to(device) for k, v in d.items()}
    batches = []
    for i in range(steps):
        cand, click, user, length = tp.lstur_batch(task, i)
        batches.append((to_dev(cand), to_dev(click), torch.from_numpy(user).to(device), torch.from_numpy(length)))
    es = (task["news"], task["eval_hist"], task["eval_cands"], task["eval_ptr"], task["eval_users"])

    def engine_run(seed):
        m = wl.make_model().to(device)
        m.load_state_dict(st0)
        m.train()
        opt = EngineAdam(m, lr=lr, row_sparse=('user_embedding.weight',))
        torch.manual_seed(1000 + seed)
        losses = []
        for cand, click, user, length in batches:
            loss = crit(m.forward_ids(user, length.clone(), cand, click), target)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        opt.flush()                          # every user row current before the table is read as a whole by the scoring
        sc = engine_scores(wl, m, device, es)
        ops_gru.persist_check()
        return tp.eval_metrics(task, sc), float(torch.stack(losses[-10:]).mean())
    out = {"model": "LSTUR", "steps": steps, "batch": B, "lr": lr, "dropout": cfg.dropout_probability, "masking_probability": cfg.masking_probability,
           "vocab": task["num_words"], "users": task["num_users"], "eval_impressions": len(task["eval_ptr"]) - 1,
           "auc_teacher": float(tp.eval_metrics(task, task["teacher_scores"])[0]),
           "auc_init": float(tp.eval_metrics(task, tp.oracle_eval_scores_lstur(task, st0))[0])}
    runs = [engine_run(s_) for s_ in engine_seeds]
    out["engine"] = [{"auc": float(r[0][0]), "ndcg10": float(r[0][3]), "last10_loss": r[1]} for r in runs]
    out["engine_seed_spread_auc"] = float(max(r["auc"] for r in out["engine"]) - min(r["auc"] for r in out["engine"]))
    ops.invalidate_packed()
    oruns = []
    for os_ in oracle_seeds:
        trained, losses = tp.train_oracle_lstur(task, st0, lr=lr, p_drop=cfg.dropout_probability, pm=cfg.masking_probability, torch_seed=os_)
        om = tp.eval_metrics(task, tp.oracle_eval_scores_lstur(task, trained))
        oruns.append({"auc": float(om[0]), "ndcg10": float(om[3]), "last10_loss": float(np.mean(losses[-10:])), "torch_seed": os_})
    out["oracle_runs"] = oruns
    out["oracle"] = {k: float(np.mean([r[k] for r in oruns])) for k in ("auc", "ndcg10", "last10_loss")}
    out["oracle_seed_spread_auc"] = float(max(r["auc"] for r in oruns) - min(r["auc"] for r in oruns))
    ea = float(np.mean([r["auc"] for r in out["engine"]]))
    out["abs_diff_auc"] = abs(ea - out["oracle"]["auc"])
    out["abs_diff_ndcg10"] = abs(float(np.mean([r["ndcg10"] for r in out["engine"]])) - out["oracle"]["ndcg10"])
    out["tolerance_auc"] = float(max(1.5e-2, 2.0 * max(out["engine_seed_spread_auc"], out["oracle_seed_spread_auc"])))
    out["within_noise"] = bool(out["abs_diff_auc"] < out["tolerance_auc"])
    out["seconds"] = time.perf_counter() - t0
    return out


def train_parity_fixture(device, model_name, engine_seeds=8, fixture_dir=None, optimizer='engine', p_drop=None, oracle_scored=False):
    """Statistical training parity against the REAL reference.  tests/golden/train_parity/<model>.npz (written in the build
    container by oracle/make_golden_train_parity.py) holds the teacher-labelled task and the held-out metrics of the reference's OWN model class
    trained on it with torch's dropout and torch.optim.Adam, one run per torch seed (8).  Here the ENGINE (bf16 operands, counter-based dropout,
    EngineAdam) trains on the same batches from the same initial weights, `engine_seeds` dropout streams, and the two samples are compared:
    |mean_e - mean_r| against 3 standard errors of the difference, sqrt(s_e^2 / n_e + s_r^2 / n_r), from the MEASURED spreads -- no floor.  The
    sign of the difference is in the record (r05 saw the engine below the oracle in 4 of 4 pairings with two seeds a side)."""
    from news_recommendation_amd import ops, ops_gru
    from news_recommendation_amd.optim import EngineAdam
    from oracle import train_parity as tp
    t0 = time.perf_counter()
    fixture_dir = fixture_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'train_parity')
    z = np.load(os.path.join(fixture_dir, f'{model_name.lower()}.npz'))
    meta = json.loads(str(z['meta']))
    task = tp.task_from_arrays(z)
    steps, B, lr = task["steps"], task["B"], meta["lr"]
    if model_name == 'NRMS':
        st0 = tp.init_state(task["num_words"])
    elif model_name == 'NAML':
        st0 = tp.init_state_naml(task["num_words"], task["num_categories"])
    else:
        st0 = tp.init_state_lstur(task["num_words"], task["num_categories"], task["num_users"])
    if not np.allclose(tp.state_checksum(st0), z['init_checksum'], rtol=1e-9, atol=1e-12):
        raise RuntimeError(f"train_parity_fixture[{model_name}]: the seeded initial state differs from the one the reference was trained from")
    cfg = make_cfg(model_name, 'small', vocab=task["num_words"])
    for k in ("num_categories", "num_users"):
        if k in task:
            setattr(cfg, k, task[k])
    wl = Workload(model_name, cfg)
    crit = torch.nn.CrossEntropyLoss()
    target = torch.zeros(B, dtype=torch.long, device=device)
    to_dev = lambda d: {k: torch.from_numpy(v).to(device) for k, v in d.items()}
    if model_name == 'NRMS':
        cand_d, click_d = torch.from_numpy(task["cand_ids"]).to(device), torch.from_numpy(task["click_ids"]).to(device)
        fwd = [lambda m, i=i: m.forward_ids(cand_d[i], click_d[i]) for i in range(steps)]
        es = ({'title': task["titles"]}, task["eval_hist"], task["eval_cands"], task["eval_ptr"], None)
    elif model_name == 'NAML':
        bs = [tuple(to_dev(x) for x in tp.naml_batch(task, i)) for i in range(steps)]
        fwd = [lambda m, b=b: m.forward_ids(b[0], b[1]) for b in bs]
        es = (task["news"], task["eval_hist"], task["eval_cands"], task["eval_ptr"], None)
    else:
        bs = []
        for i in range(steps):
            cand, click, user, length = tp.lstur_batch(task, i)
            bs.append((to_dev(cand), to_dev(click), torch.from_numpy(user).to(device), torch.from_numpy(length)))
        fwd = [lambda m, b=b: m.forward_ids(b[2], b[3].clone(), b[0], b[1]) for b in bs]
        es = (task["news"], task["eval_hist"], task["eval_cands"], task["eval_ptr"], task["eval_users"])

    if p_drop is not None:                               # diagnostic variants (tools/train_parity_diag.py): another dropout probability
        cfg.dropout_probability = p_drop
        wl = Workload(model_name, cfg)
    extra = []

    def engine_run(seed):
        m = wl.make_model().to(device)
        m.load_state_dict(st0)
        m.train()
        if optimizer == 'torch':                         # diagnostic: the reference's own optimiser on the engine's gradients
            opt = torch.optim.Adam(m.parameters(), lr=lr)
        else:
            opt = EngineAdam(m, lr=lr, row_sparse=('user_embedding.weight',) if model_name == 'LSTUR' else ())
        torch.manual_seed(1000 + seed)                   # ops.new_seed() draws the kernels' dropout seeds from torch's CPU generator
        losses = []
        for f in fwd:
            loss = crit(f(m), target)
            if optimizer == 'torch':
                opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        if model_name == 'LSTUR' and optimizer != 'torch':
            opt.flush()
        sc = engine_scores(wl, m, device, es)
        if model_name == 'LSTUR':
            ops_gru.persist_check()
        if oracle_scored and model_name == 'NRMS':       # the SAME trained weights ranked by the CPU oracle: separates training from scoring
            sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            extra.append([float(x) for x in tp.eval_metrics(task, tp.oracle_eval_scores(task, sd))])
        return [float(x) for x in tp.eval_metrics(task, sc)], float(torch.stack(losses[-10:]).mean())
    seeds = list(range(engine_seeds)) if isinstance(engine_seeds, int) else list(engine_seeds)
    runs = [engine_run(s_) for s_ in seeds]
    ops.invalidate_packed()
    em = np.array([r[0] for r in runs])
    rm = z['ref_metrics']
    out = {"model": model_name, "steps": steps, "batch": B, "lr": lr, "dropout": meta["dropout"], "vocab": task["num_words"],
           "eval_impressions": len(task["eval_ptr"]) - 1, "reference": meta["reference"], "fixture": f"tests/golden/train_parity/{model_name.lower()}.npz",
           "auc_init": float(z['init_metrics'][0]), "auc_teacher": float(tp.eval_metrics(task, task["teacher_scores"])[0]),
           "engine_auc": [float(x) for x in em[:, 0]], "reference_auc": [float(x) for x in rm[:, 0]],
           "engine_ndcg10": [float(x) for x in em[:, 3]], "reference_ndcg10": [float(x) for x in rm[:, 3]],
           "engine_last10_loss": [r[1] for r in runs], "reference_last10_loss": [float(x) for x in z['ref_last10_loss']]}
    for tag, col in (("auc", 0), ("ndcg10", 3)):
        e, r = em[:, col], rm[:, col]
        se = float(np.sqrt(e.var(ddof=1) / len(e) + r.var(ddof=1) / len(r)))
        out[f"mean_engine_{tag}"], out[f"mean_reference_{tag}"] = float(e.mean()), float(r.mean())
        out[f"sd_engine_{tag}"], out[f"sd_reference_{tag}"] = float(e.std(ddof=1)), float(r.std(ddof=1))
        out[f"diff_{tag}"] = float(e.mean() - r.mean())                     # signed: negative = the engine's trained models rank worse
        out[f"stderr_diff_{tag}"] = se
        out[f"z_{tag}"] = float((e.mean() - r.mean()) / max(se, 1e-12))
    if extra:
        out["engine_weights_scored_by_oracle_auc"] = [x[0] for x in extra]
    out["optimizer"], out["dropout"] = optimizer, cfg.dropout_probability
    out["engine_below_reference_pairs"] = int((em[:, 0][:, None] < rm[:, 0][None, :]).sum())
    out["pairs"] = int(em.shape[0] * rm.shape[0])
    out["within_3_stderr"] = bool(abs(out["z_auc"]) < 3.0 and abs(out["z_ndcg10"]) < 3.0)
    out["seconds"] = time.perf_counter() - t0
    return out


def score_eval(wl, model, device, n_impr_cap=100000):
    """Eval-shaped scoring throughput: phases A (encode every news once), B (one user vector per impression history), C
    (ragged candidate scoring + per-impression AUC / MRR / nDCG on the device) of src/evaluate.py:185-272 via evaluate_fast.run_plan,
    on a synthetic plan of the dataset shape (all news of the shape, min(eval impressions of the shape, cap) impressions)."""
    from news_recommendation_amd import evaluate_fast, synth
    cfg = wl.cfg
    n_news, n_impr = cfg.num_news, min(cfg.eval_impressions, n_impr_cap)
    rng = np.random.default_rng(11)
    news = wl.news()
    hist, cands, ptr = synth.eval_impressions(rng, n_news, n_impr)
    plan = evaluate_fast.EvalPlan()
    plan.news_ids = [f'N{i}' for i in range(n_news)]
    plan.news = {k: news[k] for k in cfg.dataset_attributes['news']}
    plan.hist_idx = np.where(hist < 0, n_news, hist).astype(np.int64)
    plan.hist_len = (hist >= 0).sum(1).astype(np.int64)
    plan.hist_user = rng.integers(1, cfg.num_users, size=n_impr).astype(np.int64)
    plan.cand_idx, plan.cand_ptr = cands, ptr
    plan.labels = (rng.random(len(cands)) < 0.1).astype(np.int32)
    plan.imp_user_row = np.arange(n_impr, dtype=np.int32)
    was_training = model.training
    model.eval()
    evaluate_fast.run_plan(model, plan, 2048, wl.name)          # warm-up
    torch.cuda.synchronize()
    dts = []
    for _ in range(3):                      # best of three: the host half (index arrays, 128-thread pool wake-ups) jitters by 3x between runs
        t0 = time.perf_counter()
        out, _ = evaluate_fast.run_plan(model, plan, 2048, wl.name)
        m = torch.nanmean(out.double(), dim=0).cpu()
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    # phase C's two kernels against the HBM roof (K7 is bandwidth-bound; per impression C x d x 4 + d x 4 + C x (4 B index + 4 B out)
    # = 46.5 KB at C = 37.5), HIP events on the launch stream, 5 launches each on the plan's own arrays
    from news_recommendation_amd import ops
    out, scores = evaluate_fast.run_plan(model, plan, 2048, wl.name)
    nnz = int(len(cands))
    with ops.profile(only={'nr_score_csr', 'nr_impression_metrics'}) as rec:
        for _ in range(5):
            evaluate_fast.phase_c(model, plan, wl.name)
    rs = rec.summary()
    D = evaluate_fast.phase_c.last_dim
    bytes_csr = nnz * (D * 4 + 4 + 4) + n_impr * (D * 4 + 8 + 4)
    bytes_met = nnz * (4 + 4) + n_impr * (8 + 16)
    roof = {}
    if 'nr_score_csr' in rs:
        a = bytes_csr / (rs['nr_score_csr'][1] * 1e-6) / 1e9
        roof["nr_score_csr"] = {"bound": "hbm", "avg_us": rs['nr_score_csr'][1], "bytes_per_launch": bytes_csr, "bytes_per_impression": bytes_csr / n_impr,
                                "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                                "note": "the news matrix (%.0f MB) is Infinity-Cache resident: candidate rows are re-read from MALL, not HBM" % (n_news * D * 4 / 1e6)}
    if 'nr_impression_metrics' in rs:
        a = bytes_met / (rs['nr_impression_metrics'][1] * 1e-6) / 1e9
        roof["nr_impression_metrics"] = {"bound": "hbm (latency: one wave per impression, rank by comparison)", "avg_us": rs['nr_impression_metrics'][1],
                                         "bytes_per_launch": bytes_met, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS}
    # fp32 device metrics vs the float64 numpy metrics of the reference's formulas (oracle/metrics.py) on the same device scores
    from oracle import metrics as om
    sc = scores.cpu().numpy().astype(np.float64)
    k = min(n_impr, 5000)
    ref = np.array([om.single_impression_metrics(plan.labels[ptr[i]:ptr[i + 1]], sc[ptr[i]:ptr[i + 1]]) for i in range(k)], dtype=np.float64)
    dev_m = out[:k].double().cpu().numpy()
    both = ~np.isnan(ref) & ~np.isnan(dev_m)
    dev_dev = {"impressions": k, "max_abs_per_impression": float(np.abs(np.where(both, dev_m - ref, 0.0)).max()),
               "abs_diff_of_means": [float(abs(np.nanmean(dev_m[:, j]) - np.nanmean(ref[:, j]))) for j in range(4)],
               "nan_pattern_equal": bool((np.isnan(ref) == np.isnan(dev_m)).all()),
               "what": "device metrics (fp32: log2f, 1.0f / rank) vs float64 numpy of src/evaluate.py:24-42,160-168 on the same scores: AUC, MRR, nDCG@5, nDCG@10"}
    model.train(was_training)
    return {"value": n_impr / dt, "unit": "impressions/s", "impressions": n_impr, "news": n_news, "candidates": int(len(cands)),
            "seconds": dt, "seconds_all_runs": dts, "roofline": roof, "metrics_fp32_vs_f64": dev_dev,
            "what": "phases A+B+C of src/evaluate.py:185-272 (batched driver), host index arrays -> four metric means; best of 3 runs"}


def comm_probe(opt, step, barrier, world, k, ms_step, device):
    """N > 1 only, every rank: (1) the same training step with every collective skipped (EngineAdam.skip_comm) -> exposed_comm_ms = step time -
    that; (2) each bucket's collective on its own, HIP-event timed on the group's stream order: payload bytes, algorithmic GB/s and bus GB/s
    (x 2 (n - 1) / n for all-reduce, x (n - 1) / n for reduce-scatter / all-gather: the per-link load a ring puts on xGMI)."""
    import torch.distributed as dist
    res = {}
    opt.skip_comm = True
    for i in range(2):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(k):
        step(i)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    opt.skip_comm = False
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["ms_per_step_without_comm"] = float(t.item()) / k * 1e3
    res["exposed_comm_ms"] = ms_step - res["ms_per_step_without_comm"]
    opt.discard_grads()

    def timed(fn, reps=5):
        fn()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        tt = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    rank = dist.get_rank()
    buckets = {}
    for r in opt.regions:
        if opt.table_rs and r.name != 'small':
            sh = (r.end - r.lo) // world
            shard = opt.flat_g[r.lo + rank * sh:r.lo + (rank + 1) * sh]
            full = opt.flat_g[r.lo:r.end]

            def fn(shard=shard, full=full):
                dist.reduce_scatter_tensor(shard, full)
                dist.all_gather_into_tensor(full, shard)
            nbytes, factor, kind = (r.end - r.lo) * 4, 2.0 * (world - 1) / world, "reduce_scatter+all_gather"
        else:
            buf = opt.flat_g[r.lo:r.hi]

            def fn(buf=buf):
                dist.all_reduce(buf)
            nbytes, factor, kind = (r.hi - r.lo) * 4, 2.0 * (world - 1) / world, "all_reduce"
        ms = timed(fn)
        buckets[r.name] = {"collective": kind, "bytes": nbytes, "ms": ms, "alg_GBs": nbytes / ms / 1e6, "bus_GBs": nbytes * factor / ms / 1e6}
    for st in opt.sparse:
        cap = opt._row_cap.get(st.name, 0)
        if cap:
            d = st.param.shape[1]
            rows = torch.zeros(cap, d, device=device)
            out = torch.empty(world * cap, d, device=device)
            ms = timed(lambda: dist.all_gather_into_tensor(out, rows))
            nbytes = world * cap * d * 4
            buckets[st.name] = {"collective": "all_gather (touched rows)", "bytes": nbytes, "ms": ms, "alg_GBs": nbytes / ms / 1e6,
                                "bus_GBs": nbytes * (world - 1) / world / ms / 1e6}
    opt.flat_g.zero_()
    res["buckets"] = buckets
    return res


def gather_point(lib, table, ids, device):
    st = torch.cuda.current_stream().cuda_stream
    gout = torch.empty(ids.numel(), 300, device=device)
    for _ in range(3):
        lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), gout.data_ptr(), ids.numel(), 300, table.shape[0], st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), gout.data_ptr(), ids.numel(), 300, table.shape[0], st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    nbytes = ids.numel() * (BYTES_PER_TOKEN_F32 + 8)
    return {"avg_us": us, "tokens": ids.numel(), "table_mb": table.numel() * 4 / 1e6, "algorithmic_bytes": nbytes,
            "achieved": nbytes / (us * 1e-6) / 1e9, "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "read_plus_write_GBs": (nbytes + ids.numel() * BYTES_PER_TOKEN_F32) / (us * 1e-6) / 1e9}


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ('k_mhsa_fwd2.h', 'k_mhsa_fwd.h', 'k_additive_fwd.h', 'k_bwd.h', 'k_attn_bwd2.h', 'k_proj.h', 'k_gemm.h', 'k_pool2.h', 'k_pool3.h', 'k_misc.h', 'k_conv.h', 'k_gru.h',
              'k_gru_persist.h', 'k_xcd.h', 'k_pool4.h', 'k_convgemm.h', 'k_optim.h', 'k_eval.h', 'k_step.h', 'nr_common.h', 'nr_prims.h'):
        with open(os.path.join(ROOT, 'news_recommendation_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def gather_source_hash():
    """Hash of the stand-alone gather probe's kernel source only (the text of gather_rows_kernel in csrc/k_misc.h): profiles/gather_traffic.json
    stays valid while other kernels of that file change."""
    with open(os.path.join(ROOT, 'news_recommendation_amd', 'csrc', 'k_misc.h')) as fh:
        src = fh.read()
    a = src.index('void gather_rows_kernel(')
    return hashlib.sha256(src[a:src.index('\n}\n', a)].encode()).hexdigest()[:16]


def graph_args(wl, model, crit, target):
    """(flat, loss_of): flat(batch) = the device tensors a captured step takes as arguments, loss_of(*those) = the step's scalar loss."""
    name, n = wl.name, len(wl.attrs)
    tail = (lambda b: [b['user'], b['length_dev']]) if name == 'LSTUR' else (lambda b: [])
    if CRITERION_STEP:
        flat = lambda b: [b[s_][a] for s_ in ('cand', 'click') for a in wl.attrs] + tail(b)

        def loss_of(*xs):
            cand, click = dict(zip(wl.attrs, xs[:n])), dict(zip(wl.attrs, xs[n:2 * n]))
            if name == 'LSTUR':
                lg_ = model.forward_ids(xs[2 * n], xs[2 * n + 1].clone(), cand, click)
            else:
                lg_ = model.forward_ids(cand['title'], click['title']) if name == 'NRMS' else model.forward_ids(cand, click)
            return crit(lg_, target)
        return flat, loss_of
    B, C = target.shape[0], 1 + wl.cfg.negative_sampling_ratio
    flat = lambda b: [b['ids'][a] for a in wl.attrs] + tail(b)

    def loss_of(*xs):
        ids = dict(zip(wl.attrs, xs[:n]))
        if name == 'LSTUR':
            return model.forward_stacked(xs[n], xs[n + 1].clone(), ids, B, C, loss=True)
        return model.forward_stacked(ids['title'] if name == 'NRMS' else ids, B, C, loss=True)
    return flat, loss_of


def build_step_graph(wl, model, opt, crit, target, batches, device):
    """The training step of workload wl as ONE HIP graph (forward + backward + Adam; news_recommendation_amd/graph.py).  Returns (graph, flat):
    flat(batch) is the graph's argument list for a batch."""
    from news_recommendation_amd.graph import StepGraph
    name = wl.name
    if name == 'LSTUR':          # the captured step keeps user ids and history lengths on the device (graph.py)
        for b in batches:
            b['length_dev'] = b['length'].to(device)
    flat, loss_of = graph_args(wl, model, crit, target)

    def step_fn(*xs):
        l_ = loss_of(*xs)
        l_.backward()
        opt.step()
        return l_
    return StepGraph(step_fn, flat(batches[0]), opt, warmup=1), flat


def other_workload(name, shape, vocab, B, device, steps=10):
    """A short graph-replay leg of another BASELINE workload on this GPU, for the driver's own record (the default line is NRMS): the same
    protocol as the headline -- one HIP graph per step, `steps` replays between synchronisations -- plus the dominant hand-written kernel of two
    profiled eager steps, priced against its roofline like the headline's."""
    from news_recommendation_amd import ops
    cfg = make_cfg(name, shape, vocab)
    wl = Workload(name, cfg)
    model = wl.make_model().to(device).train()
    opt = wl.make_optimizer(model)
    crit = torch.nn.CrossEntropyLoss()
    batches = wl.batches(0, 4, B, device)
    target = torch.zeros(B, dtype=torch.long, device=device)

    def step(i):
        loss = wl.loss(model, batches[i % len(batches)], crit, target)
        loss.backward()
        opt.step()
        return loss
    for i in range(3):
        step(i)
    with ops.profile() as rec:
        for i in range(2):
            step(i)
    prof = rec.summary()
    hand = {k: v for k, v in prof.items() if k.startswith('nr_') and not k.startswith(('nr_pack', 'nr_sort', 'nr_adam', 'nr_row_adam'))}
    dominant = max(hand, key=lambda k: hand[k][2])
    sg, flat = build_step_graph(wl, model, opt, crit, target, batches, device)
    for i in range(3):
        sg(*flat(batches[i % len(batches)]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = sg(*flat(batches[i % len(batches)]))
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lossv = float(loss.item())
    sg.close()
    if name == 'LSTUR':
        from news_recommendation_amd import ops_gru
        ops_gru.persist_check()             # the persistent GRU sweeps of the replays above came out clean
    flops, hbm = wl.flops(B), wl.hbm_bytes(B)
    dom = prof[dominant]                    # (launches, avg us, total us) over the two profiled eager steps (HIP events on the launch stream)
    if dominant in hbm:
        ach = hbm[dominant] / (dom[1] * 1e-6) / 1e9
        roof = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "avg_us": dom[1], "bytes_per_launch": hbm[dominant]}
    elif dominant in flops:
        ach = flops[dominant] / (dom[1] * 1e-6) / 1e12
        roof = {"kernel": dominant, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TF,
                "avg_us": dom[1], "flop_per_launch": flops[dominant]}
    else:
        roof = {"kernel": dominant, "avg_us": dom[1], "note": "no algorithmic figure tabulated for this kernel"}
    if dominant.startswith('nr_gru_'):
        # one launch = the whole sweep (N steps / N + 1 calls); its 2.5 GFLOP per step are ~1 us of matrix work -- what bounds it is the
        # operand traffic out of each XCD's L2 and the inter-workgroup wait: the MFMA fraction is reported, not claimed as the bound
        roof["note"] = "sweep kernel: bound by L2 -> CU operand traffic and the XCD-local wait, not by the matrix pipe (DESIGN.md 5.3b)"
    roof["launches_per_step"] = dom[0] // 2
    traffic_k, busy_k = committed_counters(name, shape, B)
    roof["top3"] = [price_kernel(k, hand[k], flops, hbm, traffic_k, busy_k) for k in sorted(hand, key=lambda k: -hand[k][2])[:3]]
    tag = {('NAML', 'small'): "BASELINE.json configs[2]", ('LSTUR', 'large'): "single-GPU shard of BASELINE.json configs[4]",
           ('NRMS', 'large'): "single-GPU shard of BASELINE.json configs[3]"}.get((name, shape), "")
    out = {"workload": f"{name} bf16, MIND-{shape}-shaped synthetic, batch {B} ({tag})", "value": B * steps / dt, "unit": "impressions/s",
           "ms_per_step": dt / steps * 1e3, "host_enqueue_ms_per_step": t_enq / steps * 1e3, "steps": steps, "loss": lossv, "roofline": roof,
           "kernel_breakdown_us_per_step": {k: round(v[2] / 2, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])[:12]}}
    del sg, opt, model, batches
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch (impressions)')
    ap.add_argument('--model', default='NRMS', choices=['NRMS', 'NAML', 'LSTUR'])
    ap.add_argument('--shape', default=None, choices=['small', 'large', 'xlarge'],
                    help='dataset shape; default: small on 1 GPU (BASELINE configs[1]/[2]), large on N > 1 (configs[3]/[4])')
    ap.add_argument('--vocab', type=int, default=0, help='override the vocabulary size of the shape (rows of the word-embedding table)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-train-parity', action='store_true', help='skip the training-parity leg (8 engine trainings vs the committed runs of the reference, ~30 s)')
    ap.add_argument('--parity-seeds', type=int, default=8, help='independent n = 1000 evaluation sets per weight state of the parity leg')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the NAML / LSTUR graph-replay legs of the default line (~30 s)')
    ap.add_argument('--no-extras', action='store_true', help='skip value_dropin / score_eval / gather points (quick A/B timing runs)')
    ap.add_argument('--seg-overlap', type=int, default=1, choices=[0, 1],
                    help='N > 1: 1 = three graph segments, the table exchange in flight under the weight-gradient segment (default); 0 = round 4 two segments (A/B)')
    ap.add_argument('--no-graph', action='store_true',
                    help='issue the step kernel by kernel (default on 1 GPU: one HIP graph of forward + backward + Adam, replayed)')
    args = ap.parse_args()

    import torch.distributed as dist
    from news_recommendation_amd import dist as nrdist, ops, _capi
    rank, world, local = nrdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    B = args.batch
    shape = args.shape or ('small' if world == 1 else 'large')
    cfg = make_cfg(args.model, shape, args.vocab)
    wl = Workload(args.model, cfg)

    model = wl.make_model().to(device).train()
    nrdist.broadcast_parameters(model)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()} if (world == 1 and not args.no_parity) else None
    opt = wl.make_optimizer(model)
    crit = torch.nn.CrossEntropyLoss()
    batches = wl.batches(rank, 4, B, device)
    target = torch.zeros(B, dtype=torch.long, device=device)

    def step(i):
        loss = wl.loss(model, batches[i % len(batches)], crit, target)
        loss.backward()                       # gradients land in the optimiser's flat buffer; the table bucket's all-reduce starts inside
        opt.step()                            # small-bucket all-reduce, fused Adam (also clears the gradients: no zero_grad pass)
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up, then two un-timed profiled steps that find the dominant kernel (HIP events on the launch stream)
    for i in range(args.warmup):
        step(i)
    NPROF = 2
    with ops.profile() as rec:
        for i in range(NPROF):
            step(i)
    prof = rec.summary()
    assert opt.check_views(), "parameter / gradient views detached from the optimiser's flat buffers"
    hand = {k: v for k, v in prof.items() if k.startswith('nr_') and not k.startswith(('nr_pack', 'nr_sort', 'nr_adam', 'nr_row_adam'))}
    dominant = max(hand, key=lambda k: hand[k][2]) if hand else 'nr_mhsa_fwd[S=20]'
    top3 = sorted(hand, key=lambda k: -hand[k][2])[:3] or [dominant]        # all three are timed over the K steps and priced on both roofs

    # ---- single GPU: the step as ONE HIP graph (news_recommendation_amd/graph.py); the per-kernel HIP-event pass that the
    # roofline needs then runs as an eager pass of the same K steps AFTER the timed region (events cannot bracket nodes of a replayed graph)
    use_graph = world == 1 and not args.no_graph
    sg = None
    if use_graph:
        sg, flat = build_step_graph(wl, model, opt, crit, target, batches, device)
        for i in range(2):
            sg(*flat(batches[i % len(batches)]))

    # ---- N > 1: the data-parallel step as TWO HIP graphs with the RCCL collectives between them (graph.SegmentedStep); every rank takes the
    # same branch (same code, same arguments); a failure to build it falls back to the kernel-by-kernel step and says so in the line
    seg, seg_note = None, None
    if world > 1 and not args.no_graph:
        try:
            from news_recommendation_amd.graph import SegmentedStep
            if args.model == 'LSTUR':
                for b in batches:
                    b['length_dev'] = b['length'].to(device)
            flat, loss_of = graph_args(wl, model, crit, target)

            def fwd_bwd(*xs):
                l_ = loss_of(*xs)
                l_.backward()
                return l_
            seg = SegmentedStep(fwd_bwd, flat(batches[0]), opt, warmup=1, overlap=bool(args.seg_overlap))
        except Exception as e:               # noqa: BLE001 -- the measured line matters more than the issue mode
            seg, seg_note = None, f"segmented graphs unavailable ({e!r}): kernel-by-kernel step"
        # the ranks must agree on the issue mode (two graphs + exchange_all vs step() with overlap are DIFFERENT collective sequences): one
        # rank that failed to build its graphs (capture error, row capacity, memory) takes every rank to the kernel-by-kernel step
        ok = torch.tensor([1 if seg is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if seg is not None:
                seg.close()
                seg, seg_note = None, "segmented graphs unavailable on another rank: kernel-by-kernel step"
            opt.overlap = True
        else:
            for i in range(2):
                seg(*flat(batches[i % len(batches)]))

    barrier()
    t0 = time.perf_counter()
    timed_name = dominant                  # (a GRU sweep is one profiled entry: nr_gru_fwd_seq / nr_gru_bwd_seq)
    if sg is not None or seg is not None:
        run_ = sg if sg is not None else seg
        for i in range(args.steps):
            loss = run_(*flat(batches[i % len(batches)]))
        rec2 = None
    else:
        with ops.profile(only=set(top3)) as rec2:
            for i in range(args.steps):
                loss = step(i)
    t_enq = time.perf_counter() - t0       # host time to ENQUEUE the timed steps (launch-bound if it approaches dt)
    barrier()
    dt = time.perf_counter() - t0
    eager_ms = None
    if sg is not None:
        # the same K steps kernel by kernel (same counter protocol), HIP events around the dominant kernel: the roofline's duration
        loss_graph = float(loss.item())
        barrier()
        t1 = time.perf_counter()
        with ops.profile(only=set(top3)) as rec2:
            for i in range(args.steps):
                loss = sg.eager_step(*flat(batches[i % len(batches)]))
        barrier()
        eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
        sg.close()
    if args.model == 'LSTUR':
        from news_recommendation_amd import ops_gru
        ops_gru.persist_check()             # (after the timed region: the check synchronises) every persistent GRU sweep came out clean
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if seg is not None:
        # per-kernel HIP events cannot bracket graph nodes: the dominant kernel's duration comes from a short eager pass on the same counter
        # protocol (collective steps: every rank takes part)
        with ops.profile(only=set(top3)) as rec2:
            for i in range(min(args.steps, 5)):
                seg.eager_step(*flat(batches[i % len(batches)]))
        barrier()
    timed = rec2.summary()
    dom = timed.get(timed_name, (0, float('nan'), 0.0))

    comm = None
    if world > 1:
        # ---- how much of the step is EXPOSED gradient exchange, and what the buckets achieve on the wire (all ranks, still in the group) ----
        step_probe = (lambda i: seg(*flat(batches[i % len(batches)]))) if seg is not None else step
        comm = comm_probe(opt, step_probe, barrier, world, min(args.steps, 10), dt / args.steps * 1e3, device)
        if seg is not None:
            seg.close()
    if world > 1:
        # every rank leaves the process group here: what follows on rank 0 (roofline probes, scoring throughput) is local work, and a
        # rank that kept the group open would wait in its destructor for peers that are still measuring
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    T = B * (1 + cfg.negative_sampling_ratio + cfg.num_clicked_news_a_user)
    flops = wl.flops(B)
    hbm = wl.hbm_bytes(B)
    if dominant in hbm:
        ach = hbm[dominant] / (dom[1] * 1e-6) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_us": dom[1], "launches": dom[0], "bytes_per_launch": hbm[dominant]}
    elif dominant in flops:
        ach = flops[dominant] / (dom[1] * 1e-6) / 1e12
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / MFMA_BF16_PEAK_TF, "traffic": None, "avg_us": dom[1], "launches": dom[0],
                    "flop_per_launch": flops[dominant]}
    else:                                   # HBM-bound helper kernel dominant (gather / scatter): price by bytes
        nbytes = T * 20 * BYTES_PER_TOKEN_F32
        ach = nbytes / (dom[1] * 1e-6) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_us": dom[1], "launches": dom[0], "bytes_per_launch": nbytes}
    if dominant.startswith('nr_gru_'):
        # one launch = the whole sweep (N steps / N + 1 calls); its 2.5 GFLOP per step are ~1 us of matrix work -- what bounds it is the
        # operand traffic out of each XCD's L2 and the inter-workgroup wait: the MFMA fraction is reported, not claimed as the bound
        roofline["note"] = "sweep kernel: bound by L2 -> CU operand traffic and the XCD-local wait, not by the matrix pipe (DESIGN.md 5.3b)"
    # HBM traffic of that kernel: PMC counters cannot be read from inside the process, so the per-launch figure comes from the committed
    # rocprofv3 --pmc passes of the same workload (profiles/traffic.json, made by tools/pmc_traffic.sh); an entry only counts when it was
    # measured on the kernel sources this library was built from (source hash) and on this workload
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            tr = json.load(f).get(dominant)
        if tr is not None and tr.get("source_hash") == kernel_source_hash() and tr.get("workload") == f"{args.model}/{shape}/B{B}":
            roofline["traffic"] = tr["bytes"]
            roofline["traffic_source"] = tr["source"]
        elif tr is not None:
            roofline["traffic_note"] = "profiles/traffic.json entry is for other kernel sources / another workload: not reported"
    except (OSError, ValueError):
        pass
    # HBM traffic of the WHOLE step (rocprofv3 PMC over an eager run of the same workload: tools/pmc_step_traffic.py), same keying
    try:
        with open(os.path.join(ROOT, 'profiles', 'step_traffic.json')) as f:
            stt = json.load(f).get(f"{args.model}_{shape}")
        if stt is not None and stt.get("source_hash") == kernel_source_hash():
            roofline["traffic_total_per_step"] = stt["bytes_per_step"]
            roofline["traffic_top_kernels"] = stt["top"]
    except (OSError, ValueError):
        pass
    # MFMA-utilisation counters (north_star): SQ_VALU_MFMA_BUSY_CYCLES of the committed rocprofv3 --pmc passes, same keying (tools/pmc_sq.py)
    try:
        with open(os.path.join(ROOT, 'profiles', 'mfma_busy.json')) as f:
            mb = json.load(f)
        cur = {k: v for k, v in mb.items() if isinstance(v, dict) and v.get("source_hash") == kernel_source_hash()}
        if dominant in cur:
            roofline["mfma_busy_frac"] = cur[dominant]["mfma_busy_frac"]
        roofline["mfma_busy_frac_by_kernel"] = {k: round(v["mfma_busy_frac"], 4) for k, v in cur.items()} or None
    except (OSError, ValueError):
        pass
    # the three largest kernels of the step, each against BOTH roofs with its PMC traffic: the same record whichever of them is the arg-max
    traffic_k, busy_k = committed_counters(args.model, shape, B)
    roofline["top3"] = [price_kernel(k, timed[k], flops, hbm, traffic_k, busy_k) for k in top3 if k in timed]
    roofline["top3_note"] = ("avg_us: HIP events on the launch stream over the timed eager pass; frac_mfma = algorithmic flops / avg / 2.5 PFLOP/s, "
                             "frac_hbm = algorithmic bytes / avg / 8 TB/s (SURVEY d6 figures, padding not counted), traffic = PMC HBM bytes per launch "
                             "(profiles/traffic.json, only when measured on these kernel sources)")
    # the dense contraction of the forward on its own (the dominant kernel above may be a bandwidth-bound one)
    pj = prof.get('nr_qkv_proj_fwd[S=20]')
    if pj is not None:
        roofline["projection_gemm"] = {"kernel": "nr_qkv_proj_fwd[S=20]", "bound": "mfma", "avg_us": pj[1], "flop_per_launch": flops['nr_qkv_proj_fwd[S=20]'],
                                       "achieved": flops['nr_qkv_proj_fwd[S=20]'] / (pj[1] * 1e-6) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": flops['nr_qkv_proj_fwd[S=20]'] / (pj[1] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF,
                                       "timed_in": "the two profiled warm-up steps (HIP events)"}

    extras = {}
    if not args.no_extras:
        # ---- the embedding gather on its own (north_star: fraction of HBM roofline for the gather) -------------------------------------
        lib = _capi.load()
        b0 = batches[0]
        ids = torch.cat([b0['cand']['title'].reshape(-1, 20), b0['click']['title'].reshape(-1, 20)]).contiguous()
        table = next(p for n, p in model.named_parameters() if n.endswith('word_embedding.weight')).detach()
        g_mall = gather_point(lib, table, ids, device)
        big_rows = 400001                                                # 480 MB fp32: larger than the 256 MB Infinity Cache
        big = torch.randn(big_rows, 300, device=device)
        uni = torch.randint(1, big_rows, (ids.numel(),), device=device)
        g_hbm = gather_point(lib, big, uni, device)
        del big, uni
        extras["gather_roofline"] = {
            "kernel": "nr_gather_rows_f32", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": g_hbm["achieved"], "frac": g_hbm["frac"],
            "hbm_point": dict(g_hbm, ids="uniform over 400,001 rows", note="table larger than the Infinity Cache: rows come from HBM"),
            "workload_point": dict(g_mall, ids="the batch's title tokens (Zipf, 45 % padding id 0)",
                                   note=f"the workload's {table.numel() * 4 / 1e6:.0f} MB table is Infinity-Cache resident: this is MALL, not HBM, bandwidth"),
            "note": "achieved / frac count the algorithmic READ bytes (1200 B row + 8 B id per token); the stand-alone kernel also writes the "
                    "gathered rows to HBM (read_plus_write_GBs).  In NRMS training the gather is fused into nr_mhsa_fwd (rows go straight into "
                    "MFMA operand registers); this kernel is the north_star's stand-alone roofline probe"}
        # HBM traffic of the probe from the committed rocprofv3 passes (tools/gather_prof.sh -> profiles/gather_traffic.json), same keying as above
        try:
            with open(os.path.join(ROOT, 'profiles', 'gather_traffic.json')) as f:
                gt = json.load(f)
            if gt.get("hbm", {}).get("source_hash") == gather_source_hash():
                extras["gather_roofline"]["traffic"] = gt["hbm"].get("traffic_bytes")
                extras["gather_roofline"]["rocprof"] = {k: {kk: v[kk] for kk in ("avg_us_rocprof", "achieved_GBs", "frac_of_8TBs", "traffic_bytes", "hbm_GBs_moved") if kk in v}
                                                        for k, v in gt.items()}
            else:
                extras["gather_roofline"]["traffic"] = None
        except (OSError, ValueError):
            extras["gather_roofline"]["traffic"] = None
        # ---- the same training step through the real boundary: model(candidate_news, clicked_news) on CPU list-of-dicts ------------------
        # (single-GPU runs only: a training step is a collective operation, and the other ranks have left by now)
        if world == 1:
            cpu_batches = [wl.as_dataloader_batch(b) for b in wl.batches(rank, 2, B, 'cpu')]

            def step_dropin(i):
                loss = crit(wl.forward_dropin(model, cpu_batches[i % len(cpu_batches)]), target)
                loss.backward()
                opt.step()
            for i in range(4):
                step_dropin(i)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(args.steps):
                step_dropin(i)
            torch.cuda.synchronize()
            dtd = time.perf_counter() - ts
            extras["value_dropin"] = {"value": B * args.steps / dtd, "unit": "impressions/s", "ms_per_step": dtd / args.steps * 1e3,
                                      "what": "model(candidate_news, clicked_news) on the DataLoader's pinned CPU list-of-dicts "
                                              "(train.py:166-203): host stacking + id range check + H2D copy inside the timed step"}
        # ---- forward-only (scoring) throughput on train-shaped batches, and eval-shaped scoring (phases A-C) ------------------------------
        model.eval()
        with torch.no_grad():
            for i in range(3):
                wl.forward(model, batches[i % len(batches)])
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(10):
                wl.forward(model, batches[i % len(batches)])
            torch.cuda.synchronize()
            extras["score_impressions_per_s_fwd_only"] = 10 * B / (time.perf_counter() - ts)
        model.train()
        extras["score_eval"] = score_eval(wl, model, device)
        # ---- the other BASELINE workloads in the driver's own record: 10-step graph replays (north_star names NAML; configs[2], configs[4]) ----
        if world == 1 and args.model == 'NRMS' and shape == 'small' and not args.no_other_workloads:
            extras["other_workloads"] = {f"{m}_{sh}": other_workload(m, sh, 0, B, device) for m, sh in (('NAML', 'small'), ('LSTUR', 'large'))}

    wname = {'NRMS': "NRMS", 'NAML': "NAML (title+abstract+category+subcategory views, Conv1d k=3)",
             'LSTUR': "LSTUR (GRU user encoder 'ini' + per-user embedding row)"}[args.model]
    cfgidx = {('NRMS', 'small'): "configs[1]", ('NAML', 'small'): "configs[2]", ('NRMS', 'large'): "configs[3]", ('LSTUR', 'large'): "configs[4]"}
    tag = cfgidx.get((args.model, shape))
    if tag in ("configs[3]", "configs[4]") and world == 1:
        tag = f"single-GPU shard of {tag}"
    out = {
        "metric": f"impressions/sec ({args.model} training step: fwd+bwd+allreduce+Adam)", "value": world * B * args.steps / dt,
        "unit": "impressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{wname} bf16 on MI355X, MIND-{shape}-shaped synthetic, batch {B} per GPU"
                               + (f" (BASELINE.json {tag})" if tag else "") + (f", RCCL gradient exchange over xGMI, dp{world}" if world > 1 else ""),
                   "shape": shape, "per_gpu_batch": B, "global_batch": B * world, "news_per_impression": 53, "title_len": 20, "abstract_len": 50,
                   "num_clicked": 50, "d": 300, "heads": 15, "vocab": cfg.num_words, "num_news": cfg.num_news, "num_users": cfg.num_users,
                   "dropout": cfg.dropout_probability, "parallelism": f"dp{world}"},
        "roofline": roofline,
        "step_issue": ("one HIP graph per step (forward + backward + Adam), replayed; roofline durations from an eager pass of the same "
                       f"{args.steps} steps after the timed region" if sg is not None else
                       (("three HIP graphs per step ([forward + backward to the embedding scatter] | table exchange started | [weight-gradient GEMMs + row "
                         "staging] | rows + small bucket | [Adam]), replayed" if args.seg_overlap else
                         "two HIP graphs per step ([forward + backward] | RCCL collectives | [Adam]), replayed (--seg-overlap 0)") if seg is not None else
                        (seg_note or "kernel by kernel"))),
        "ms_per_step_eager": eager_ms,
        "loss": float(loss.item()),
        "grad_exchange": {"dense_allreduce_bytes": opt.dense_nbytes, "buckets": [[r.name, (r.hi - r.lo) * 4] for r in opt.regions],
                          "row_sparse_tables": [[s.name, list(s.param.shape), f"{B} (id, row) pairs per rank and step"] for s in opt.sparse],
                          "table_bucket_form": "reduce_scatter + sharded Adam + all_gather" if opt.table_rs else "all_reduce + Adam",
                          "bytes_per_step": dict(opt.comm_bytes)},
        "kernel_breakdown_us_per_step": {k: round(v[2] / NPROF, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])},
    }
    out.update(extras)
    if comm is not None:
        out["exposed_comm_ms"] = comm.pop("exposed_comm_ms")
        out["comm"] = comm
    if world == 1 and not args.no_parity:
        states = {"trained": model}
        m0 = wl.make_model().to(device)
        m0.load_state_dict(init_state)
        states["init"] = m0
        torch.manual_seed(5)
        pre = torch.randn(cfg.num_words, 300)                               # data_preprocess.py:272-277: N(0,1) rows incl. row 0
        states["pretrained_table"] = wl.make_model(seed=1, pretrained=pre).to(device)
        out["parity"] = parity_eval(wl, states, device)
        ps = parity_seeds(wl, states, device, seeds=args.parity_seeds)
        out["parity"]["seeds"] = ps["seeds"]
        out["parity"]["multi_seed_n1000"] = ps
        out["parity"]["within_tolerance"] = bool(out["parity"]["within_tolerance"] and ps["within_tolerance"])
        del states, m0
        # the other two model families of the hot path, initial weights, n = 1,000 impressions (BASELINE configs[0] size), in the SAME line
        # the driver records (the full three-state tables of all models: bench.py --model NAML / LSTUR)
        others = {}
        for other in ('NRMS', 'NAML', 'LSTUR'):
            if other == args.model:
                continue
            wo = Workload(other, make_cfg(other, shape, args.vocab))
            mo = wo.make_model(seed=0).to(device)
            pe = parity_eval(wo, {"init": mo}, device, n_news=2000, n_impr=1000)
            others[other] = {"abs_diff_auc_n1000": pe["worst_abs_diff_auc_n1000"], "abs_diff_ndcg10_n1000": pe["worst_abs_diff_ndcg10_n1000"],
                             "rms_logit_err": pe["states"]["init"]["rms_logit_err"], "logit_scale": pe["states"]["init"]["logit_scale"],
                             "within_tolerance": pe["within_tolerance"]}
            del mo, wo
        out["parity_models"] = others
        if not args.no_train_parity:
            # the engine, eight dropout streams, against the REAL reference's eight runs on the same task (committed fixture; no oracle training here)
            try:
                out["train_parity"] = train_parity_fixture(device, args.model, engine_seeds=8)
            except FileNotFoundError as e:
                out["train_parity"] = {"error": f"fixture missing: {e}"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, shape, args.vocab)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == '__main__':
    main()
