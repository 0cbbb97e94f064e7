#!/usr/bin/env python
"""Benchmark of the news-recommendation hot path on MI355X: training impressions/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W [--model NRMS|NAML|LSTUR]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of MIND-small-shaped synthetic impressions resident in HBM:
forward (embedding gather -> news encoder for 53 news/impression -> user encoder -> dot-product scorer),
cross-entropy, backward, gradient all-reduce over RCCL (N > 1) and the Adam update -- everything
src/train.py:182-233 does per batch.  Default workload = BASELINE.json configs[1]: NRMS, bf16 operands / fp32 accumulate,
batch 512 per GPU, title_len 20, 50 clicked news, d 300, vocabulary 70,976.  --model NAML is configs[2] (title + abstract +
category + subcategory views), --model LSTUR the single-GPU shard of configs[4].  Weak scaling: per-GPU batch is fixed.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant hand-written kernel: algorithmic FLOPs per launch / HIP-event duration vs the dense
                bf16 MFMA peak (MI355X_MICROARCH.md: 2.5 PFLOP/s); traffic = HBM bytes per launch from the committed PMC passes
  gather_roofline  the embedding gather (north_star): algorithmic bytes / duration vs 8 TB/s HBM peak
  cpu_baseline  the oracle's torch-CPU port of the reference timed on this host (bounded sample)
  parity        AUC / nDCG@10 of engine vs oracle on synthetic eval impressions (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s
BYTES_PER_TOKEN_F32 = 1200   # one fp32 embedding row
N_NEWS = 65238               # MIND-small news count


class Cfg:
    """The reference's BaseConfig / NRMSConfig / NAMLConfig / LSTURConfig knobs (src/config.py:10-69) at MIND-small shape."""
    num_words = 1 + 70975
    num_categories = 1 + 274
    num_users = 1 + 50000
    word_embedding_dim = 300
    category_embedding_dim = 100
    num_attention_heads = 15
    query_vector_dim = 200
    dropout_probability = 0.2
    num_clicked_news_a_user = 50
    num_words_title = 20
    num_words_abstract = 50
    negative_sampling_ratio = 2
    learning_rate = 0.0001
    num_filters = 300
    window_size = 3
    long_short_term_method = 'ini'
    masking_probability = 0.5


class NamlCfg(Cfg):
    dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}


class LsturCfg(Cfg):
    dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}


# ----------------------------------------------------------------------------------------------------------------------
# synthetic news table + batches (ids resident in HBM)
# ----------------------------------------------------------------------------------------------------------------------
def news_table(seed, n_news):
    from news_recommendation_amd import synth
    rng = np.random.default_rng(seed)
    return {'title': synth.news_titles(rng, n_news, Cfg.num_words_title, Cfg.num_words),
            'abstract': synth.news_abstracts(rng, n_news, Cfg.num_words_abstract, Cfg.num_words),
            'category': rng.integers(1, Cfg.num_categories, size=n_news).astype(np.int64),
            'subcategory': rng.integers(1, Cfg.num_categories, size=n_news).astype(np.int64)}


def take(news, attr, idx):
    """news-index batch -> attribute tensor; padded history slots (-1) are all-zero news (dataset.py:44-60,79-83)."""
    a = news[attr]
    pad = np.zeros((1,) + a.shape[1:], dtype=np.int64)
    return np.concatenate([a, pad])[np.where(idx < 0, a.shape[0], idx)]


class Workload:
    def __init__(self, name):
        self.name = name
        self.attrs = {'NRMS': ('title',), 'NAML': ('title', 'abstract', 'category', 'subcategory'),
                      'LSTUR': ('title', 'category', 'subcategory')}[name]

    def make_model(self, seed=0):
        torch.manual_seed(seed)
        if self.name == 'NRMS':
            from news_recommendation_amd.dropin.model.NRMS import NRMS
            return NRMS(Cfg)
        if self.name == 'NAML':
            from news_recommendation_amd.dropin.model.NAML import NAML
            return NAML(NamlCfg)
        from news_recommendation_amd.dropin.model.LSTUR import LSTUR
        return LSTUR(LsturCfg)

    def make_oracle(self):
        if self.name == 'NRMS':
            from oracle.nrms_torch import OracleNRMS
            return OracleNRMS(Cfg.num_words, 300, 15, 200, Cfg.dropout_probability)
        if self.name == 'NAML':
            from oracle.naml_torch import OracleNAML
            return OracleNAML(Cfg.num_words, 300, Cfg.num_categories, 100, 300, 3, 200, Cfg.dropout_probability)
        from oracle.lstur_torch import OracleLSTUR
        return OracleLSTUR(Cfg.num_words, 300, Cfg.num_categories, Cfg.num_users, 300, 3, 200, Cfg.dropout_probability, 0.5, 'ini')

    def batches(self, rank, n_batches, B, device):
        from news_recommendation_amd import synth
        news = news_table(0, N_NEWS)
        rng = np.random.default_rng(1000 + rank)
        out = []
        for _ in range(n_batches):
            cand, hist = synth.train_batch(rng, news['title'], B, Cfg.num_clicked_news_a_user, Cfg.negative_sampling_ratio)
            b = {'cand': {k: torch.from_numpy(take(news, k, cand)).to(device) for k in self.attrs},
                 'click': {k: torch.from_numpy(take(news, k, hist)).to(device) for k in self.attrs}}
            if self.name == 'LSTUR':
                b['user'] = torch.from_numpy(rng.integers(1, Cfg.num_users, size=B).astype(np.int64)).to(device)
                b['length'] = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))          # CPU, as the reference requires
            out.append(b)
        return out

    def forward(self, model, b):
        if self.name == 'NRMS':
            return model.forward_ids(b['cand']['title'], b['click']['title'])
        if self.name == 'NAML':
            return model.forward_ids(b['cand'], b['click'])
        return model.forward_ids(b['user'], b['length'].clone(), b['cand'], b['click'])

    def oracle_forward(self, ref, b):
        C, N = b['cand']['title'].shape[1], b['click']['title'].shape[1]
        cl = [{k: b['cand'][k][:, j] for k in self.attrs} for j in range(C)]
        hl = [{k: b['click'][k][:, j] for k in self.attrs} for j in range(N)]
        if self.name == 'LSTUR':
            return ref(b['user'], b['length'].clone(), cl, hl)
        return ref(cl, hl)

    def flops(self, B):
        """Algorithmic FLOPs per launch of the kernels we know how to price (SURVEY 8 d6 / DESIGN.md)."""
        T = B * 53
        conv = lambda S: T * 2 * S * 900 * 300
        return {
            'nr_mhsa_fwd[S=20]': T * (2 * 20 * 300 * 900 + 2 * 2 * 15 * 20 * 20 * 20),
            'nr_mhsa_fwd[S=50]': B * (2 * 50 * 300 * 900 + 2 * 2 * 15 * 50 * 50 * 20),
            'nr_attn_bwd[S=20]': T * 15 * 6 * 2 * 20 * 20 * 20,
            'nr_attn_bwd[S=50]': B * 15 * 6 * 2 * 50 * 50 * 20,
            'nr_additive_fwd[S=20]': T * 2 * 20 * 300 * 200, 'nr_additive_bwd[S=20]': T * 2 * 20 * 300 * 200,
            'nr_additive_fwd[title]': T * 2 * 20 * 300 * 200, 'nr_additive_bwd[title]': T * 2 * 20 * 300 * 200,
            'nr_additive_fwd[abstract]': T * 2 * 50 * 300 * 200, 'nr_additive_bwd[abstract]': T * 2 * 50 * 300 * 200,
            'nr_conv3_fwd[title]': conv(20), 'nr_conv3_dgrad[title]': conv(20),
            'nr_conv3_fwd[abstract]': conv(50), 'nr_conv3_dgrad[abstract]': conv(50),
            'nr_gru_fwd_step': 2 * B * 900 * 2700, 'nr_gru_bwd_step': 2 * B * 2700 * 900,
        }


def cpu_baseline(wl, seconds_budget=20.0, B=128):
    """The oracle's CPU PyTorch port of the reference (per-position encoder loop and all), full train steps
    (forward + backward + Adam) on a bounded sample."""
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # 128 threads on tiny per-title ops is slower than 32
    m = wl.make_oracle().train()
    opt = torch.optim.Adam(m.parameters(), lr=Cfg.learning_rate)
    b = wl.batches(7, 1, B, 'cpu')[0]
    crit = torch.nn.CrossEntropyLoss()

    def step():
        loss = crit(wl.oracle_forward(m, b), torch.zeros(B, dtype=torch.long))
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "impressions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps (fwd+bwd+Adam) of B={B} train-shaped impressions, oracle {wl.name} torch port on CPU fp32"}


def parity_eval(wl, model, device, n_news=4000, n_impr=5000):
    """AUC / nDCG@10 of the engine vs the CPU oracle on the same synthetic eval-shaped impressions and weights
    (phases A-C of src/evaluate.py:185-260: news vectors, user vectors with a zero PADDED_NEWS vector, per-impression dot products).
    5,000 impressions: the bf16-operand logit noise (rms ~1e-3 of the logit scale) flips near-tied candidate pairs at random, and on
    1,000 impressions of a barely trained model that alone moves AUC by 1e-5 .. 1.2e-3 from one weight state to the next."""
    from news_recommendation_amd import synth, ops
    from oracle import metrics
    rng = np.random.default_rng(3)
    news = {k: v for k, v in news_table(2, n_news).items() if k in wl.attrs}
    hist, cands, ptr = synth.eval_impressions(rng, n_news, n_impr)
    users = torch.from_numpy(rng.integers(1, Cfg.num_users, size=n_impr).astype(np.int64))
    lengths = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))
    ref = wl.make_oracle()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    ref.eval()
    tn = {k: torch.from_numpy(v) for k, v in news.items()}
    sl = lambda i, n: {k: v[i:i + n] for k, v in tn.items()}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        nv_ref = torch.cat([ref.get_news_vector(sl(i, 1024)) for i in range(0, n_news, 1024)])
        D = nv_ref.shape[1]
        nv_pad = torch.cat([nv_ref, torch.zeros(1, D)])                        # PADDED_NEWS = zero vector (evaluate.py:203)
        hidx = torch.from_numpy(np.where(hist < 0, n_news, hist))
        nv = torch.cat([model.get_news_vector(sl(i, 2048)) for i in range(0, n_news, 2048)])
        nvp = torch.cat([nv, torch.zeros(1, D, device=device)])
        if wl.name == 'LSTUR':
            uv_ref = torch.cat([ref.get_user_vector(users[i:i + 256], lengths[i:i + 256].clone(), nv_pad[hidx[i:i + 256]]) for i in range(0, n_impr, 256)])
            uv = model.get_user_vector(users, lengths.clone(), nvp[hidx.to(device)])
        else:
            uv_ref = torch.cat([ref.get_user_vector(nv_pad[hidx[i:i + 256]]) for i in range(0, n_impr, 256)])
            uv = model.get_user_vector(nvp[hidx.to(device)])
        sc = ops.score_csr(nv, uv, torch.from_numpy(cands).to(device), torch.from_numpy(ptr).to(device),
                           torch.arange(n_impr, dtype=torch.int32, device=device)).cpu().numpy()
    model.train(was_training)
    sc_ref = np.concatenate([(nv_ref[cands[ptr[i]:ptr[i + 1]]] @ uv_ref[i]).numpy() for i in range(n_impr)])
    labels = synth.teacher_labels(np.random.default_rng(4), sc_ref.astype(np.float64), ptr)
    split = lambda a: [a[ptr[i]:ptr[i + 1]] for i in range(n_impr)]
    auc_r, _, _, nd_r = metrics.evaluate_impressions(split(labels), split(sc_ref))
    auc_e, _, _, nd_e = metrics.evaluate_impressions(split(labels), split(sc))
    return {"n_impressions": n_impr, "auc_oracle": auc_r, "auc_engine": auc_e, "ndcg10_oracle": nd_r, "ndcg10_engine": nd_e,
            "abs_diff_auc": abs(auc_r - auc_e), "abs_diff_ndcg10": abs(nd_r - nd_e), "tolerance": 1e-3,
            "max_abs_logit_err": float(np.abs(sc - sc_ref).max()), "rms_logit_err": float(np.sqrt(np.mean((sc - sc_ref) ** 2))),
            "mean_logit_err": float(np.mean(sc - sc_ref)), "logit_scale": float(np.abs(sc_ref).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch (impressions)')
    ap.add_argument('--model', default='NRMS', choices=['NRMS', 'NAML', 'LSTUR'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    args = ap.parse_args()

    import torch.distributed as dist
    from news_recommendation_amd import dist as nrdist, ops, _capi
    rank, world, local = nrdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    B = args.batch
    wl = Workload(args.model)

    model = wl.make_model().to(device).train()
    nrdist.broadcast_parameters(model)
    fgb = nrdist.FlatGradBuffer(model.parameters())
    try:
        opt = torch.optim.Adam(model.parameters(), lr=Cfg.learning_rate, fused=True)
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam(model.parameters(), lr=Cfg.learning_rate)
    crit = torch.nn.CrossEntropyLoss()
    batches = wl.batches(rank, 4, B, device)
    target = torch.zeros(B, dtype=torch.long, device=device)

    def step(i):
        y = wl.forward(model, batches[i % len(batches)])
        loss = crit(y, target)
        fgb.zero()
        loss.backward()
        fgb.allreduce_mean()
        opt.step()
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
    assert fgb.check_views(), "gradient views detached from the flat buffer"
    hand = {k: v for k, v in prof.items() if k.startswith('nr_')}
    dominant = max(hand, key=lambda k: hand[k][2]) if hand else 'nr_mhsa_fwd[S=20]'

    barrier()
    t0 = time.perf_counter()
    # the GRU steps are issued as one C call per recurrence in the timed region (per-step launches from Python make the LSTUR step
    # host-bound): the event pair then brackets T (+1) launches and the per-launch average is total / launches
    SEQ = {'nr_gru_fwd_step': 'nr_gru_fwd_seq', 'nr_gru_bwd_step': 'nr_gru_bwd_seq'}
    timed_name = SEQ.get(dominant, dominant)
    with ops.profile(only={timed_name}) as rec2:
        for i in range(args.steps):
            loss = step(i)
    t_enq = time.perf_counter() - t0       # host time to ENQUEUE the timed steps (launch-bound if it approaches dt)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    dom = rec2.summary().get(timed_name, (0, float('nan'), 0.0))
    if timed_name != dominant:
        per_call = ops.seq_launches.get(timed_name, 1)
        dom = (dom[0] * per_call, dom[1] / per_call, dom[2])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    T = B * (1 + Cfg.negative_sampling_ratio + Cfg.num_clicked_news_a_user)
    flops = wl.flops(B)
    if dominant in flops:
        ach = flops[dominant] / (dom[1] * 1e-6) / 1e12
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / MFMA_BF16_PEAK_TF, "traffic": None, "avg_us": dom[1], "launches": dom[0],
                    "flop_per_launch": flops[dominant]}
    else:                                   # HBM-bound helper kernel dominant (gather / scatter): price by bytes
        nbytes = T * 20 * BYTES_PER_TOKEN_F32
        ach = nbytes / (dom[1] * 1e-6) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_us": dom[1], "launches": dom[0], "bytes_per_launch": nbytes}
    # HBM traffic of that kernel: PMC counters cannot be read from inside the process, so the per-launch figure comes from
    # the committed rocprofv3 --pmc passes of the same workload (profiles/traffic.json, made by tools/gpu_check.sh)
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            tr = json.load(f).get(dominant)
        if tr is not None and B == 512 and args.model == 'NRMS':
            roofline["traffic"] = tr["bytes"]
            roofline["traffic_source"] = tr["source"]
    except (OSError, ValueError):
        pass

    # the embedding gather on its own (north_star: fraction of HBM roofline for the gather)
    lib = _capi.load()
    b0 = batches[0]
    ids = torch.cat([b0['cand']['title'].reshape(-1, 20), b0['click']['title'].reshape(-1, 20)]).contiguous()
    gout = torch.empty(ids.numel(), 300, device=device)
    table = next(p for n, p in model.named_parameters() if n.endswith('word_embedding.weight')).detach()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), gout.data_ptr(), ids.numel(), 300, table.shape[0], st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), gout.data_ptr(), ids.numel(), 300, table.shape[0], st)
    e1.record()
    torch.cuda.synchronize()
    g_us = e0.elapsed_time(e1) * 1e3 / 10
    g_bytes = ids.numel() * (BYTES_PER_TOKEN_F32 + 8)
    gather = {"kernel": "nr_gather_rows_f32", "bound": "hbm", "achieved": g_bytes / (g_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": g_bytes / (g_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": g_us,
              "algorithmic_bytes": g_bytes, "read_plus_write_GBs": (g_bytes + ids.numel() * BYTES_PER_TOKEN_F32) / (g_us * 1e-6) / 1e9,
              "frac_read_plus_write": (g_bytes + ids.numel() * BYTES_PER_TOKEN_F32) / (g_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
              "note": "achieved / frac count the algorithmic READ bytes only (1200 B row + 8 B id per token; the 85 MB table is Infinity-Cache "
                      "resident); the stand-alone kernel also writes the gathered rows (same volume, to HBM): read_plus_write_GBs"}
    del gout

    # forward-only (scoring) throughput, same batches
    model.eval()
    with torch.no_grad():
        for i in range(3):
            wl.forward(model, batches[i % len(batches)])
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(10):
            wl.forward(model, batches[i % len(batches)])
        torch.cuda.synchronize()
        fwd_ips = 10 * B / (time.perf_counter() - ts)
    model.train()

    cfg_names = {'NRMS': "NRMS bf16 on MI355X, MIND-small-shaped synthetic, batch 512 per GPU (BASELINE.json configs[1])",
                 'NAML': "NAML (title+abstract+category+subcategory views, Conv1d k=3) bf16 on MI355X, MIND-small-shaped synthetic, batch 512 per GPU (BASELINE.json configs[2])",
                 'LSTUR': "LSTUR (GRU user encoder 'ini' + per-user embedding row) bf16 on MI355X, MIND-small-shaped synthetic, batch 512 per GPU; single-GPU shard of BASELINE.json configs[4]"}
    out = {
        "metric": f"impressions/sec ({args.model} training step: fwd+bwd+allreduce+Adam)", "value": world * B * args.steps / dt,
        "unit": "impressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": cfg_names[args.model].replace('batch 512', f'batch {B}'),
                   "per_gpu_batch": B, "global_batch": B * world, "news_per_impression": 53, "title_len": 20, "abstract_len": 50,
                   "num_clicked": 50, "d": 300, "heads": 15, "vocab": Cfg.num_words, "dropout": Cfg.dropout_probability,
                   "parallelism": f"dp{world}"},
        "roofline": roofline,
        "gather_roofline": gather,
        "score_impressions_per_s_fwd_only": fwd_ips,
        "loss": float(loss.item()),
        "grad_allreduce_bytes": fgb.nbytes,
        "kernel_breakdown_us_per_step": {k: round(v[2] / NPROF, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])},
    }
    if world == 1 and not args.no_parity:
        out["parity"] = parity_eval(wl, model, device)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
