#!/usr/bin/env python
"""Benchmark of the NRMS hot path on MI355X: training impressions/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of MIND-small-shaped synthetic impressions resident in HBM:
forward (embedding gather -> news encoder for 53 titles/impression -> user encoder -> dot-product scorer),
cross-entropy, backward, gradient all-reduce over RCCL (N > 1) and the Adam update -- everything
src/train.py:202-233 does per batch.  Workload = BASELINE.json configs[1]: NRMS, bf16 operands / fp32 accumulate,
batch 512 per GPU, title_len 20, 50 clicked news, d 300, vocabulary 70,976.  Weak scaling: per-GPU batch is fixed.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant hand-written kernel: algorithmic FLOPs per launch / HIP-event duration vs the dense
                bf16 MFMA peak (MI355X_MICROARCH.md: 2.5 PFLOP/s)
  gather_roofline  the embedding gather (north_star): algorithmic bytes / duration vs 8 TB/s HBM peak
  cpu_baseline  the oracle's torch-CPU port of the reference (oracle/nrms_torch.py) timed on this host
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

# algorithmic work per unit (SURVEY.md 8 d6 / DESIGN.md)
FLOP_PER_TITLE_MHSA = 2 * 20 * 300 * 900 + 2 * 2 * 15 * 20 * 20 * 20      # QKV 10.8 M + QK^T 0.24 M + PV 0.24 M
BYTES_PER_TOKEN_F32 = 1200                                                 # one fp32 embedding row


class Cfg:
    """The reference's NRMSConfig knobs (src/config.py:10-45) at MIND-small shape."""
    num_words = 1 + 70975
    word_embedding_dim = 300
    num_attention_heads = 15
    query_vector_dim = 200
    dropout_probability = 0.2
    num_clicked_news_a_user = 50
    num_words_title = 20
    negative_sampling_ratio = 2
    learning_rate = 0.0001


def make_model(seed=0):
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    torch.manual_seed(seed)
    return NRMS(Cfg)


def synth_batches(rank, n_batches, B, device):
    from news_recommendation_amd import synth
    rng = np.random.default_rng(1000 + rank)
    news = synth.news_titles(np.random.default_rng(0), 65238, Cfg.num_words_title, Cfg.num_words)
    out = []
    for _ in range(n_batches):
        cand, hist = synth.train_batch(rng, news, B, Cfg.num_clicked_news_a_user, Cfg.negative_sampling_ratio)
        c, h = synth.batch_token_ids(news, cand, hist)
        out.append((torch.from_numpy(c).to(device), torch.from_numpy(h).to(device)))
    return out


def cpu_baseline(seconds_budget=20.0, B=128):
    """The oracle's CPU PyTorch port of the reference NRMS (per-position encoder loop and all), one full train
    step (forward + backward + Adam) per iteration on a bounded sample."""
    from oracle.nrms_torch import OracleNRMS
    from news_recommendation_amd import synth
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # 128 threads on tiny per-title ops is slower than 32
    m = OracleNRMS(Cfg.num_words, 300, 15, 200, Cfg.dropout_probability).train()
    opt = torch.optim.Adam(m.parameters(), lr=Cfg.learning_rate)
    rng = np.random.default_rng(7)
    news = synth.news_titles(np.random.default_rng(0), 65238, 20, Cfg.num_words)
    cand, hist = synth.train_batch(rng, news, B)
    c, h = synth.batch_token_ids(news, cand, hist)
    cl = [{'title': torch.from_numpy(c[:, j])} for j in range(c.shape[1])]
    hl = [{'title': torch.from_numpy(h[:, j])} for j in range(h.shape[1])]
    crit = torch.nn.CrossEntropyLoss()

    def step():
        y = m(cl, hl)
        loss = crit(y, torch.zeros(B, dtype=torch.long))
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
            "sample": f"{n} train steps (fwd+bwd+Adam) of B={B} train-shaped impressions, oracle/nrms_torch.py on CPU fp32"}


def parity_eval(model, device, n_news=4000, n_impr=1000):
    """AUC / nDCG@10 of the engine vs the CPU oracle on the same synthetic eval-shaped impressions and weights."""
    from news_recommendation_amd import synth, ops
    from oracle.nrms_torch import OracleNRMS
    from oracle import metrics
    rng = np.random.default_rng(3)
    news = synth.news_titles(np.random.default_rng(2), n_news, 20, Cfg.num_words)
    hist, cands, ptr = synth.eval_impressions(rng, n_news, n_impr)
    ref = OracleNRMS(Cfg.num_words, 300, 15, 200, 0.2)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        t = torch.from_numpy(news)
        nv_ref = torch.cat([ref.get_news_vector({'title': t[i:i + 1024]}) for i in range(0, n_news, 1024)])
        nv_pad = torch.cat([nv_ref, torch.zeros(1, 300)])                      # PADDED_NEWS = zero vector (evaluate.py:203)
        hidx = torch.from_numpy(np.where(hist < 0, n_news, hist))
        uv_ref = torch.cat([ref.get_user_vector(nv_pad[hidx[i:i + 256]]) for i in range(0, n_impr, 256)])
        was_training = model.training
        model.eval()
        nv = torch.cat([model.get_news_vector({'title': t[i:i + 2048]}) for i in range(0, n_news, 2048)])
        nvp = torch.cat([nv, torch.zeros(1, 300, device=device)])
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
            "max_abs_logit_err": float(np.abs(sc - sc_ref).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch (impressions)')
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

    model = make_model().to(device).train()
    nrdist.broadcast_parameters(model)
    fgb = nrdist.FlatGradBuffer(model.parameters())
    try:
        opt = torch.optim.Adam(model.parameters(), lr=Cfg.learning_rate, fused=True)
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam(model.parameters(), lr=Cfg.learning_rate)
    crit = torch.nn.CrossEntropyLoss()
    batches = synth_batches(rank, 4, B, device)
    target = torch.zeros(B, dtype=torch.long, device=device)

    def step(i):
        cand, click = batches[i % len(batches)]
        y = model.forward_ids(cand, click)
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
    with ops.profile(only={dominant}) as rec2:
        for i in range(args.steps):
            loss = step(i)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    dom = rec2.summary().get(dominant, (0, float('nan'), 0.0))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    T = B * (1 + Cfg.negative_sampling_ratio + Cfg.num_clicked_news_a_user)
    # algorithmic work per launch of the kernels we know how to price
    flops = {
        'nr_mhsa_fwd[S=20]': T * FLOP_PER_TITLE_MHSA,
        'nr_mhsa_fwd[S=50]': B * (2 * 50 * 300 * 900 + 2 * 2 * 15 * 50 * 50 * 20),
        'nr_attn_bwd[S=20]': T * 15 * 6 * 2 * 20 * 20 * 20,       # 6 products of 20x20x20 per (title, head)
        'nr_attn_bwd[S=50]': B * 15 * 6 * 2 * 50 * 50 * 20,
        'nr_additive_fwd[S=20]': T * 2 * 20 * 300 * 200,
        'nr_additive_bwd[S=20]': T * 2 * 20 * 300 * 200,
    }
    roofline = None
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
        if tr is not None and B == 512:
            roofline["traffic"] = tr["bytes"]
            roofline["traffic_source"] = tr["source"]
    except (OSError, ValueError):
        pass

    # the embedding gather on its own (north_star: fraction of HBM roofline for the gather)
    lib = _capi.load()
    cand, click = batches[0]
    ids = torch.cat([cand.reshape(-1, 20), click.reshape(-1, 20)]).contiguous()
    gout = torch.empty(ids.numel(), 300, device=device)
    table = model.news_encoder.word_embedding.weight.detach()
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
              "algorithmic_bytes": g_bytes, "note": "reads only (1200 B row + 8 B id per token); the kernel also writes the same volume"}
    del gout

    # forward-only (scoring) throughput, same batches
    model.eval()
    with torch.no_grad():
        for i in range(3):
            model.forward_ids(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(10):
            model.forward_ids(*batches[i % len(batches)])
        torch.cuda.synchronize()
        fwd_ips = 10 * B / (time.perf_counter() - ts)
    model.train()

    out = {
        "metric": "impressions/sec (NRMS training step: fwd+bwd+allreduce+Adam)", "value": world * B * args.steps / dt,
        "unit": "impressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "NRMS bf16 on MI355X, MIND-small-shaped synthetic, batch 512 per GPU (BASELINE.json configs[1])",
                   "per_gpu_batch": B, "global_batch": B * world, "titles_per_impression": 53, "title_len": 20,
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
        out["parity"] = parity_eval(model, device)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
