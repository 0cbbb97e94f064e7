#!/usr/bin/env python
"""CPU baseline legs of bench.py, run in a child process that cannot see the GPUs.

    python tools/cpu_baseline.py --model NRMS --shape small --budget 24 [--reference /root/reference/src]

The reference hard-codes ``device = cuda:0 if available`` at import time (src/model/NRMS/news_encoder.py:7) and moves its INPUTS there
(:38), so it only runs on the host cores when the process sees no GPU: bench.py launches this script with HIP_VISIBLE_DEVICES="" /
CUDA_VISIBLE_DEVICES="".  With ``--reference`` (and the directory present) the timed model is the reference's own
``model.<NAME>.<NAME>`` and leg (iii) its own ``evaluate()``: kind "reference".  Without it (the GPU box has no /root/reference) the
oracle's torch restatement of the same modules is timed: kind "port".  Prints ONE JSON object.

Legs, each on a bounded sample (about a third of --budget seconds, at least one iteration after one warm-up):
  (i)   eval-mode ``model(candidate_news, clicked_news)`` on B = 128 train-shaped impressions            -> eval_forward
  (ii)  a full train step (forward, CrossEntropy, backward, Adam), B = 128, src/train.py:202-233         -> train_step (the headline)
  (iii) ``evaluate()`` phases A-C on a small synthetic data tree in the reference's file formats           -> evaluate
"""
import argparse
import json
import os
import sys
import tempfile
import time

os.environ['HIP_VISIBLE_DEVICES'] = ''
os.environ['CUDA_VISIBLE_DEVICES'] = ''
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

ATTRS = {'NRMS': ('title',), 'NAML': ('title', 'abstract', 'category', 'subcategory'), 'LSTUR': ('title', 'category', 'subcategory')}


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def timed(fn, budget, max_iter=64):
    fn()                                     # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        fn()
        n += 1
        if time.perf_counter() - t0 > budget or n >= max_iter:
            break
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='NRMS')
    ap.add_argument('--shape', default='small')
    ap.add_argument('--vocab', type=int, default=0)
    ap.add_argument('--budget', type=float, default=24.0)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--reference', default='')
    a = ap.parse_args()
    from news_recommendation_amd import synth
    shape = dict(synth.SHAPES[a.shape])
    if a.vocab:
        shape['num_words'] = a.vocab
    nproc = os.cpu_count() or 1
    threads = min(32, nproc)                 # more threads on the reference's tiny per-title ops is slower
    torch.set_num_threads(threads)
    use_ref = bool(a.reference) and os.path.isdir(a.reference)
    name = a.model
    if use_ref:
        os.environ['MODEL_NAME'] = name
        sys.dont_write_bytecode = True
        sys.path.insert(0, os.path.abspath(a.reference))
        import importlib
        from news_recommendation_amd.launcher import install_shims
        install_shims()
        config = getattr(importlib.import_module('config'), f'{name}Config')
        for k in ('num_words', 'num_users', 'num_categories'):
            setattr(config, k, shape[k])
        Model = getattr(importlib.import_module(f'model.{name}'), name)
        torch.manual_seed(0)
        model = Model(config)
    else:
        from news_recommendation_amd import default_config
        config = type('Cfg', (getattr(default_config, f'{name}Config'),), {k: shape[k] for k in ('num_words', 'num_users', 'num_categories')})
        torch.manual_seed(0)
        if name == 'NRMS':
            from oracle.nrms_torch import OracleNRMS
            model = OracleNRMS(config.num_words, 300, 15, 200, config.dropout_probability)
        elif name == 'NAML':
            from oracle.naml_torch import OracleNAML
            model = OracleNAML(config.num_words, 300, config.num_categories, 100, 300, 3, 200, config.dropout_probability)
        else:
            from oracle.lstur_torch import OracleLSTUR
            model = OracleLSTUR(config.num_words, 300, config.num_categories, config.num_users, 300, 3, 200, config.dropout_probability, 0.5, 'ini')

    # ---- train-shaped batch as the DataLoader delivers it (train.py:166-203): lists of per-position dicts of CPU tensors ----------
    B = a.batch
    rng = np.random.default_rng(7)
    n_news = min(shape['num_news'], 20000)
    news = {'title': synth.news_titles(rng, n_news, 20, shape['num_words']),
            'abstract': synth.news_abstracts(rng, n_news, 50, shape['num_words']),
            'category': rng.integers(1, shape['num_categories'], size=n_news).astype(np.int64),
            'subcategory': rng.integers(1, shape['num_categories'], size=n_news).astype(np.int64)}
    cand, hist = synth.train_batch(rng, news['title'], B)

    def take(attr, idx):
        arr = news[attr]
        pad = np.zeros((1,) + arr.shape[1:], dtype=np.int64)
        return torch.from_numpy(np.concatenate([arr, pad])[np.where(idx < 0, arr.shape[0], idx)])
    cl = [{k: take(k, cand[:, j]) for k in ATTRS[name]} for j in range(cand.shape[1])]
    hl = [{k: take(k, hist[:, j]) for k in ATTRS[name]} for j in range(hist.shape[1])]
    users = torch.from_numpy(rng.integers(1, shape['num_users'], size=B).astype(np.int64))
    lengths = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))

    def fwd():
        if name == 'LSTUR':
            return model(users, lengths.clone(), cl, hl)
        return model(cl, hl)

    out = {'kind': 'reference' if use_ref else 'port', 'cores': threads, 'threads': threads, 'nproc': nproc, 'cpu_model': cpu_model(),
           'unit': 'impressions/s', 'model': name, 'shape': a.shape, 'batch': B}
    leg = a.budget / 3.0
    # (i) eval forward
    model.eval()
    with torch.no_grad():
        n, dt = timed(fwd, leg)
    out['eval_forward'] = {'value': n * B / dt, 'iterations': n}
    # (ii) train step
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=config.learning_rate)
    crit = torch.nn.CrossEntropyLoss()
    y = torch.zeros(B, dtype=torch.long)

    def step():
        loss = crit(fwd(), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    n, dt = timed(step, leg, max_iter=16)
    out['train_step'] = {'value': n * B / dt, 'iterations': n}
    out['value'] = out['train_step']['value']
    # (iii) evaluate() on a small reference-format tree
    model.eval()
    n_val = 400
    with tempfile.TemporaryDirectory() as root:
        synth.write_reference_dataset(root, n_news=1500, n_users=200, n_train=8, n_val_impr=n_val, num_words=shape['num_words'], seed=3)
        cwd = os.getcwd()
        os.chdir(root)
        try:
            t0 = time.perf_counter()
            if use_ref:
                import evaluate as ref_eval
                metrics = ref_eval.evaluate(model, './data/val', config.num_workers)
            else:
                metrics = port_evaluate(model, './data/val', config, name)
            dt = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
    out['evaluate'] = {'value': n_val / dt, 'impressions': n_val, 'news': 1500, 'seconds': dt, 'auc': float(metrics[0]), 'ndcg10': float(metrics[3]),
                       'what': ("the reference's own evaluate() (src/evaluate.py:171-272): DataLoaders, batch-size-1 impression loop, metric Pool" if use_ref else
                                "oracle model in the reference's loop shape (evaluate.py:185-272): per-news / per-history dicts, one impression at a time with "
                                ".tolist(), metric Pool of num_workers; without its DataLoader workers")}
    out['sample'] = (f"{out['train_step']['iterations']} train steps (fwd+bwd+Adam) of B={B} train-shaped impressions; also "
                     f"{out['eval_forward']['iterations']} eval-mode forwards and evaluate() on {n_val} impressions / 1500 news; "
                     f"{'reference model imported from ' + a.reference if use_ref else 'oracle torch port of the reference'}, CPU fp32, "
                     f"{threads} torch threads on a host with {nproc} logical CPUs ({out['cpu_model']})")
    print(json.dumps(out))


def _single_metric(pair):
    from oracle import metrics
    return metrics.single_impression_metrics(np.asarray(pair[0]), np.asarray(pair[1]))


def port_evaluate(model, directory, config, name):
    """The reference's evaluate() in ITS loop shape (src/evaluate.py:185-272) on the oracle model: news vectors in chunks of batch_size * 16 into a
    dict keyed by news id (:185-203), user vectors per chunk of histories from per-news dict look-ups + torch.stack into a dict keyed by the
    history (:218-233), then ONE impression at a time (:235-260: the batch-size-1 loader): stack the candidates' vectors from the dict,
    get_prediction, .tolist(); the per-impression metrics in a multiprocessing Pool (:267-268).  What is not restated: the DataLoader worker
    processes and pandas row parsing (the files are parsed once by evaluate_fast.build_plan)."""
    from multiprocessing import Pool
    from news_recommendation_amd import evaluate_fast
    plan = evaluate_fast.build_plan(directory, config.dataset_attributes['news'], config.num_clicked_news_a_user)
    chunk = config.batch_size * 16
    with torch.no_grad():
        n = len(plan.news_ids)
        news2vector = {}
        for i in range(0, n, chunk):
            vec = model.get_news_vector({k: torch.from_numpy(v[i:i + chunk]) for k, v in plan.news.items()})
            for nid, v in zip(plan.news_ids[i:i + chunk], vec):
                if nid not in news2vector:
                    news2vector[nid] = v
        pad = torch.zeros(next(iter(news2vector.values())).size())
        ids = list(plan.news_ids)
        user2vector = {}
        nh = plan.hist_idx.shape[0]
        for i in range(0, nh, chunk):
            rows = plan.hist_idx[i:i + chunk]
            cv = torch.stack([torch.stack([news2vector[ids[j]] if 0 <= j < n else pad for j in row], dim=0) for row in rows], dim=0)
            if name == 'LSTUR':
                uv = model.get_user_vector(torch.from_numpy(plan.hist_user[i:i + chunk]), torch.from_numpy(plan.hist_len[i:i + chunk].copy()), cv)
            else:
                uv = model.get_user_vector(cv)
            for r, v in enumerate(uv):
                user2vector[i + r] = v
        tasks = []
        for i in range(len(plan.imp_user_row)):
            lo, hi = plan.cand_ptr[i], plan.cand_ptr[i + 1]
            cand = torch.stack([news2vector[ids[j]] if j >= 0 else pad for j in plan.cand_idx[lo:hi]], dim=0)
            y_pred = model.get_prediction(cand, user2vector[int(plan.imp_user_row[i])]).tolist()
            tasks.append((plan.labels[lo:hi].tolist(), y_pred))
    with Pool(processes=config.num_workers) as pool:
        res = np.array(pool.map(_single_metric, tasks), dtype=np.float64)
    return np.nanmean(res, axis=0)


if __name__ == '__main__':
    main()
