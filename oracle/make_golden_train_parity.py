"""Fixtures of the STATISTICAL training-parity legs, from the REAL reference (read-only import from /root/reference/src, CPU).
Only runnable in the build container; writes tests/golden/train_parity/{nrms,naml,lstur}.npz (committed, travel to the GPU box).

    python oracle/make_golden_train_parity.py [--seeds 8] [--procs 3] [--models NRMS,NAML,LSTUR]

For each model: the teacher-labelled task of oracle/train_parity.py (batches as news indices + the held-out eval set), and the result of
training the reference's OWN model class on it -- its own forward, torch's dropout, torch.optim.Adam, the loop body of src/train.py:202-233 --
from the seeded initial state, once per torch seed: AUC / MRR / nDCG@5 / nDCG@10 on the held-out set (reference metric functions restated in
oracle/metrics.py, pinned by tests/golden/metrics.npz) and the mean of the last ten losses.  tests/test_training_parity_gpu.py trains the
ENGINE on the same arrays from the same initial state, as many seeds, and compares the two samples."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

REF = os.environ.get('NR_REFERENCE_SRC', '/root/reference/src')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault('MODEL_NAME', 'NRMS')
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

LR = 1e-3
SPEC = {'NRMS': dict(steps=200, B=16), 'NAML': dict(steps=100, B=16), 'LSTUR': dict(steps=100, B=16)}


def build(name):
    """(task, initial state) exactly as bench.train_parity* builds them."""
    from oracle import train_parity as tp
    s = SPEC[name]
    if name == 'NRMS':
        task = tp.make_task(steps=s['steps'], B=s['B'])
        return task, tp.init_state(task["num_words"])
    if name == 'NAML':
        task = tp.make_task_naml(steps=s['steps'], B=s['B'])
        return task, tp.init_state_naml(task["num_words"], task["num_categories"])
    task = tp.make_task_lstur(steps=s['steps'], B=s['B'])
    return task, tp.init_state_lstur(task["num_words"], task["num_categories"], task["num_users"])


def reference_model(name, task, p_drop, pm):
    """The reference's own model class on the task's sizes (src/model/<NAME>/__init__.py)."""
    class Cfg:
        num_words, word_embedding_dim = task["num_words"], 300
        num_attention_heads, query_vector_dim = 15, 200
        num_filters, window_size = 300, 3
        num_categories, category_embedding_dim = task.get("num_categories", 0), 100
        num_users = task.get("num_users", 0)
        dropout_probability, masking_probability = p_drop, pm
        long_short_term_method = 'ini'
        num_clicked_news_a_user, num_words_title, num_words_abstract = 50, 20, 50
    if name == 'NRMS':
        from model.NRMS import NRMS
        return NRMS(Cfg)
    if name == 'NAML':
        from model.NAML import NAML
        Cfg.dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}
        return NAML(Cfg)
    from model.LSTUR import LSTUR
    Cfg.dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}
    return LSTUR(Cfg)


def lists(d):
    import torch
    return [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in d.items()} for j in range(next(iter(d.values())).shape[1])]


def eval_reference(name, model, task):
    """Held-out scores of a reference model: every news once, one user vector per history, dot products (evaluate.py:185-260 in batched form;
    PADDED_NEWS = zero vector)."""
    import torch
    from oracle import train_parity as tp
    model.eval()
    cands, ptr = task["eval_cands"], task["eval_ptr"]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if name == 'LSTUR':
            nv, uv = tp._lstur_vectors(model, task["news"], task["eval_hist"], task["eval_users"])
        else:
            news = task["news"] if name == 'NAML' else {'title': task["titles"]}
            nv, uv = tp._naml_vectors(model, news, task["eval_hist"])
        sc = np.concatenate([(nv[cands[ptr[i]:ptr[i + 1]]] @ uv[i]).numpy() for i in range(len(task["eval_hist"]))])
    return tp.eval_metrics(task, sc)


def run(args):
    name, seed, threads = args
    import torch
    from oracle import train_parity as tp
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    task, st0 = build(name)
    if seed < 0:                                            # the initial model's metrics
        m = reference_model(name, task, 0.0, 0.0)
        m.load_state_dict(st0)
        return name, seed, [float(x) for x in eval_reference(name, m, task)], None, time.perf_counter() - t0
    m = reference_model(name, task, 0.2, 0.5)
    m.load_state_dict(st0)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=LR)           # train.py:127-128
    crit = torch.nn.CrossEntropyLoss()
    torch.manual_seed(seed)
    y = torch.zeros(task["B"], dtype=torch.long)            # train.py:205: the positive candidate is first
    losses = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i in range(task["steps"]):
            if name == 'NRMS':
                y_pred = m(tp.as_lists(task["cand_ids"][i]), tp.as_lists(task["click_ids"][i]))
            elif name == 'NAML':
                cand, click = tp.naml_batch(task, i)
                y_pred = m(lists(cand), lists(click))
            else:
                cand, click, user, length = tp.lstur_batch(task, i)
                y_pred = m(torch.from_numpy(user), torch.from_numpy(length), lists(cand), lists(click))
            loss = crit(y_pred, y)                          # train.py:202-233
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    met = eval_reference(name, m, task)
    return name, seed, [float(x) for x in met], float(np.mean(losses[-10:])), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--procs', type=int, default=3)
    ap.add_argument('--threads', type=int, default=2)
    ap.add_argument('--models', default='NRMS,NAML,LSTUR')
    a = ap.parse_args()
    models = a.models.split(',')
    jobs = [(n, s, a.threads) for n in models for s in range(-1, a.seeds)]
    with mp.get_context('spawn').Pool(a.procs) as pool:
        res = {}
        for name, seed, met, last10, sec in pool.imap_unordered(run, jobs):
            res.setdefault(name, {})[seed] = (met, last10)
            print(f'{name} seed {seed}: auc {met[0]:.4f} ndcg10 {met[3]:.4f} last10 {last10} ({sec:.0f} s)', flush=True)
    import torch
    from oracle import train_parity as tp
    out_dir = os.path.join(ROOT, 'tests', 'golden', 'train_parity')
    os.makedirs(out_dir, exist_ok=True)
    for name in models:
        task, st0 = build(name)
        arr = tp.task_to_arrays(task)
        seeds = sorted(s for s in res[name] if s >= 0)
        arr["ref_metrics"] = np.array([res[name][s][0] for s in seeds], dtype=np.float64)          # [seed][AUC, MRR, nDCG@5, nDCG@10]
        arr["ref_last10_loss"] = np.array([res[name][s][1] for s in seeds], dtype=np.float64)
        arr["ref_torch_seeds"] = np.array(seeds, dtype=np.int64)
        arr["init_metrics"] = np.array(res[name][-1][0], dtype=np.float64)
        arr["init_checksum"] = tp.state_checksum(st0)
        arr["meta"] = np.array(json.dumps({"model": name, "lr": LR, "dropout": 0.2, "masking_probability": 0.5, "torch": torch.__version__,
                                           "generator": "oracle/make_golden_train_parity.py", "reference": "src/model/%s, src/train.py:127-128,202-233" % name}))
        np.savez_compressed(os.path.join(out_dir, f'{name.lower()}.npz'), **arr)
        m = arr["ref_metrics"]
        print(name, 'reference AUC', np.round(m[:, 0], 4), 'mean', m[:, 0].mean(), 'sd', m[:, 0].std(ddof=1), 'init', arr["init_metrics"][0])


if __name__ == '__main__':
    main()
