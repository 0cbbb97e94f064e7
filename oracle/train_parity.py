"""Statistical training parity (TEST INFRASTRUCTURE; SURVEY.md section 7 step 5: "AUC after N steps with dropout on").

The reference trains with fp32 arithmetic and torch's dropout RNG (src/train.py:182-233); the engine trains with bf16 operands and its own
counter-based dropout RNG, so trained weights cannot agree bit for bit -- what must agree is the QUALITY of the trained model.  This
module holds the CPU half of that comparison: a learnable synthetic task (clicks sampled from a seeded teacher model), the oracle's
training loop (the reference's loop: CrossEntropyLoss on the positive-first candidate list, torch.optim.Adam), and the scoring of a held-out
eval-shaped impression set with the reference's metrics.  bench.py's `train_parity` leg and tests/test_training_parity_gpu.py run the
engine on the SAME batches from the SAME initial weights and compare AUC / nDCG@10 of the two trained models.

Only tests/, bench.py and __graft_entry__.smoke() may import this package (oracle/__init__.py)."""
import numpy as np
import torch

from oracle import metrics
from oracle.nrms_torch import OracleNRMS
from oracle.naml_torch import OracleNAML, random_naml_params
from oracle.lstur_torch import OracleLSTUR, random_lstur_params


def make_task(num_words=6000, n_news=2500, steps=200, B=16, n_eval=1000, seed=0, beta=3.0, neg_k=2, num_clicked=50, title_len=20):
    """A learnable NRMS task: a teacher (OracleNRMS, its own seed) scores every (history, candidate) pair; the clicked candidate of a
    training impression is sampled ~ softmax(beta * z) over its 1 + K candidates (z = teacher logits z-scored per impression) and put FIRST
    (data_preprocess.py:63-66); eval impressions get Bernoulli labels from the same teacher (synth.teacher_labels).
    Returns a dict of numpy arrays (token ids) + the teacher's state_dict."""
    from news_recommendation_amd import synth          # pure numpy generator (no device code): shared with bench.py
    rng = np.random.default_rng(seed)
    titles = synth.news_titles(rng, n_news, title_len, num_words)
    torch.manual_seed(10_000 + seed)
    teacher = OracleNRMS(num_words, 300, 15, 200, 0.0).eval()
    with torch.no_grad():
        nv = torch.cat([teacher.news_encoder(torch.from_numpy(titles[i:i + 512])) for i in range(0, n_news, 512)])
        nvp = torch.cat([nv, torch.zeros(1, 300)])

        def user_vecs(hist):
            idx = torch.from_numpy(np.where(hist < 0, n_news, hist))
            return torch.cat([teacher.user_encoder(nvp[idx[i:i + 256]]) for i in range(0, len(hist), 256)])
        n_train = steps * B
        cand = rng.integers(0, n_news, size=(n_train, 1 + neg_k))
        hl = synth.history_lengths(rng, n_train, num_clicked)
        hist = rng.integers(0, n_news, size=(n_train, num_clicked))
        hist[np.arange(num_clicked)[None, :] < (num_clicked - hl)[:, None]] = -1
        uv = user_vecs(hist)
        z = torch.einsum('bcd,bd->bc', nv[torch.from_numpy(cand)], uv).numpy().astype(np.float64)
        z = (z - z.mean(1, keepdims=True)) / (z.std(1, keepdims=True) + 1e-9)
        pr = np.exp(beta * z)
        pr /= pr.sum(1, keepdims=True)
        pick = (rng.random(n_train)[:, None] > np.cumsum(pr, axis=1)).sum(1).clip(max=neg_k)
        first = cand[np.arange(n_train), pick].copy()
        cand[np.arange(n_train), pick] = cand[:, 0]
        cand[:, 0] = first
        e_hist, e_cands, e_ptr = synth.eval_impressions(rng, n_news, n_eval, num_clicked)
        e_uv = user_vecs(e_hist)
        e_sc = np.concatenate([(nv[e_cands[e_ptr[i]:e_ptr[i + 1]]] @ e_uv[i]).numpy() for i in range(n_eval)])
    labels = synth.teacher_labels(np.random.default_rng(seed + 77), e_sc.astype(np.float64), e_ptr)
    cand_ids, click_ids = synth.batch_token_ids(titles, cand, hist)
    return {"titles": titles, "cand_ids": cand_ids.reshape(steps, B, 1 + neg_k, title_len), "click_ids": click_ids.reshape(steps, B, num_clicked, title_len),
            "cand": cand.reshape(steps, B, 1 + neg_k), "hist": hist.reshape(steps, B, num_clicked),
            "eval_hist": e_hist, "eval_cands": e_cands, "eval_ptr": e_ptr, "eval_labels": labels, "teacher_scores": e_sc,
            "num_words": num_words, "steps": steps, "B": B}


def init_state(num_words, seed=1):
    """Initial student weights (the reference's own initialisers: nn.Embedding N(0, 1) with a zero padding row, xavier / default Linear)."""
    torch.manual_seed(20_000 + seed)
    return {k: v.clone() for k, v in OracleNRMS(num_words, 300, 15, 200, 0.2).state_dict().items()}


def as_lists(ids):
    return [{'title': torch.from_numpy(np.ascontiguousarray(ids[:, j]))} for j in range(ids.shape[1])]


def train_oracle(task, state, lr=1e-3, p_drop=0.2, torch_seed=0, steps=None):
    """The reference's training loop on the CPU oracle (src/train.py:127-128,202-233): dropout on, torch.optim.Adam.  Returns the trained
    state_dict and the per-step losses."""
    # small batches of the 53-call per-position loop: with every hardware thread of a 128-thread host in torch's intra-op pool the
    # synchronisation costs 10 x the arithmetic (measured: 647 s on the GPU box's host vs 64 s on 8 cores) -- cap the pool for this loop
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(8, nthr))
    try:
        return _train_oracle(task, state, lr, p_drop, torch_seed, steps)
    finally:
        torch.set_num_threads(nthr)


def _train_oracle(task, state, lr, p_drop, torch_seed, steps):
    m = OracleNRMS(task["num_words"], 300, 15, 200, p_drop)
    m.load_state_dict(state)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    crit = torch.nn.CrossEntropyLoss()
    torch.manual_seed(torch_seed)
    B = task["B"]
    target = torch.zeros(B, dtype=torch.long)
    losses = []
    for i in range(task["steps"] if steps is None else steps):
        loss = crit(m(as_lists(task["cand_ids"][i]), as_lists(task["click_ids"][i])), target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}, losses


def oracle_eval_scores(task, state):
    m = OracleNRMS(task["num_words"], 300, 15, 200, 0.0)
    m.load_state_dict(state)
    m.eval()
    titles, hist, cands, ptr = task["titles"], task["eval_hist"], task["eval_cands"], task["eval_ptr"]
    n_news = titles.shape[0]
    with torch.no_grad():
        nv = torch.cat([m.news_encoder(torch.from_numpy(titles[i:i + 512])) for i in range(0, n_news, 512)])
        nvp = torch.cat([nv, torch.zeros(1, 300)])
        idx = torch.from_numpy(np.where(hist < 0, n_news, hist))
        uv = torch.cat([m.user_encoder(nvp[idx[i:i + 256]]) for i in range(0, len(hist), 256)])
        return np.concatenate([(nv[cands[ptr[i]:ptr[i + 1]]] @ uv[i]).numpy() for i in range(len(hist))])


def eval_metrics(task, scores):
    """(AUC, MRR, nDCG@5, nDCG@10) of `scores` on the task's held-out impressions (src/evaluate.py:160-168, 262-272)."""
    ptr, lab = task["eval_ptr"], task["eval_labels"]
    split = lambda a: [a[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]
    return metrics.evaluate_impressions(split(lab), split(np.asarray(scores)))


# ---- NAML leg (src/model/NAML: four views per news -- title, abstract, category, subcategory) -------------------------------------------------
NAML_ATTRS = ('title', 'abstract', 'category', 'subcategory')


def _naml(num_words, num_categories, p_drop):
    return OracleNAML(num_words, 300, num_categories, 100, 300, 3, 200, p_drop)


def _take(news, attr, idx):
    """news-index batch -> attribute array; padded history slots (-1) are all-zero news (dataset.py:44-60,79-83)."""
    a = news[attr]
    pad = np.zeros((1,) + a.shape[1:], dtype=np.int64)
    return np.concatenate([a, pad])[np.where(idx < 0, a.shape[0], idx)]


def _naml_vectors(model, news, hist):
    """news vectors of the whole table and user vectors of the histories (PADDED_NEWS = zero vector, evaluate.py:203)."""
    n_news = news['title'].shape[0]
    tn = {k: torch.from_numpy(v) for k, v in news.items()}
    nv = torch.cat([model.get_news_vector({k: v[i:i + 512] for k, v in tn.items()}) for i in range(0, n_news, 512)])
    nvp = torch.cat([nv, torch.zeros(1, nv.shape[1])])
    idx = torch.from_numpy(np.where(hist < 0, n_news, hist))
    uv = torch.cat([model.get_user_vector(nvp[idx[i:i + 256]]) for i in range(0, len(hist), 256)])
    return nv, uv


def make_task_naml(num_words=6000, num_categories=40, n_news=2000, steps=120, B=16, n_eval=800, seed=0, beta=3.0, neg_k=2, num_clicked=50):
    """The NAML counterpart of make_task: a seeded OracleNAML teacher labels training impressions (clicked candidate ~ softmax(beta z), put
    first) and eval impressions (Bernoulli).  Batches are kept as NEWS INDICES ([steps, B, 1 + K] and [steps, B, N], -1 = padded slot);
    naml_batch() turns one into the four attribute arrays."""
    from news_recommendation_amd import synth
    rng = np.random.default_rng(seed)
    news = {'title': synth.news_titles(rng, n_news, 20, num_words), 'abstract': synth.news_abstracts(rng, n_news, 50, num_words),
            'category': rng.integers(1, num_categories, size=n_news).astype(np.int64),
            'subcategory': rng.integers(1, num_categories, size=n_news).astype(np.int64)}
    teacher = _naml(num_words, num_categories, 0.0).eval()
    teacher.load_state_dict(random_naml_params(10_000 + seed, num_words, 300, num_categories, 100, 300, 3, 200, emb_std=0.5))
    n_train = steps * B
    cand = rng.integers(0, n_news, size=(n_train, 1 + neg_k))
    hl = synth.history_lengths(rng, n_train, num_clicked)
    hist = rng.integers(0, n_news, size=(n_train, num_clicked))
    hist[np.arange(num_clicked)[None, :] < (num_clicked - hl)[:, None]] = -1
    e_hist, e_cands, e_ptr = synth.eval_impressions(rng, n_news, n_eval, num_clicked)
    with torch.no_grad():
        nv, uv = _naml_vectors(teacher, news, np.concatenate([hist, e_hist]))
        z = torch.einsum('bcd,bd->bc', nv[torch.from_numpy(cand)], uv[:n_train]).numpy().astype(np.float64)
        e_uv = uv[n_train:]
        e_sc = np.concatenate([(nv[e_cands[e_ptr[i]:e_ptr[i + 1]]] @ e_uv[i]).numpy() for i in range(n_eval)])
    z = (z - z.mean(1, keepdims=True)) / (z.std(1, keepdims=True) + 1e-9)
    pr = np.exp(beta * z)
    pr /= pr.sum(1, keepdims=True)
    pick = (rng.random(n_train)[:, None] > np.cumsum(pr, axis=1)).sum(1).clip(max=neg_k)
    first = cand[np.arange(n_train), pick].copy()
    cand[np.arange(n_train), pick] = cand[:, 0]
    cand[:, 0] = first
    labels = synth.teacher_labels(np.random.default_rng(seed + 77), e_sc.astype(np.float64), e_ptr)
    return {"news": news, "cand": cand.reshape(steps, B, 1 + neg_k), "hist": hist.reshape(steps, B, num_clicked),
            "eval_hist": e_hist, "eval_cands": e_cands, "eval_ptr": e_ptr, "eval_labels": labels, "teacher_scores": e_sc,
            "num_words": num_words, "num_categories": num_categories, "steps": steps, "B": B}


def naml_batch(task, i):
    """(candidates, history) of training step i as {attr: int64 array [B, 1 + K or N, ...]}."""
    return ({k: _take(task["news"], k, task["cand"][i]) for k in NAML_ATTRS}, {k: _take(task["news"], k, task["hist"][i]) for k in NAML_ATTRS})


def init_state_naml(num_words, num_categories, seed=1):
    """Initial student weights: the seeded generator of the NAML parity tests (reference initialiser shapes; zero padding rows)."""
    return random_naml_params(20_000 + seed, num_words, 300, num_categories, 100, 300, 3, 200, emb_std=0.5)


def train_oracle_naml(task, state, lr=1e-3, p_drop=0.2, torch_seed=0):
    """The reference's training loop (src/train.py:127-128,202-233) on OracleNAML: dropout on, torch.optim.Adam."""
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(8, nthr))          # (see train_oracle)
    try:
        m = _naml(task["num_words"], task["num_categories"], p_drop)
        m.load_state_dict(state)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        crit = torch.nn.CrossEntropyLoss()
        torch.manual_seed(torch_seed)
        target = torch.zeros(task["B"], dtype=torch.long)
        lists = lambda d: [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in d.items()} for j in range(d['title'].shape[1])]
        losses = []
        for i in range(task["steps"]):
            cand, click = naml_batch(task, i)
            loss = crit(m(lists(cand), lists(click)), target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return {k: v.detach().clone() for k, v in m.state_dict().items()}, losses
    finally:
        torch.set_num_threads(nthr)


def oracle_eval_scores_naml(task, state):
    m = _naml(task["num_words"], task["num_categories"], 0.0)
    m.load_state_dict(state)
    m.eval()
    cands, ptr = task["eval_cands"], task["eval_ptr"]
    with torch.no_grad():
        nv, uv = _naml_vectors(m, task["news"], task["eval_hist"])
        return np.concatenate([(nv[cands[ptr[i]:ptr[i + 1]]] @ uv[i]).numpy() for i in range(len(task["eval_hist"]))])


# ---- LSTUR leg (src/model/LSTUR: title CNN + two category views, GRU over the click history initialised from a per-user row) -------------------
LSTUR_ATTRS = ('title', 'category', 'subcategory')


def _lstur(num_words, num_categories, num_users, p_drop, pm):
    return OracleLSTUR(num_words, 300, num_categories, num_users, 300, 3, 200, p_drop, pm, 'ini')


def _lstur_vectors(model, news, hist, users):
    """news vectors of the whole table and user vectors of the histories (PADDED_NEWS = zero vector; lengths = real history lengths,
    evaluate.py:203-233)."""
    n_news = news['title'].shape[0]
    tn = {k: torch.from_numpy(v) for k, v in news.items()}
    nv = torch.cat([model.get_news_vector({k: v[i:i + 512] for k, v in tn.items()}) for i in range(0, n_news, 512)])
    nvp = torch.cat([nv, torch.zeros(1, nv.shape[1])])
    idx = torch.from_numpy(np.where(hist < 0, n_news, hist))
    lengths = torch.from_numpy((hist >= 0).sum(1).astype(np.int64))
    ut = torch.from_numpy(users)
    uv = torch.cat([model.get_user_vector(ut[i:i + 256], lengths[i:i + 256].clone(), nvp[idx[i:i + 256]]) for i in range(0, len(hist), 256)])
    return nv, uv


def make_task_lstur(num_words=6000, num_categories=40, num_users=300, n_news=2000, steps=100, B=16, n_eval=800, seed=0, beta=3.0, neg_k=2,
                    num_clicked=50):
    """The LSTUR counterpart of make_task_naml: a seeded OracleLSTUR teacher labels training impressions (clicked candidate ~ softmax(beta z),
    put first) and eval impressions (Bernoulli); every impression has a user id (row 0 = the padding user is never drawn) and a history of
    at least one click.  Batches are news indices; lstur_batch() turns one into attribute arrays + user ids + lengths."""
    from news_recommendation_amd import synth
    rng = np.random.default_rng(seed)
    news = {'title': synth.news_titles(rng, n_news, 20, num_words),
            'category': rng.integers(1, num_categories, size=n_news).astype(np.int64),
            'subcategory': rng.integers(1, num_categories, size=n_news).astype(np.int64)}
    teacher = _lstur(num_words, num_categories, num_users, 0.0, 0.0).eval()
    teacher.load_state_dict(random_lstur_params(10_000 + seed, num_words, 300, num_categories, num_users, 300, 3, 200, 'ini', emb_std=0.5))
    n_train = steps * B
    cand = rng.integers(0, n_news, size=(n_train, 1 + neg_k))
    hl = np.maximum(synth.history_lengths(rng, n_train, num_clicked), 1)
    hist = rng.integers(0, n_news, size=(n_train, num_clicked))
    hist[np.arange(num_clicked)[None, :] < (num_clicked - hl)[:, None]] = -1
    users = rng.integers(1, num_users, size=n_train).astype(np.int64)
    e_hist, e_cands, e_ptr = synth.eval_impressions(rng, n_news, n_eval, num_clicked)
    for i in range(n_eval):                                  # at least one click per eval history too
        if (e_hist[i] >= 0).sum() == 0:
            e_hist[i, -1] = rng.integers(0, n_news)
    e_users = rng.integers(1, num_users, size=n_eval).astype(np.int64)
    with torch.no_grad():
        nv, uv = _lstur_vectors(teacher, news, np.concatenate([hist, e_hist]), np.concatenate([users, e_users]))
        z = torch.einsum('bcd,bd->bc', nv[torch.from_numpy(cand)], uv[:n_train]).numpy().astype(np.float64)
        e_uv = uv[n_train:]
        e_sc = np.concatenate([(nv[e_cands[e_ptr[i]:e_ptr[i + 1]]] @ e_uv[i]).numpy() for i in range(n_eval)])
    z = (z - z.mean(1, keepdims=True)) / (z.std(1, keepdims=True) + 1e-9)
    pr = np.exp(beta * z)
    pr /= pr.sum(1, keepdims=True)
    pick = (rng.random(n_train)[:, None] > np.cumsum(pr, axis=1)).sum(1).clip(max=neg_k)
    first = cand[np.arange(n_train), pick].copy()
    cand[np.arange(n_train), pick] = cand[:, 0]
    cand[:, 0] = first
    labels = synth.teacher_labels(np.random.default_rng(seed + 77), e_sc.astype(np.float64), e_ptr)
    return {"news": news, "cand": cand.reshape(steps, B, 1 + neg_k), "hist": hist.reshape(steps, B, num_clicked), "users": users.reshape(steps, B),
            "eval_hist": e_hist, "eval_cands": e_cands, "eval_ptr": e_ptr, "eval_users": e_users, "eval_labels": labels, "teacher_scores": e_sc,
            "num_words": num_words, "num_categories": num_categories, "num_users": num_users, "steps": steps, "B": B}


def lstur_batch(task, i):
    """(candidates, history, user ids, history lengths) of training step i."""
    cand = {k: _take(task["news"], k, task["cand"][i]) for k in LSTUR_ATTRS}
    click = {k: _take(task["news"], k, task["hist"][i]) for k in LSTUR_ATTRS}
    return cand, click, task["users"][i], (task["hist"][i] >= 0).sum(1).astype(np.int64)


def init_state_lstur(num_words, num_categories, num_users, seed=1):
    return random_lstur_params(20_000 + seed, num_words, 300, num_categories, num_users, 300, 3, 200, 'ini', emb_std=0.5)


def train_oracle_lstur(task, state, lr=1e-3, p_drop=0.2, pm=0.5, torch_seed=0):
    """The reference's training loop (src/train.py:127-128,183-233) on OracleLSTUR: dropout and whole-row user masking on, torch.optim.Adam."""
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(8, nthr))          # (see train_oracle)
    try:
        m = _lstur(task["num_words"], task["num_categories"], task["num_users"], p_drop, pm)
        m.load_state_dict(state)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        crit = torch.nn.CrossEntropyLoss()
        torch.manual_seed(torch_seed)
        target = torch.zeros(task["B"], dtype=torch.long)
        lists = lambda d: [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in d.items()} for j in range(d['title'].shape[1])]
        losses = []
        for i in range(task["steps"]):
            cand, click, user, length = lstur_batch(task, i)
            loss = crit(m(torch.from_numpy(user), torch.from_numpy(length), lists(cand), lists(click)), target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return {k: v.detach().clone() for k, v in m.state_dict().items()}, losses
    finally:
        torch.set_num_threads(nthr)


def oracle_eval_scores_lstur(task, state):
    m = _lstur(task["num_words"], task["num_categories"], task["num_users"], 0.0, 0.0)
    m.load_state_dict(state)
    m.eval()
    cands, ptr = task["eval_cands"], task["eval_ptr"]
    with torch.no_grad():
        nv, uv = _lstur_vectors(m, task["news"], task["eval_hist"], task["eval_users"])
        return np.concatenate([(nv[cands[ptr[i]:ptr[i + 1]]] @ uv[i]).numpy() for i in range(len(task["eval_hist"]))])


# ---- committed fixtures of the statistical legs (tests/golden/train_parity/*.npz, written by oracle/make_golden_train_parity.py) ------------------
# The task arrays (teacher-labelled batches as NEWS INDICES, the held-out eval set) + the results of the REAL reference trained on them.  The
# teacher and the reference runs only exist in the build container; the GPU test rebuilds the task from these arrays, never from the teacher.
_SCALARS = ("num_words", "num_categories", "num_users", "steps", "B")


def task_to_arrays(task):
    """Compact arrays of a make_task* result (ids as int32 / int16)."""
    out = {}
    news = task["news"] if "news" in task else {"title": task["titles"]}
    for k, v in news.items():
        out[f"news_{k}"] = v.astype(np.int16 if v.max() < 32000 else np.int32)
    for k in ("cand", "hist", "eval_hist"):
        out[k] = task[k].astype(np.int16 if task[k].max() < 32000 else np.int32)
    out["eval_cands"] = task["eval_cands"].astype(np.int32)
    out["eval_ptr"] = task["eval_ptr"].astype(np.int64)
    out["eval_labels"] = task["eval_labels"].astype(np.int8)
    out["teacher_scores"] = task["teacher_scores"].astype(np.float32)
    for k in ("users", "eval_users"):
        if k in task:
            out[k] = task[k].astype(np.int32)
    for k in _SCALARS:
        if k in task:
            out[f"scalar_{k}"] = np.array(task[k], dtype=np.int64)
    return out


def task_from_arrays(z):
    """Inverse of task_to_arrays (z: a loaded npz or a dict)."""
    from news_recommendation_amd import synth
    news = {k[5:]: z[k].astype(np.int64) for k in z.keys() if k.startswith("news_")}
    task = {k: z[k].astype(np.int64) for k in ("cand", "hist", "eval_hist", "eval_ptr", "eval_labels")}
    task["eval_cands"] = z["eval_cands"].astype(np.int32)
    task["teacher_scores"] = z["teacher_scores"].astype(np.float64)
    for k in ("users", "eval_users"):
        if k in z.keys():
            task[k] = z[k].astype(np.int64)
    for k in _SCALARS:
        if f"scalar_{k}" in z.keys():
            task[k] = int(z[f"scalar_{k}"])
    if set(news) == {"title"}:                     # the NRMS task carries token ids per step
        task["titles"] = news["title"]
        steps, B = task["steps"], task["B"]
        c, h = synth.batch_token_ids(news["title"], task["cand"].reshape(steps * B, -1), task["hist"].reshape(steps * B, -1))
        task["cand_ids"] = c.reshape(steps, B, -1, c.shape[-1])
        task["click_ids"] = h.reshape(steps, B, -1, h.shape[-1])
    else:
        task["news"] = news
    return task


def state_checksum(state):
    """A few float64 numbers that pin an initial state_dict across machines (sum, sum of squares, first element of every tensor)."""
    ks = sorted(state)
    return np.array([[float(state[k].double().sum()), float((state[k].double() ** 2).sum()), float(state[k].reshape(-1)[0])] for k in ks])
