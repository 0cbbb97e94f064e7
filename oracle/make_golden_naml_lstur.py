"""Golden vectors for NAML and LSTUR from the REAL reference (read-only import from /root/reference/src, CPU).
Only runnable in the build container; writes tests/golden/{naml,lstur}_*.npz (committed, travel to the GPU box).

    python oracle/make_golden_naml_lstur.py

Parameters are not stored: oracle.naml_torch.random_naml_params / oracle.lstur_torch.random_lstur_params regenerate
them from the seed, the reference gets them through load_state_dict (its own key names).
"""
import os
import sys
import warnings
import numpy as np

REF = os.environ.get('NR_REFERENCE_SRC', '/root/reference/src')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault('MODEL_NAME', 'NRMS')
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

import torch  # noqa: E402
from oracle.naml_torch import random_naml_params  # noqa: E402
from oracle.lstur_torch import random_lstur_params  # noqa: E402

NAML_CASES = {
    # name: dict(V, d, ncat, dcat, F, window, Q, B, C, N, L, La, seed)
    'tiny': dict(V=60, d=32, ncat=9, dcat=12, F=24, window=3, Q=16, B=3, C=3, N=5, L=6, La=9, seed=21),
    'base': dict(V=500, d=300, ncat=275, dcat=100, F=300, window=3, Q=200, B=2, C=3, N=50, L=20, La=50, seed=22),
}
LSTUR_CASES = {
    'tiny': dict(V=60, d=32, ncat=9, nusers=11, F=24, window=3, Q=16, B=4, C=3, N=5, L=6, method='ini', seed=31),
    'tinycon': dict(V=60, d=32, ncat=9, nusers=11, F=24, window=3, Q=16, B=4, C=3, N=5, L=6, method='con', seed=32),
    'base': dict(V=500, d=300, ncat=275, nusers=40, F=300, window=3, Q=200, B=4, C=3, N=50, L=20, method='ini', seed=33),
}


def text_ids(rng, n, L, V):
    ids = rng.integers(1, V, size=(n, L))
    lens = rng.integers(max(1, L // 4), L + 1, size=n)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    return ids.astype(np.int64)


def synth_news(rng, c, n, with_abstract):
    """n news items: title (+abstract) right-padded with 0, category / subcategory in [1, ncat)."""
    news = {'category': rng.integers(1, c['ncat'], size=n).astype(np.int64),
            'subcategory': rng.integers(1, c['ncat'], size=n).astype(np.int64),
            'title': text_ids(rng, n, c['L'], c['V'])}
    if with_abstract:
        news['abstract'] = text_ids(rng, n, c['La'], c['V'])
    return news


def synth_batch(rng, c, with_abstract):
    B, C, N = c['B'], c['C'], c['N']
    cand = synth_news(rng, c, B * C, with_abstract)
    click = synth_news(rng, c, B * N, with_abstract)
    hist = rng.integers(0, N + 1, size=B)
    hist[0] = N
    if B > 1:
        hist[1] = 0
    pad = (np.arange(N)[None, :] < (N - hist)[:, None]).reshape(-1)       # left padding (dataset.py:79-83): all-zero news
    for k in click:
        click[k][pad] = 0
    cand = {k: v.reshape(B, C, *v.shape[1:]) for k, v in cand.items()}
    click = {k: v.reshape(B, N, *v.shape[1:]) for k, v in click.items()}
    return cand, click, hist.astype(np.int64)


def as_lists(cand, click):
    C, N = cand['title'].shape[1], click['title'].shape[1]
    cl = [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in cand.items()} for j in range(C)]
    hl = [{k: torch.from_numpy(np.ascontiguousarray(v[:, j])) for k, v in click.items()} for j in range(N)]
    return cl, hl


def dump_grads(out, tag, model, small):
    for k, p in model.named_parameters():
        g = p.grad.detach().numpy()
        if small or g.size <= 4096:
            out[f'{tag}_grad/{k}'] = g
        else:
            out[f'{tag}_gradnorm/{k}'] = np.array(np.linalg.norm(g.astype(np.float64)))
            out[f'{tag}_gradslice/{k}'] = g.reshape(g.shape[0], -1)[:8, :16].copy()
            if 'embedding' in k:
                out[f'{tag}_gradrow0/{k}'] = g[0].copy()
                out[f'{tag}_gradrowsum/{k}'] = g.sum(axis=1)


def run_naml(name):
    from model.NAML import NAML                       # the reference's own model
    c = NAML_CASES[name]

    class Cfg:
        dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}
        num_words, word_embedding_dim = c['V'], c['d']
        num_categories, category_embedding_dim = c['ncat'], c['dcat']
        num_filters, window_size, query_vector_dim = c['F'], c['window'], c['Q']
        dropout_probability = 0.2
        num_clicked_news_a_user, num_words_title, num_words_abstract = c['N'], c['L'], c['La']
    rng = np.random.default_rng(c['seed'])
    params = random_naml_params(c['seed'], c['V'], c['d'], c['ncat'], c['dcat'], c['F'], c['window'], c['Q'])
    cand, click, _ = synth_batch(rng, c, True)
    out = {f'cand_{k}': v for k, v in cand.items()}
    out.update({f'click_{k}': v for k, v in click.items()})
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        model = NAML(Cfg)
        model.load_state_dict(params)
        model = model.to(dt).eval()
        cl, hl = as_lists(cand, click)
        logits = model(cl, hl)
        loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(c['B'], dtype=torch.long))
        loss.backward()
        flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
        nv = model.get_news_vector(flat)
        cv = torch.stack([model.get_news_vector(x) for x in hl], dim=1)
        uv = model.get_user_vector(cv)
        out[f'{tag}_logits'] = logits.detach().numpy()
        out[f'{tag}_loss'] = np.array(loss.item())
        out[f'{tag}_news_vec'] = nv.detach().numpy()
        out[f'{tag}_user_vec'] = uv.detach().numpy()
        out[f'{tag}_pred0'] = model.get_prediction(nv[:c['C']], uv[0]).detach().numpy()
        dump_grads(out, tag, model, name == 'tiny')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'naml_{name}.npz'), **out)
    print('naml', name, out['f32_logits'][0], out['f32_loss'])


def run_lstur(name):
    from model.LSTUR import LSTUR                     # the reference's own model
    c = LSTUR_CASES[name]

    class Cfg:
        dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}
        num_words, word_embedding_dim = c['V'], c['d']
        num_categories, num_users = c['ncat'], c['nusers']
        num_filters, window_size, query_vector_dim = c['F'], c['window'], c['Q']
        dropout_probability, masking_probability = 0.2, 0.5
        long_short_term_method = c['method']
        num_clicked_news_a_user, num_words_title = c['N'], c['L']
    rng = np.random.default_rng(c['seed'])
    params = random_lstur_params(c['seed'], c['V'], c['d'], c['ncat'], c['nusers'], c['F'], c['window'], c['Q'], c['method'])
    cand, click, hist = synth_batch(rng, c, False)
    user = rng.integers(0, c['nusers'], size=c['B']).astype(np.int64)
    out = {f'cand_{k}': v for k, v in cand.items()}
    out.update({f'click_{k}': v for k, v in click.items()})
    out['user'] = user
    out['clicked_news_length'] = hist
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        model = LSTUR(Cfg)
        model.load_state_dict(params)
        model = model.to(dt).eval()
        cl, hl = as_lists(cand, click)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            logits = model(torch.from_numpy(user), torch.from_numpy(hist.copy()), cl, hl)
        loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(c['B'], dtype=torch.long))
        loss.backward()
        flat = {k: torch.from_numpy(v.reshape(-1, *v.shape[2:])) for k, v in cand.items()}
        nv = model.get_news_vector(flat)
        cv = torch.stack([model.get_news_vector(x) for x in hl], dim=1)
        uv = model.get_user_vector(torch.from_numpy(user), torch.from_numpy(hist.copy()), cv)
        out[f'{tag}_logits'] = logits.detach().numpy()
        out[f'{tag}_loss'] = np.array(loss.item())
        out[f'{tag}_news_vec'] = nv.detach().numpy()
        out[f'{tag}_user_vec'] = uv.detach().numpy()
        dump_grads(out, tag, model, name.startswith('tiny'))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'lstur_{name}.npz'), **out)
    print('lstur', name, out['f32_logits'][0], out['f32_loss'])


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for n in NAML_CASES:
        run_naml(n)
    for n in LSTUR_CASES:
        run_lstur(n)
