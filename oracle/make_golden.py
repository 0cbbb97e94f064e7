"""Generate golden vectors by running the REAL reference (read-only import from
/root/reference/src) on CPU.  Only runnable in the build container; the vectors
it writes to tests/golden/ are committed and travel to the GPU box.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

Parameters are NOT stored: they are regenerated from a seed with
oracle.nrms_numpy.random_nrms_params, loaded into the reference with
load_state_dict, and regenerated identically by the tests.
"""
import os
import sys
import numpy as np

REF = os.environ.get('NR_REFERENCE_SRC', '/root/reference/src')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault('MODEL_NAME', 'NRMS')
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

import torch  # noqa: E402
from oracle.nrms_numpy import random_nrms_params  # noqa: E402

CASES = {
    # name: (num_words, d, heads, qdim, B, C, N, L, seed)
    'tiny': (50, 60, 3, 16, 3, 3, 5, 6, 11),
    'base': (500, 300, 15, 200, 4, 3, 50, 20, 12),
}


def make_cfg(num_words, d, heads, qdim, N, L, p):
    class Cfg:
        pass
    Cfg.num_words = num_words
    Cfg.word_embedding_dim = d
    Cfg.num_attention_heads = heads
    Cfg.query_vector_dim = qdim
    Cfg.dropout_probability = p
    Cfg.num_clicked_news_a_user = N
    Cfg.num_words_title = L
    return Cfg


def synth_ids(rng, B, C, N, L, V):
    """MIND-shaped ids: right-padded titles, left-padded history (#5)."""
    def titles(n):
        ids = rng.integers(1, V, size=(n, L))
        lens = rng.integers(max(1, L // 4), L + 1, size=n)
        ids[np.arange(L)[None, :] >= lens[:, None]] = 0
        return ids
    cand = titles(B * C).reshape(B, C, L)
    click = titles(B * N).reshape(B, N, L)
    hist = rng.integers(0, N + 1, size=B)
    for b in range(B):
        click[b, :N - hist[b]] = 0            # left padding with all-zero titles
    return cand.astype(np.int64), click.astype(np.int64)


def run_case(name):
    from model.NRMS import NRMS                      # the reference's own model
    V, d, H, Q, B, C, N, L, seed = CASES[name]
    rng = np.random.default_rng(seed)
    params = random_nrms_params(rng, V, d, Q, np.float32, emb_std=0.5)
    cand, click = synth_ids(rng, B, C, N, L, V)
    out = dict(cand_ids=cand, click_ids=click)
    for dt_name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        model = NRMS(make_cfg(V, d, H, Q, N, L, 0.2))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        model = model.to(dt).eval()                  # eval: dropout off => deterministic
        cand_l = [{'title': torch.from_numpy(cand[:, j])} for j in range(C)]
        click_l = [{'title': torch.from_numpy(click[:, j])} for j in range(N)]
        logits = model(cand_l, click_l)
        loss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(B, dtype=torch.long))
        loss.backward()
        nv = model.get_news_vector({'title': torch.from_numpy(cand.reshape(-1, L))})
        cv = torch.stack([model.get_news_vector(x) for x in click_l], dim=1)
        uv = model.get_user_vector(cv)
        pr = model.get_prediction(nv[:C], uv[0])
        out[f'{dt_name}_logits'] = logits.detach().numpy()
        out[f'{dt_name}_loss'] = np.array(loss.item())
        out[f'{dt_name}_news_vec'] = nv.detach().numpy()
        out[f'{dt_name}_user_vec'] = uv.detach().numpy()
        out[f'{dt_name}_pred0'] = pr.detach().numpy()
        for k, p in model.named_parameters():
            g = p.grad.detach().numpy()
            if name == 'tiny' or g.size <= 4096:
                out[f'{dt_name}_grad/{k}'] = g
            else:                                    # big tensors: norm + a fixed slice
                out[f'{dt_name}_gradnorm/{k}'] = np.array(np.linalg.norm(g.astype(np.float64)))
                out[f'{dt_name}_gradslice/{k}'] = g[:8, :16].copy()
                if k.endswith('word_embedding.weight'):
                    out[f'{dt_name}_gradrow0/{k}'] = g[0].copy()
                    out[f'{dt_name}_gradrowsum/{k}'] = g.sum(axis=1)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'nrms_{name}.npz'), **out)
    print(name, 'logits', out['f32_logits'][0], 'loss', out['f32_loss'])


def run_metrics():
    import evaluate as ref_eval                      # the reference's evaluate.py
    rng = np.random.default_rng(5)
    ys, ss, res = [], [], []
    for i in range(40):
        n = int(rng.integers(2, 40))
        y = rng.integers(0, 2, size=n)
        if i % 7 == 0:
            y[:] = 0                                              # all-negative impression: 0 / 0 -> four NaNs
        elif i % 7 == 3:
            y[:] = 1                                              # all-positive: roc_auc_score is undefined; what the reference returns then depends
        else:                                                     # on its scikit-learn (1.7.2 here: NaN AUC + a warning, MRR / nDCG kept)
            y[0], y[-1] = 1, 0
        s = rng.normal(size=n).round(1 if i % 3 == 0 else 6)      # rounded => ties
        ys.append(y.astype(np.int64)); ss.append(s)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res.append(ref_eval.calculate_single_user_metric((y.tolist(), s.tolist())))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'metrics.npz'),
                        lens=np.array([len(y) for y in ys]), y=np.concatenate(ys),
                        s=np.concatenate(ss), res=np.array(res, dtype=np.float64))
    print('metrics', np.nanmean(np.array(res, dtype=np.float64), axis=0))


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for c in CASES:
        run_case(c)
    run_metrics()
