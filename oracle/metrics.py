"""Restatement of the reference's ranking metrics (TEST INFRASTRUCTURE).

Follows src/evaluate.py:24-42 (dcg/ndcg/mrr) and :160-168
(calculate_single_user_metric).  AUC restates sklearn.metrics.roc_auc_score for
binary labels via the rank-sum (Mann-Whitney) identity with average ranks for
ties, which is what sklearn's trapezoidal ROC area evaluates to.
"""
import numpy as np


def dcg_score(y_true, y_score, k=10):
    """src/evaluate.py:24-29."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    gains = 2.0 ** y_true - 1
    discounts = np.log2(np.arange(len(y_true)) + 2)
    return np.sum(gains / discounts)


def ndcg_score(y_true, y_score, k=10):
    """src/evaluate.py:32-35."""
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true, y_score):
    """src/evaluate.py:38-42."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    rr = y_true / (np.arange(len(y_true)) + 1)
    return np.sum(rr) / np.sum(y_true)


def auc_score(y_true, y_score):
    """Binary ROC-AUC (what roc_auc_score computes at src/evaluate.py:162)."""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    npos = int(y_true.sum())
    nneg = len(y_true) - npos
    if npos == 0 or nneg == 0:
        return np.nan
    order = np.argsort(y_score, kind='mergesort')
    s = y_score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty(len(s), dtype=np.float64)
    r[order] = ranks
    return (r[y_true == 1].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def single_impression_metrics(y_true, y_score):
    """src/evaluate.py:160-168 -> [auc, mrr, ndcg5, ndcg10].  Single-class impressions as the reference treats them on the
    scikit-learn it runs with here (1.7.2; tests/test_evaluate_fast.py holds this to the imported reference): roc_auc_score returns
    NaN (a warning, no ValueError), so an all-positive impression keeps MRR / nDCG and an all-negative one is 0 / 0 = NaN in all four
    (SURVEY.md 5.9 #11)."""
    y_true = np.asarray(y_true)
    if y_true.sum() == 0:
        return [np.nan] * 4
    return [auc_score(y_true, y_score), mrr_score(y_true, y_score),
            ndcg_score(y_true, y_score, 5), ndcg_score(y_true, y_score, 10)]


def evaluate_impressions(labels, scores):
    """nanmean over impressions, src/evaluate.py:270-272. labels/scores: lists of arrays."""
    res = np.array([single_impression_metrics(l, s) for l, s in zip(labels, scores)], dtype=np.float64)
    return tuple(np.nanmean(res, axis=0))
