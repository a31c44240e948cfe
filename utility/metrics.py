"""Ranking metrics with the reference's exact definitions (reference utility/metrics.py).

All take ``r``: the 0/1 hit vector of a ranked list (position 0 = best). Note ndcg's ideal DCG is
computed from the retrieved hit vector itself, sorted (reference metrics.py:61-71), not from the
number of held-out items."""
import numpy as np


def _f64(r, k=None):
    a = np.asarray(r, dtype=np.float64)          # np.asfarray in the reference (removed in numpy 2)
    return a if k is None else a[:k]


def precision_at_k(r, k):
    assert k >= 1
    return np.mean(np.asarray(r)[:k])


def dcg_at_k(r, k, method=1):
    r = _f64(r, k)
    if not r.size:
        return 0.
    if method == 0:
        return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
    if method == 1:
        return np.sum(r / np.log2(np.arange(2, r.size + 2)))
    raise ValueError('method must be 0 or 1.')


def ndcg_at_k(r, k, method=1):
    best = dcg_at_k(sorted(r, reverse=True), k, method)
    return dcg_at_k(r, k, method) / best if best else 0.


def recall_at_k(r, k, all_pos_num):
    return 0 if all_pos_num == 0 else np.sum(_f64(r, k)) / all_pos_num


def hit_at_k(r, k):
    return 1. if np.sum(np.array(r)[:k]) > 0 else 0.


def recall(rank, ground_truth, N):
    return len(set(rank[:N]) & set(ground_truth)) / float(len(set(ground_truth)))


def average_precision(r, cut):
    r = np.asarray(r)
    out = [precision_at_k(r, k + 1) for k in range(cut) if r[k]]
    return np.sum(out) / float(min(cut, np.sum(r))) if out else 0.


def F1(pre, rec):
    return (2.0 * pre * rec) / (pre + rec) if pre + rec > 0 else 0.


def auc(ground_truth, prediction):
    try:
        from sklearn.metrics import roc_auc_score
        return roc_auc_score(y_true=ground_truth, y_score=prediction)
    except Exception:
        return 0.


def metrics_from_hit_matrix(hits, n_pos, Ks, list_len=None):
    """Vectorised over users: ``hits`` [n, Kmax] 0/1, ``n_pos`` [n] held-out counts,
    ``list_len`` [n] number of ranked items actually available (< Kmax only when a user has fewer
    than Kmax candidate items; the reference's precision is then a mean over the shorter list).
    Returns dict of per-user arrays [n, len(Ks)] for precision / recall / ndcg / hit_ratio,
    equal to calling the scalar functions above per user (reference batch_test.py:70-80)."""
    hits = np.asarray(hits, dtype=np.float64)
    n_pos = np.asarray(n_pos, dtype=np.float64)
    n, kmax = hits.shape
    list_len = np.full(n, kmax, dtype=np.float64) if list_len is None else np.asarray(list_len, dtype=np.float64)
    disc = 1.0 / np.log2(np.arange(2, kmax + 2))
    ideal = -np.sort(-hits, axis=1)                    # sorted(r, reverse=True)
    out = {k: np.zeros((n, len(Ks))) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for j, K in enumerate(Ks):
        K = min(K, kmax)
        h = hits[:, :K]
        s = h.sum(1)
        out["precision"][:, j] = s / np.maximum(np.minimum(list_len, K), 1)      # np.mean(r[:K]) over the real list
        out["recall"][:, j] = np.where(n_pos > 0, s / np.maximum(n_pos, 1), 0.0)
        dcg = (h * disc[:K]).sum(1)
        idcg = (ideal[:, :K] * disc[:K]).sum(1)
        out["ndcg"][:, j] = np.where(idcg > 0, dcg / np.where(idcg > 0, idcg, 1.0), 0.0)
        out["hit_ratio"][:, j] = (s > 0).astype(np.float64)
    return out
