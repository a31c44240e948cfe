"""Full-rank evaluation with the reference's module surface (reference utility/batch_test.py):
``data_generator``, ``Ks``, ``USR_NUM``, ``ITEM_NUM``, ``BATCH_SIZE``, ``test_torch`` ...

``test_torch`` keeps its signature and result dict, but the work is one HIP launch that scores
every listed user against all items with fp32 MFMA, removes the user's train items and keeps the
top max(Ks) by (score desc, item id asc) on the device (llmrec_score_topk_f32), a second tiny
launch for the hit vectors, and vectorised metrics on the 50-wide hit matrix. The reference moves
each 2048 x I score block to the host and ranks it per user in Python (batch_test.py:149-157)."""
import heapq
import multiprocessing

import numpy as np
import torch

import utility.metrics as metrics
from utility.load_data import Data
from utility.parser import parse_args

cores = multiprocessing.cpu_count() // 5

args = parse_args()
Ks = eval(args.Ks)

data_generator = Data(path=args.data_path + args.dataset, batch_size=args.batch_size)
USR_NUM, ITEM_NUM = data_generator.n_users, data_generator.n_items
N_TRAIN, N_TEST = data_generator.n_train, data_generator.n_test
BATCH_SIZE = args.batch_size


def ranklist_by_heapq(user_pos_test, test_items, rating, Ks):
    """Host ranking of one user from a dense rating vector (reference batch_test.py:21-36)."""
    top = heapq.nlargest(max(Ks), test_items, key=lambda i: rating[i])
    pos = set(user_pos_test)
    return [1 if i in pos else 0 for i in top], 0.


def get_performance(user_pos_test, r, auc, Ks):
    return {'recall': np.array([metrics.recall_at_k(r, K, len(user_pos_test)) for K in Ks]),
            'precision': np.array([metrics.precision_at_k(r, K) for K in Ks]),
            'ndcg': np.array([metrics.ndcg_at_k(r, K) for K in Ks]),
            'hit_ratio': np.array([metrics.hit_at_k(r, K) for K in Ks]), 'auc': auc}


def test_one_user(x):
    """(rating vector, uid, is_val) -> metric dict, on the host (reference batch_test.py:83-109)."""
    rating, u, is_val = x[0], x[1], x[-1]
    training_items = data_generator.train_items.get(u, [])
    user_pos_test = data_generator.val_set[u] if is_val else data_generator.test_set[u]
    banned = set(training_items)
    test_items = [i for i in range(ITEM_NUM) if i not in banned]
    r, auc = ranklist_by_heapq(user_pos_test, test_items, rating, Ks)
    if args.test_flag != 'part':
        order = sorted(test_items, key=lambda i: rating[i], reverse=True)
        pos = set(user_pos_test)
        auc = metrics.auc(ground_truth=[1 if i in pos else 0 for i in order], prediction=[rating[i] for i in order])
    return get_performance(user_pos_test, r, auc, Ks)


def topk_lists(ua_embeddings, ia_embeddings, users_to_test):
    """Device top-max(Ks) item ids [n, Kmax] (int32, -1 = none) for the listed users."""
    from llmrec_amd import ops
    st = data_generator.device_state(ua_embeddings.device)
    q = torch.as_tensor(list(users_to_test), dtype=torch.int64, device=ua_embeddings.device)
    idx, _ = ops.score_topk(ua_embeddings, ia_embeddings, q, st["train"], max(Ks))
    return q, idx


def test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val, drop_flag=False, batch_test_flag=False, topk=None):
    """topk: optional (query tensor, ranked lists) already computed on the device for exactly users_to_test
    (the graph-captured evaluation of llmrec_amd.fused.FusedStep.eval_topk)."""
    result = {'precision': np.zeros(len(Ks)), 'recall': np.zeros(len(Ks)), 'ndcg': np.zeros(len(Ks)),
              'hit_ratio': np.zeros(len(Ks)), 'auc': 0.}
    n_test_users = len(users_to_test)
    if n_test_users == 0:
        return result
    test_users = users_to_test if topk is not None else list(users_to_test)     # (with device lists in hand the host copy is not needed)
    from llmrec_amd import ops
    held = data_generator.val_set if is_val else data_generator.test_set
    if args.test_flag == 'part':
        st = data_generator.device_state(ua_embeddings.device)
        q, idx = topk if topk is not None else topk_lists(ua_embeddings, ia_embeddings, test_users)
        rp, ci = st["val"] if is_val else st["test"]
        hits = ops.topk_hits(idx, q, rp, ci)
        # precision / recall / ndcg / hit-ratio per user on the device (llmrec_topk_metrics, the formulae of
        # utility/metrics.py); only the 4 x len(Ks) sums come back
        sums = ops.topk_metrics(idx, hits, q, rp, Ks).sum(0).cpu().numpy() / n_test_users
        for j, k in enumerate(('precision', 'recall', 'ndcg', 'hit_ratio')):
            result[k] = sums[j]
        return result
    # test_flag == 'full': AUC needs every item's score -> score blocks on the device, rank on the host
    u_batch_size = BATCH_SIZE * 2
    for start in range(0, n_test_users, u_batch_size):
        blk = test_users[start:start + u_batch_size]
        q = torch.as_tensor(blk, dtype=torch.int64, device=ua_embeddings.device)
        rate = ops.scores(ua_embeddings, ia_embeddings, q).cpu().numpy()
        for row, u in enumerate(blk):
            re = test_one_user((rate[row], u, is_val))
            for k in ('precision', 'recall', 'ndcg', 'hit_ratio', 'auc'):
                result[k] += re[k] / n_test_users
    return result
