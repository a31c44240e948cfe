"""CPU: the oracle restatement (oracle/oracle.py) against vectors captured from the UNMODIFIED
reference (oracle/make_golden.py). This is what pins the oracle."""
import numpy as np
import torch

from oracle import oracle as O
from llmrec_amd.synth import DATASET_KEYS

TRAINABLE = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias",
             "user_trans.weight", "user_trans.bias", "item_trans.weight", "item_trans.bias",
             "user_id_embedding.weight", "item_id_embedding.weight"]


def close(a, b, rtol=2e-5, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() <= atol + rtol * scale


def run_oracle(g):
    keys = DATASET_KEYS[g.dataset]
    cfg = O.Config.from_args(g.args, keys)
    data = O.load_dataset(g.data_dir, keys)
    a_ui, a_iu = O.normalized_graphs(data.train_mat)
    init = g.init_params()
    params = {k: torch.tensor(init[k]).clone().requires_grad_(True) for k in TRAINABLE}
    opt = O.AdamW(params, lr=cfg.lr)
    detail = set(g.detail_steps())
    report = []
    for s in range(g.n_steps):
        users, pos, neg = (g.z["step%d/%s" % (s, n)] for n in ("users", "pos", "neg"))
        fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
        loss, parts = O.step_loss(fw, users, pos, neg, data.n_items, cfg)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        gold_bpr = g.z["step%d/bpr" % s]
        mine = np.array([[float(m), float(e)] for m, e in parts["bpr"]])
        assert close(mine, gold_bpr, rtol=1e-5), (s, mine, gold_bpr)
        assert close(float(parts["feat_reg"]), g.z["step%d/feat_reg" % s], rtol=1e-5)
        if s in detail:
            for nm in ("E_u", "E_i", "img_i", "txt_i", "img_u", "txt_u", "P_usr", "prof_u", "prof_i"):
                assert close(fw[nm].detach().numpy(), g.z["step%d/%s" % (s, nm)]), (s, nm)
            for k in keys:
                assert close(fw["att_u"][k].detach().numpy(), g.z["step%d/att_u/%s" % (s, k)])
                assert close(fw["att_i"][k].detach().numpy(), g.z["step%d/att_i/%s" % (s, k)])
            for nm in TRAINABLE:
                assert close(grads[nm].numpy(), g.z["step%d/grad/%s" % (s, nm)], rtol=1e-4), (s, nm)
        opt.step(grads)
        if s in detail:
            for nm in TRAINABLE:
                assert close(params[nm].detach().numpy(), g.z["step%d/param/%s" % (s, nm)], rtol=1e-6), (s, nm)
        report.append(float(loss))
    return cfg, data, params, a_ui, a_iu, report


def test_training_steps_match_reference(golden):
    cfg, data, params, a_ui, a_iu, report = run_oracle(golden)
    assert np.all(np.isfinite(report))
    # epoch-end evaluation (reference main.py:297-300): embeddings, ranked lists, metrics
    with torch.no_grad():
        fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
    assert close(fw["E_u"].numpy(), golden.z["eval/E_u"], rtol=1e-5)
    assert close(fw["E_i"].numpy(), golden.z["eval/E_i"], rtol=1e-5)
    users = golden.z["eval/users"].tolist()
    # rank with the reference's own embeddings so fp32 near-ties cannot flip a position
    res, lists = O.evaluate(golden.z["eval/E_u"], golden.z["eval/E_i"], users, data.train_items,
                            data.test_set, cfg.Ks, batch_size=cfg.batch_size)
    gold = golden.z["eval/topk"]
    for row, l in enumerate(lists):
        assert list(l) == [int(x) for x in gold[row] if x >= 0], row
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(res[k], golden.z["eval/" + k], rtol=0, atol=1e-12), k


def test_rank_rule_matches_heapq(golden):
    """rank_topk (heapq restatement) == rank_topk_np (lexsort) incl. exact ties."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        n = int(rng.integers(5, 90))
        s = rng.integers(0, 6, size=n).astype(np.float32)        # many exact ties
        train = rng.choice(n, size=int(rng.integers(0, n // 2 + 1)), replace=False).tolist()
        a = O.rank_topk(s, train, 50)
        b = O.rank_topk_np(s, train, 50).tolist()
        assert a == b


def test_host_sampler_stream_matches_reference(golden):
    """Same seed -> same (users, pos, neg) stream as the reference (load_data.py:157-195,
    main.py:216-224), which is what makes the drop-in reproduce the reference's runs."""
    import random
    keys = DATASET_KEYS[golden.dataset]
    cfg = O.Config.from_args(golden.args, keys)
    data = O.load_dataset(golden.data_dir, keys)
    seed = golden.args["seed"]
    np.random.seed(seed); random.seed(seed)
    exist_users = list(data.train_items.keys())
    for s in range(golden.n_steps):
        u, p, n = O.sample_batch(exist_users, data.train_items, data.n_items, data.n_users, cfg.batch_size)
        u, p, n = O.augment_batch(u, p, n, data.aug_dict, data.n_items, cfg.aug_sample_rate)
        assert u == golden.z["step%d/users" % s].tolist()
        assert p == golden.z["step%d/pos" % s].tolist()
        assert n == golden.z["step%d/neg" % s].tolist()
