"""CPU: the oracle restatement (oracle/oracle.py) over a training HORIZON - the first epochs of the unmodified reference's
12 / 16-epoch trajectories (tests/golden/nf_mid*/trajectory.npz, oracle/make_trajectory.py) on the regenerated mid-size dataset:
identical sample stream, per-step loss / mf / emb and the epoch-end metric dicts. tests/test_oracle_golden.py pins the oracle
tensor by tensor over one tiny epoch; this pins it over hundreds of optimiser steps at 2500 x 3500."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import make_trajectory as MT
from llmrec_amd.synth import DATASET_KEYS
from tests._dropin import load_dropin
from tests.conftest import GOLDEN

TRAINABLE = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_trans.weight", "user_trans.bias",
             "item_trans.weight", "item_trans.bias", "user_id_embedding.weight", "item_id_embedding.weight"]
EPOCHS = 4


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    """case -> data root of its regenerated dataset; the bytes the reference trained on (content digests in meta.json)."""
    roots = {}
    for name in ("nf_mid", "ml_mid"):
        root = str(tmp_path_factory.mktemp("traj_cpu_" + name))
        ds_dir, _ = MT.write_case_dataset(name, root)
        got = MT.digests(ds_dir)
        for case, cfg in MT.CASES.items():
            if cfg["dataset"] == MT.CASES[name]["dataset"]:
                assert got == json.load(open(os.path.join(GOLDEN, case, "meta.json")))["digests"], case
                roots[case] = root
    return roots


def test_regenerated_dataset_has_every_user_in_the_test_split(dataset):
    d = json.load(open(os.path.join(dataset["nf_mid"], "netflix_valid_item", "test.json")))
    assert len(d) == 2500 and all(len(v) == 1 for v in d.values())


@pytest.mark.parametrize("case,n_epochs", [("nf_mid_lr", EPOCHS), ("ml_mid", 2)])
def test_oracle_follows_the_reference_trajectory(case, n_epochs, dataset):
    dataset = dataset[case]
    z = np.load(os.path.join(GOLDEN, case, "trajectory.npz"))
    meta = json.load(open(os.path.join(GOLDEN, case, "meta.json")))
    keys = DATASET_KEYS[meta["config"]["dataset"]]
    cfg = O.Config.from_args(meta["args"], keys)
    data = O.load_dataset(os.path.join(dataset, meta["config"]["dataset"]), keys)
    a_ui, a_iu = O.normalized_graphs(data.train_mat)
    # initial parameters: the drop-in's initialisation, bit-identical to the reference's for the same seed
    # (tests/test_host_cpu.py::test_dropin_init_matches_reference_init)
    m = load_dropin(["--dataset", meta["config"]["dataset"], "--data_path", dataset + "/"] + meta["config"]["argv"])
    m.set_seed(m.args.seed)
    tr = m.Trainer(data_config={})
    sd = tr.model_mm.state_dict()
    params = {k: sd[k].detach().cpu().clone().requires_grad_(True) for k in TRAINABLE}
    opt = O.AdamW(params, lr=cfg.lr)
    seed = meta["args"]["seed"]
    np.random.seed(seed); random.seed(seed)                 # main.py:355-357: the sampler's two streams
    exist = list(data.train_items.keys())
    n_batch = int(z["n_batch"])
    assert n_batch == data.n_train // cfg.batch_size + 1
    users_to_test = list(data.test_set.keys())
    ev = 0
    best = 0.0
    for epoch in range(n_epochs):
        for b in range(n_batch):
            s = epoch * n_batch + b
            u, p, n = O.sample_batch(exist, data.train_items, data.n_items, data.n_users, cfg.batch_size)
            u, p, n = O.augment_batch(u, p, n, data.aug_dict, data.n_items, cfg.aug_sample_rate)
            fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
            loss, parts = O.step_loss(fw, u, p, n, data.n_items, cfg)
            grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
            opt.step(grads)
            mf, emb = (float(x) for x in parts["bpr"][0])
            assert abs(float(loss) - z["step_loss"][s]) <= 2e-5 * abs(z["step_loss"][s]), (s, float(loss), z["step_loss"][s])
            assert abs(mf - z["step_mf"][s]) <= 2e-5 * abs(z["step_mf"][s]) and abs(emb - z["step_emb"][s]) <= 2e-5 * abs(z["step_emb"][s]), s
        with torch.no_grad():
            fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
        res, _ = O.evaluate(fw["E_u"].numpy(), fw["E_i"].numpy(), users_to_test, data.train_items, data.test_set, cfg.Ks, batch_size=cfg.batch_size)
        assert int(z["eval_after_steps"][ev]) == (epoch + 1) * n_batch
        for k in ("precision", "recall", "ndcg", "hit_ratio"):   # one user's hit moves recall by 0.0004
            assert np.abs(res[k] - z["eval_" + k][ev]).max() <= 0.0004 + 1e-12, (epoch, k, res[k], z["eval_" + k][ev])
        ev += 1
        if res["recall"][1] > best:                         # main.py:314-317: a new best triggers a second evaluation
            best = res["recall"][1]
            ev += 1
