"""TEST INFRASTRUCTURE ONLY - generates tests/golden/* by running the UNMODIFIED reference.

Run in the build container (needs /root/reference):

    python oracle/make_golden.py            # all cases
    python oracle/make_golden.py nf_tiny    # one case

For each case it writes a tiny synthetic dataset (llmrec_amd/synth.py, seeded) under
``tests/golden/<case>/data/<dataset>/``, imports the reference through ``oracle/ref_loader.py``
(shims only, no arithmetic changes), trains one epoch on CPU and records, per step: the sampled
(users, pos, neg) triples after LLM-augmentation (reference main.py:213-224), the 14 forward
outputs (Models.py:199), every bpr_loss result (main.py:330-342), the feature regulariser
(main.py:151-156), parameter gradients, and post-AdamW parameters (main.py:276-278); and for the
epoch-end evaluation the per-user ranked top-50 lists (utility/batch_test.py:21-36) and the
metric dict (batch_test.py:112-169). Library versions are stored with the vectors because the
reference publishes no golden values of its own (parity is pinned by these runs).
"""
from __future__ import annotations

import importlib.util
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

CASES = {
    # Netflix-shaped keys, L=2, defaults (aug 0.1, prune 0.71); cfg 1/2 of BASELINE.json in miniature
    "nf_tiny": dict(
        dataset="netflix_valid_item", n_users=96, n_items=80, n_edges=420, seed=3,
        image_dim=24, text_dim=40, llm_dim=56,
        argv=["--batch_size", "32", "--epoch", "1", "--seed", "2022", "--debug"]),
    # same data, augmentation and pruning off (README ablation flags; cfg 1 "aug off")
    "nf_tiny_noaug": dict(
        dataset="netflix_valid_item", n_users=96, n_items=80, n_edges=420, seed=3,
        image_dim=24, text_dim=40, llm_dim=56,
        argv=["--batch_size", "32", "--epoch", "1", "--seed", "7", "--debug",
              "--aug_sample_rate", "0", "--prune_loss_drop_rate", "0"]),
    # MovieLens-shaped keys, L=3, d=16 (cfg 3 in miniature)
    "ml_tiny": dict(
        dataset="preprocessed_raw_MovieLens", n_users=72, n_items=120, n_edges=500, seed=11,
        image_dim=20, text_dim=28, llm_dim=36,
        argv=["--batch_size", "24", "--epoch", "1", "--seed", "2022", "--debug",
              "--embed_size", "16", "--weight_size", "[16,16,16]", "--layers", "2"]),
}


def _load_synth():
    spec = importlib.util.spec_from_file_location("_synth", os.path.join(ROOT, "llmrec_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["_synth"] = mod          # dataclasses resolves the module by name
    spec.loader.exec_module(mod)
    return mod


def run_case(name: str):
    import numpy as np

    cfg = CASES[name]
    case_dir = os.path.join(GOLD, name)
    shutil.rmtree(case_dir, ignore_errors=True)
    data_root = os.path.join(case_dir, "data")
    ds_dir = os.path.join(data_root, cfg["dataset"])
    synth = _load_synth()
    stats = synth.write_dataset(ds_dir, cfg["n_users"], cfg["n_items"], cfg["n_edges"], seed=cfg["seed"],
                                image_dim=cfg["image_dim"], text_dim=cfg["text_dim"], llm_dim=cfg["llm_dim"],
                                keys=synth.DATASET_KEYS[cfg["dataset"]], max_deg=40)

    sys.path.insert(0, HERE)
    import ref_loader
    argv = ["--dataset", cfg["dataset"], "--data_path", data_root + "/"] + cfg["argv"]
    ref = ref_loader.load_reference(argv)
    import torch
    import scipy

    ref.set_seed(ref.args.seed)
    trainer = ref.Trainer(data_config={"n_users": ref.data_generator.n_users,
                                       "n_items": ref.data_generator.n_items})
    out = {}
    for k, v in trainer.model_mm.state_dict().items():
        out["init/" + k] = v.detach().clone().numpy()

    n_batch = ref.data_generator.n_train // ref.args.batch_size + 1          # main.py:203
    DETAIL_STEPS = {0, 1, n_batch - 1}
    rec = {"step": -1, "lists": None, "bpr": [], "evals": []}

    # -- samples: keep the list objects; main.py:222-224 extends them in place with the aug triples
    orig_sample = ref.data_generator.sample

    def sample_wrap():
        u, p, n = orig_sample()
        rec["lists"] = (u, p, n)
        return u, p, n
    ref.data_generator.sample = sample_wrap

    # -- forward outputs
    names14 = ["E_u", "E_i", "img_i", "txt_i", "img_u", "txt_u", "P_usr", None, "prof_u", "prof_i",
               None, None, None, None]

    def fwd_hook(mod, inp, outp):
        if not mod.training:
            rec["eval_E"] = (outp[0].detach().clone().numpy(), outp[1].detach().clone().numpy())
            return
        rec["step"] += 1
        s = rec["step"]
        u, p, n = rec["lists"]
        out["step%d/users" % s] = np.asarray(u, dtype=np.int64)
        out["step%d/pos" % s] = np.asarray(p, dtype=np.int64)
        out["step%d/neg" % s] = np.asarray(n, dtype=np.int64)
        rec["bpr"] = []
        if s not in DETAIL_STEPS:        # samples for every step; tensors only for a few (fixture size)
            return
        for nm, t in zip(names14, outp):
            if nm is not None:
                out["step%d/%s" % (s, nm)] = t.detach().clone().numpy()
        for key, t in outp[10].items():
            out["step%d/att_u/%s" % (s, key)] = t.detach().clone().numpy()
        for key, t in outp[11].items():
            out["step%d/att_i/%s" % (s, key)] = t.detach().clone().numpy()
    trainer.model_mm.register_forward_hook(fwd_hook)

    # -- losses
    orig_bpr = trainer.bpr_loss

    def bpr_wrap(users, pos, neg):
        mf, emb, reg = orig_bpr(users, pos, neg)
        rec["bpr"].append((float(mf), float(emb)))
        return mf, emb, reg
    trainer.bpr_loss = bpr_wrap

    orig_freg = trainer.feat_reg_loss_calculation

    def freg_wrap(*a):
        r = orig_freg(*a)
        rec["feat_reg"] = float(r)
        return r
    trainer.feat_reg_loss_calculation = freg_wrap

    # -- grads + post-step params
    orig_step = trainer.optimizer.step

    def step_wrap(*a, **k):
        s = rec["step"]
        out["step%d/bpr" % s] = np.asarray(rec["bpr"], dtype=np.float64)          # [8, 2] (mf, emb)
        out["step%d/feat_reg" % s] = np.float64(rec["feat_reg"])
        detail = s in DETAIL_STEPS
        for nm, p_ in trainer.model_mm.named_parameters():
            if p_.grad is not None and detail:
                out["step%d/grad/%s" % (s, nm)] = p_.grad.detach().clone().numpy()
        r = orig_step(*a, **k)
        for nm, p_ in trainer.model_mm.named_parameters():
            if p_.grad is not None and detail:
                out["step%d/param/%s" % (s, nm)] = p_.detach().clone().numpy()
        return r
    trainer.optimizer.step = step_wrap

    # -- evaluation: ranked lists via the heapq call in utility/batch_test.py:27
    import utility.batch_test as bt
    import heapq as _heapq

    class HeapqProxy:
        def __init__(self):
            self.lists = []

        def nlargest(self, k, it, key=None):
            r = _heapq.nlargest(k, it, key=key)
            self.lists.append(list(r))
            return r
    proxy = HeapqProxy()
    bt.heapq = proxy

    orig_test_torch = ref.test_torch

    def test_torch_wrap(ua, ia, users_to_test, is_val, *a, **k):
        proxy.lists = []
        res = orig_test_torch(ua, ia, users_to_test, is_val, *a, **k)
        rec["evals"].append((list(users_to_test), [list(x) for x in proxy.lists], res, rec["eval_E"]))
        return res
    ref.test_torch = test_torch_wrap

    trainer.train()

    out["n_steps"] = np.int64(rec["step"] + 1)
    users_to_test, lists, res, (eu, ei) = rec["evals"][0]          # first eval = after epoch 0
    kmax = max(len(x) for x in lists)
    topk = -np.ones((len(lists), kmax), dtype=np.int64)
    for r_, l_ in enumerate(lists):
        topk[r_, :len(l_)] = l_
    out["eval/users"] = np.asarray(users_to_test, dtype=np.int64)
    out["eval/topk"] = topk
    out["eval/E_u"] = eu
    out["eval/E_i"] = ei
    for k_ in ("precision", "recall", "ndcg", "hit_ratio"):
        out["eval/" + k_] = np.asarray(res[k_], dtype=np.float64)
    np.savez_compressed(os.path.join(case_dir, "golden.npz"), **out)

    # files the reference writes into the dataset dir at init (main.py:66,78) are not inputs
    for extra in ("augmented_user_init_embedding_final", "augmented_total_embed_dict"):
        try:
            os.remove(os.path.join(ds_dir, extra))
        except FileNotFoundError:
            pass
    meta = {"case": name, "config": {k: v for k, v in cfg.items()}, "stats": stats,
            "argv": argv[:2] + ["--data_path", "<case>/data/"] + cfg["argv"],
            "versions": {"torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                         "python": sys.version.split()[0]},
            "torch_threads": torch.get_num_threads(), "n_steps": int(rec["step"] + 1),
            "args": {k: v for k, v in vars(ref.args).items()}}
    with open(os.path.join(case_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("[golden] %s: %d steps, %d eval users -> %s" % (name, rec["step"] + 1, len(lists), case_dir))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    if len(names) == 1 and os.environ.get("_GOLDEN_CHILD") == "1":
        run_case(names[0])
    else:
        # one process per case: the reference parses sys.argv and builds its dataset at import
        for n in names:
            env = dict(os.environ, _GOLDEN_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), n], check=True, env=env, cwd="/tmp")
