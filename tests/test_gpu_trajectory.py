"""Long-horizon accuracy parity (north_star: "Recall@20 within +-0.002 of reference").

tests/golden/{nf_mid,nf_mid_lr,ml_mid}/trajectory.npz hold 12 / 16 / 10 epochs (948 / 1264 / 860 optimiser steps) of the UNMODIFIED reference's
``Trainer.train()`` on a 2500-user x 3500-item Netflix-shaped set and a 3000 x 2400 MovieLens-shaped one (three propagation layers, more users than
items) with planted communities (oracle/make_trajectory.py): per
epoch the logged sums, the metric dict of every evaluation and the best-epoch / early-stopping log lines. Here the drop-in
(`main.Trainer.train()`, reference main.py:189-327) trains on the regenerated dataset (content digests checked) with the HOST
sampler - the reference's RNG stream, so every batch is identical - on each execution path, and must stay on the reference's
trajectory: Recall / NDCG / precision / hit-ratio @10/20/50 within +-0.002 at EVERY evaluation, epoch loss and mf_loss within 1e-3
relative at every epoch, the same sequence of best-epoch / early-stopping decisions, and E_u / E_i after the last epoch."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._dropin import load_dropin
from tests.conftest import GOLDEN
from oracle import make_trajectory as MT

METRIC_TOL = 0.002       # north_star's Recall@20 tolerance, applied to all 12 metric values of every evaluation
LOSS_RTOL = 1e-3
# ||E_gpu - E_reference|| / ||E_reference|| after the LAST epoch (~1000 AdamW steps). Since round 6 the step is bitwise reproducible
# (tests/test_gpu_reproducible.py: no float atomics in the gradient scatter), so this number is a CONSTANT of (case, path), not a sample:
# measured (E_u, E_i), identical on every box so far -
#   nf_mid     (lr 1e-4)  4.3e-7, 4.1e-7 on every path
#   nf_mid_lr  (lr 1e-3)  3.5e-4, 3.0e-4 on every path
#   ml_mid     (lr 1e-3)  4.4e-4, 6.1e-4 pre-propagated; 3.65e-3, 5.11e-3 in the reference's order of operations (graph / fused_reference_order)
# - the per-step difference is 1e-6-class on both orders (tests/test_gpu_step.py, bench.py's gate); over 860 steps at lr 1e-3 it is amplified
# by the trajectory itself, differently for each rounding pattern. The gate is twice the largest measured value of the case. (Round 5's single
# 3e-3 was calibrated on ONE run of a then non-reproducible step - 1.0e-3 on the builder's box, 3.6e-3 on the driver's - and failed there.)
E_L2_TOL = {"nf_mid": 1e-6, "nf_mid_lr": 7e-4, "ml_mid": 1.1e-2}


@pytest.fixture(scope="module")
def datasets(tmp_path_factory):
    """case -> data root of its regenerated dataset (nf_mid and nf_mid_lr train on the same one), content digests checked."""
    roots, by_cfg = {}, {}
    for name, cfg in MT.CASES.items():
        key = json.dumps({k: v for k, v in cfg.items() if k != "argv"}, sort_keys=True)
        if key not in by_cfg:
            root = str(tmp_path_factory.mktemp("traj_" + name))
            ds_dir, _ = MT.write_case_dataset(name, root)
            by_cfg[key] = (root, MT.digests(ds_dir))
        root, got = by_cfg[key]
        want = json.load(open(os.path.join(GOLDEN, name, "meta.json")))["digests"]
        assert got == want, "the regenerated dataset differs from the one the reference trained on: %s" % [k for k in want if got[k] != want[k]]
        roots[name] = root
    return roots


def _decisions(lines):
    """The best-epoch / early-stopping decisions of main.py:314-325 as the log shows them (metric values stripped)."""
    out = []
    for l in lines:
        l = l.split("  ", 1)[-1].strip() if l[:2] == "20" else l.strip()      # (the logger prefixes a timestamp)
        if l.startswith("Test_Recall"):
            out.append("best")
        elif l.startswith("#####"):
            out.append(l)
    return out


@pytest.mark.parametrize("path", ["fused", "graph", "fused_reference_order", "modular"])
@pytest.mark.parametrize("case", ["nf_mid", "nf_mid_lr", "ml_mid"])
def test_training_trajectory_tracks_reference(case, path, datasets, monkeypatch):
    if path == "modular" and case != "nf_mid_lr":
        pytest.skip("the per-op autograd path runs the nf_mid_lr horizon only (same code, the faster-moving trajectory)")
    z = np.load(os.path.join(GOLDEN, case, "trajectory.npz"))
    meta = json.load(open(os.path.join(GOLDEN, case, "meta.json")))
    monkeypatch.setenv("LLMREC_FUSED", "0" if path == "modular" else "1")
    monkeypatch.setenv("LLMREC_GRAPH", "1" if path == "graph" else "0")
    # the order of the two constant products on the item side: forced either way on the two fused paths, LEFT TO THE SHAPE RULE on the graph path
    # (pre-propagated iff U <= I: the Netflix-shaped cases pre-propagate, the MovieLens-shaped one projects first)
    if path == "graph":
        monkeypatch.delenv("LLMREC_PREPROPAGATE", raising=False)
    else:
        monkeypatch.setenv("LLMREC_PREPROPAGATE", "0" if path == "fused_reference_order" else "1")
    monkeypatch.delenv("LLMREC_DEVICE_SAMPLER", raising=False)            # host sampler = the reference's sample stream
    argv = ["--dataset", meta["config"]["dataset"], "--data_path", datasets[case] + "/"] + meta["config"]["argv"]
    m = load_dropin(argv)
    m._progress = lambda it: it
    m.set_seed(m.args.seed)
    tr = m.Trainer(data_config={})
    epochs, evals, lines = [], [], []
    tr._on_epoch = lambda ep, loss, mf, emb, ret, t: epochs.append((loss, mf, emb))
    orig_test = tr.test                                                   # (every evaluation of train() goes through Trainer.test; since round 6
                                                                          #  the graph path ends with the metrics and does not call test_torch)
    def test_wrap(*a, **k):
        res = orig_test(*a, **k)
        evals.append(np.stack([np.asarray(res[k_], dtype=np.float64) for k_ in ("precision", "recall", "ndcg", "hit_ratio")]))
        return res
    tr.test = test_wrap
    orig_log = tr.logger.logging
    tr.logger.logging = lambda s: (lines.append(str(s)), orig_log(s))[1]
    best_recall, _ = tr.train()

    n_ep = int(z["n_epochs"])
    assert len(epochs) == n_ep and len(evals) == z["eval_recall"].shape[0], (len(epochs), n_ep, len(evals))
    want_eval = np.stack([z["eval_precision"], z["eval_recall"], z["eval_ndcg"], z["eval_hit_ratio"]], axis=1)
    got_eval = np.stack(evals)
    diff = np.abs(got_eval - want_eval)
    worst_metric = float(diff.max())
    ep = np.asarray(epochs, dtype=np.float64)
    loss_rel = np.abs(ep[:, 0] - z["epoch_loss"]) / np.abs(z["epoch_loss"])
    mf_rel = np.abs(ep[:, 1] - z["epoch_mf"]) / np.abs(z["epoch_mf"])
    emb_rel = np.abs(ep[:, 2] - z["epoch_emb"]) / np.abs(z["epoch_emb"])
    e_rel = []
    if tr._fused:                                                         # the last evaluation's embeddings (after ~1000 AdamW steps)
        for got, want in ((tr._fused.E_u, z["final_E_u"]), (tr._fused.E_i, z["final_E_i"])):
            got = got.detach().cpu().numpy().astype(np.float64)
            e_rel.append(float(np.linalg.norm(got - want) / np.linalg.norm(want)))
    print("[trajectory %s/%s] %d epochs, %d evaluations: max |metric diff| %.2e (recall@20 %.2e, ndcg@20 %.2e), loss rel %.2e, mf rel %.2e, emb rel %.2e, "
          "evaluations with all 12 metrics EQUAL: %d/%d, final E_u / E_i rel L2 %s" % (case, path, n_ep, len(evals), worst_metric, float(diff[:, 1, 1].max()), float(diff[:, 2, 1].max()),
                                                            float(loss_rel.max()), float(mf_rel.max()), float(emb_rel.max()),
                                                            int((diff.max(axis=(1, 2)) == 0).sum()), len(evals), ["%.2e" % e for e in e_rel]))
    assert worst_metric <= METRIC_TOL, (np.argwhere(diff > METRIC_TOL)[:5], worst_metric)
    assert float(loss_rel.max()) <= LOSS_RTOL and float(mf_rel.max()) <= LOSS_RTOL and float(emb_rel.max()) <= LOSS_RTOL
    assert abs(best_recall - float(z["best_recall"])) <= METRIC_TOL
    assert _decisions(lines) == _decisions(meta["log_lines"])
    assert all(e <= E_L2_TOL[case] for e in e_rel), e_rel


def test_in_graph_sampler_path_tracks_the_oracle_over_many_steps(datasets, monkeypatch):
    """The path bench.py times - device sampler inside the step graph, graph replay - cannot replay the reference's batches (its sampler is
    not stream-compatible with the host RNG), so its horizon is checked against the ORACLE fed with the batches the device drew: 120 optimiser steps
    at the nf_mid shape and lr 1e-3, every step's logged scalars, then the evaluation's embeddings and metrics (bench.py's own gate covers two steps)."""
    from oracle import oracle as O
    from llmrec_amd.synth import DATASET_KEYS
    case, n_steps = "nf_mid_lr", 120
    meta = json.load(open(os.path.join(GOLDEN, case, "meta.json")))
    monkeypatch.setenv("LLMREC_FUSED", "1"); monkeypatch.setenv("LLMREC_GRAPH", "1"); monkeypatch.setenv("LLMREC_DEVICE_SAMPLER", "1")
    monkeypatch.delenv("LLMREC_PREPROPAGATE", raising=False)
    argv = ["--dataset", meta["config"]["dataset"], "--data_path", datasets[case] + "/"] + meta["config"]["argv"]
    m = load_dropin(argv)
    m.set_seed(m.args.seed)
    tr = m.Trainer(data_config={})
    keys = DATASET_KEYS[meta["config"]["dataset"]]
    cfg = O.Config.from_args(meta["args"], keys)
    data = O.load_dataset(os.path.join(datasets[case], meta["config"]["dataset"]), keys)
    a_ui, a_iu = O.normalized_graphs(data.train_mat)
    names = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_trans.weight", "user_trans.bias",
             "item_trans.weight", "item_trans.bias", "user_id_embedding.weight", "item_id_embedding.weight"]
    sd = tr.model_mm.state_dict()
    params = {k: sd[k].detach().cpu().clone().requires_grad_(True) for k in names}
    opt = O.AdamW(params, lr=cfg.lr)
    worst = {"loss": 0.0, "mf": 0.0, "emb": 0.0}
    seen = set()
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))                               # (the oracle's small CPU products crawl on all 256 cores of the GPU box's host)
    for s in range(n_steps):
        loss, mf, emb = (float(x) for x in tr.train_step_sampled())
        st = tr._fused_step().static
        nv = int(st["n_valid"])
        u, p, n = (st[k][:nv].cpu().numpy() for k in ("users", "pos", "neg"))
        seen.add(hash(u.tobytes()))
        fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
        l_or, parts = O.step_loss(fw, u, p, n, data.n_items, cfg)
        grads = dict(zip(params, torch.autograd.grad(l_or, list(params.values()))))
        opt.step(grads)
        mf_or, emb_or = (float(x.detach()) for x in parts["bpr"][0])
        worst["loss"] = max(worst["loss"], abs(loss - float(l_or.detach())) / abs(float(l_or.detach())))
        worst["mf"] = max(worst["mf"], abs(mf - mf_or) / abs(mf_or)); worst["emb"] = max(worst["emb"], abs(emb - emb_or) / abs(emb_or))
    torch.set_num_threads(threads)
    assert len(seen) == n_steps                                           # a fresh batch every replay
    users = list(m.data_generator.test_set.keys())
    ret = tr.test(users, is_val=False)
    with torch.no_grad():
        fw = O.forward(params, data.feats, a_ui, a_iu, cfg)
    fused = tr._fused_step()
    e_rel = [float((g.detach().cpu().double() - w.double()).norm() / w.double().norm()) for g, w in ((fused.E_u, fw["E_u"]), (fused.E_i, fw["E_i"]))]
    res, _ = O.evaluate(fw["E_u"].numpy(), fw["E_i"].numpy(), users, data.train_items, data.test_set, cfg.Ks, batch_size=cfg.batch_size)
    mdiff = max(float(np.abs(np.asarray(ret[k]) - res[k]).max()) for k in ("precision", "recall", "ndcg", "hit_ratio"))
    print("[in-graph sampler vs oracle] %d steps: loss rel %.2e, mf rel %.2e, emb rel %.2e, E_u / E_i rel L2 %s, max |metric diff| %.2e, recall@20 %.4f" % (
        n_steps, worst["loss"], worst["mf"], worst["emb"], ["%.2e" % e for e in e_rel], mdiff, float(ret["recall"][1])))
    assert worst["loss"] <= 1e-5 and worst["mf"] <= 1e-5 and worst["emb"] <= 1e-5, worst     # measured 4e-7 / 3e-7 / 2e-7
    assert all(e <= 1e-5 for e in e_rel), e_rel                                                  # measured 3.7e-7
    assert mdiff <= METRIC_TOL


def test_python_main_py_prints_the_reference_log(datasets):
    """The drop-in as a user runs it: `python main.py <the reference's flags>` in a fresh process (default mode), its log lines against the
    lines the UNMODIFIED reference logged on the same dataset (meta.json `log_lines`): the same sequence of epoch / best-epoch / early-stopping
    lines, every printed metric equal to the 5 decimals of the log format, the epoch losses within 1e-3 relative (the timings differ)."""
    import re
    import subprocess
    import sys
    case = "nf_mid_lr"
    meta = json.load(open(os.path.join(GOLDEN, case, "meta.json")))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = ["--dataset", meta["config"]["dataset"], "--data_path", datasets[case] + "/"] + meta["config"]["argv"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("LLMREC_")}
    r = subprocess.run([sys.executable, "main.py"] + argv, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]

    def parse(lines):
        out = []
        for l in lines:
            l = re.sub(r"^\d{4}-\d\d-\d\d \d\d:\d\d:\s+", "", l.strip())
            if l.startswith("Epoch") and "recall=" in l:
                nums = [float(x) for x in re.findall(r"-?\d+\.\d+", re.sub(r"\[[0-9.]+s \+ [0-9.]+s\]", "", l))]
                out.append(("epoch", int(re.match(r"Epoch (\d+)", l).group(1)), nums))
            elif l.startswith("Test_Recall"):
                out.append(("best", None, [float(x) for x in re.findall(r"-?\d+\.\d+", l.split(":", 1)[1])]))
            elif l.startswith("#####"):
                out.append(("stop", l, []))
        return out
    got, want = parse((r.stdout + r.stderr).splitlines()), parse(meta["log_lines"])
    assert [(k, t) for k, t, _ in got] == [(k, t) for k, t, _ in want], ([(k, t) for k, t, _ in got][:8], [(k, t) for k, t, _ in want][:8])
    worst_metric, worst_loss = 0.0, 0.0
    for (kind, _, a), (_, _, b) in zip(got, want):
        assert len(a) == len(b)
        if kind == "epoch":                                        # train==[loss=mf + emb + reg], then 16 metrics
            worst_loss = max(worst_loss, max(abs(x - y) / max(abs(y), 1e-12) for x, y in zip(a[:2], b[:2])))
            worst_metric = max(worst_metric, max(abs(x - y) for x, y in zip(a[4:], b[4:])))
        elif kind == "best":
            worst_metric = max(worst_metric, max(abs(x - y) for x, y in zip(a, b)))
    print("[python main.py vs the reference's log] %d lines: max |metric diff| %.1e, loss rel %.1e" % (len(got), worst_metric, worst_loss))
    assert worst_metric <= 1.5e-5 and worst_loss <= LOSS_RTOL
