"""GPU tests of the reference's off-by-default options on the drop-in (they run on the modular path, never on the
fused step): --test_flag full (AUC from full score rows, reference utility/batch_test.py:38-68,105-108), --mask /
--mask_rate (feature masking + attribute-restoration loss, Models.py:131-142, main.py:258-271) and --drop_rate > 0
(nn.Dropout on the projections, Models.py:145-150). The reference ships no vectors for them: these are property
tests (the arithmetic they add is torch's own), plus the check that the fused step steps aside."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._dropin import load_dropin, golden_argv
from tests.conftest import GoldenCase


def _trainer(extra, seed=3):
    g = GoldenCase("nf_tiny")
    m = load_dropin(golden_argv(g) + list(extra))
    m.set_seed(seed)
    return g, m, m.Trainer(data_config={})


def test_test_flag_full_gives_the_same_ranking_metrics_plus_auc():
    g, m, tr = _trainer([])
    users = g.z["eval/users"].tolist()
    part = tr.test(users, is_val=False)
    g2, m2, tr2 = _trainer(["--test_flag", "full"])
    full = tr2.test(users, is_val=False)
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(part[k], full[k], rtol=0, atol=1e-12), k
    # AUC: independent Mann-Whitney statistic over the non-train items, from the kernel's score rows
    from llmrec_amd import ops
    tr2.model_mm.eval()
    with torch.no_grad():
        fw = tr2.model_mm(tr2.ui_graph, tr2.iu_graph)
    dg = m2.data_generator
    S = ops.scores(fw[0], fw[1], torch.tensor(users, device="cuda")).cpu().numpy()
    aucs = []
    for r, u in enumerate(users):
        banned = set(dg.train_items.get(u, []))
        cand = np.array([i for i in range(dg.n_items) if i not in banned])
        pos = np.isin(cand, dg.test_set[u])
        s = S[r][cand].astype(np.float64)
        order = s.argsort(kind="stable"); ranks = np.empty(len(s)); ranks[order] = np.arange(1, len(s) + 1)
        for v in np.unique(s):                                      # average ranks of ties
            tie = s == v
            ranks[tie] = ranks[tie].mean()
        n_p, n_n = pos.sum(), (~pos).sum()
        aucs.append((ranks[pos].sum() - n_p * (n_p + 1) / 2) / (n_p * n_n))
    assert abs(full["auc"] - float(np.mean(aucs))) < 1e-9
    assert 0.0 < full["auc"] < 1.0 and part["auc"] == 0.0


def test_mask_branch_runs_on_the_modular_path_and_masks_rows():
    g, m, tr = _trainer(["--mask", "True", "--mask_rate", "0.25", "--att_re_rate", "0.001"])
    assert tr._fused_step() is False                                # the fused step steps aside
    model = tr.model_mm
    before_u = model.user_feats.clone()
    users, pos, neg = (torch.tensor(g.z["step0/" + n]).cuda() for n in ("users", "pos", "neg"))
    dec_before = [p.detach().clone() for p in tr.decoder.parameters()]
    loss, mf, emb = tr.train_step(users, pos, neg)
    assert np.isfinite(float(loss))
    # a quarter of the user rows now hold one common vector (the column mean at masking time)
    changed = (model.user_feats != before_u).any(dim=1)
    assert abs(int(changed.sum()) - int(0.25 * model.n_users)) <= 1
    rows = model.user_feats[changed]
    assert torch.allclose(rows, rows[0].expand_as(rows))
    assert torch.allclose(rows[0], before_u.mean(0), rtol=1e-5, atol=1e-6)
    # the restoration loss reaches the model (loss differs from the unmasked step) but the decoder's optimizer never steps (reference main.py:276-278)
    for a, b in zip(dec_before, tr.decoder.parameters()):
        assert torch.equal(a, b.detach())
    g0, m0, tr0 = _trainer([])
    loss0, _, _ = tr0.train_step(users, pos, neg)
    assert abs(float(loss) - float(loss0)) > 1e-7


def test_mask_rate_without_mask_flag_still_masks_users_and_leaves_the_fused_step():
    g, m, tr = _trainer(["--mask_rate", "0.2"])
    assert tr._fused_step() is False                                # reference Models.py:139-142 masks users whenever mask_rate > 0
    before = tr.model_mm.user_feats.clone()
    users, pos, neg = (torch.tensor(g.z["step0/" + n]).cuda() for n in ("users", "pos", "neg"))
    tr.train_step(users, pos, neg)
    changed = (tr.model_mm.user_feats != before).any(dim=1)
    assert abs(int(changed.sum()) - int(0.2 * tr.model_mm.n_users)) <= 1


def test_dropout_is_inverted_dropout_in_training_and_identity_in_eval():
    g, m, tr = _trainer(["--drop_rate", "0.5"])
    assert tr._fused_step() is False
    model = tr.model_mm
    model.eval()
    with torch.no_grad():
        ref = model(tr.ui_graph, tr.iu_graph)[6]                    # user_feats projection (no propagation after it)
    model.train()
    torch.manual_seed(0)
    with torch.no_grad():
        out = model(tr.ui_graph, tr.iu_graph)[6]
    zero = out == 0
    frac = float(zero.float().mean())
    assert 0.42 < frac < 0.58                                        # 96 x 64 Bernoulli(0.5) draws
    assert torch.allclose(out[~zero], 2.0 * ref[~zero], rtol=1e-6, atol=0)
    users, pos, neg = (torch.tensor(g.z["step0/" + n]).cuda() for n in ("users", "pos", "neg"))
    loss, _, _ = tr.train_step(users, pos, neg)
    assert np.isfinite(float(loss))
