"""Bitwise reproducibility of the training step (VERDICT r05 next #1b).

The reference's step is deterministic on CPU given the seed (main.py:228-283: index_put(accumulate) adds the gradient rows several samples
share in sample order; SURVEY.md 8c "same-seed reruns agree"). Rounds 1 - 5 scattered the BPR gradient rows with fp32 atomicAdd, so two
runs of one seed drifted apart (final E_i 1.0e-3 vs 3.6e-3 from the reference on two boxes). Round 6: the scatter has one owner per
destination row (llmrec_bpr_scatter_plan + the run-owner backward, csrc/bpr.hip) and this file holds the step to it:

  * the same seed, two FRESH Trainers, one epoch each (79 / 86 optimiser steps + one evaluation) on nf_mid_lr and ml_mid, on every execution
    path -> every parameter, both Adam moments and the evaluation's embeddings `torch.equal`;
  * the paths that issue the SAME launches in a different way (eager launches on four streams, eager launches on one stream, HIP-graph
    replay fed through the packed H2D copy) must agree with EACH OTHER bit for bit as well: a difference there is a cross-stream race,
    not rounding."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._dropin import load_dropin
from tests.conftest import GOLDEN
from oracle import make_trajectory as MT

PATHS = {
    # name: environment of the run
    "fused": {"LLMREC_FUSED": "1", "LLMREC_GRAPH": "0"},
    "streams0": {"LLMREC_FUSED": "1", "LLMREC_GRAPH": "0", "LLMREC_STREAMS": "0"},
    "graph": {"LLMREC_FUSED": "1", "LLMREC_GRAPH": "1"},
    "device_sampler": {"LLMREC_FUSED": "1", "LLMREC_GRAPH": "1", "LLMREC_DEVICE_SAMPLER": "1"},
    "unfolded": {"LLMREC_FUSED": "1", "LLMREC_GRAPH": "0", "LLMREC_FOLD": "0"},
    "modular": {"LLMREC_FUSED": "0", "LLMREC_GRAPH": "0"},
}
SAME_LAUNCHES = ("fused", "streams0", "graph")          # identical kernels on identical operands: bit-identical results
_KNOBS = ("LLMREC_FUSED", "LLMREC_GRAPH", "LLMREC_STREAMS", "LLMREC_DEVICE_SAMPLER", "LLMREC_FOLD", "LLMREC_PREPROPAGATE", "LLMREC_GEMM",
          "LLMREC_WGRAD_ROWS")


@pytest.fixture(scope="module")
def datasets(tmp_path_factory):
    roots = {}
    for name in ("nf_mid_lr", "ml_mid"):
        root = str(tmp_path_factory.mktemp("repro_" + name))
        ds_dir, _ = MT.write_case_dataset(name, root)
        want = json.load(open(os.path.join(GOLDEN, name, "meta.json")))["digests"]
        assert MT.digests(ds_dir) == want
        roots[name] = root
    return roots


def _state_of_one_epoch(case, path, root, monkeypatch):
    """Train ONE epoch (and evaluate once) in a fresh Trainer; every tensor the step owns, on the host."""
    meta = json.load(open(os.path.join(GOLDEN, case, "meta.json")))
    for k in _KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in PATHS[path].items():
        monkeypatch.setenv(k, v)
    argv = ["--dataset", meta["config"]["dataset"], "--data_path", root + "/"] + meta["config"]["argv"] + ["--epoch", "1"]
    m = load_dropin(argv)
    m._progress = lambda it: it
    m.set_seed(m.args.seed)
    tr = m.Trainer(data_config={})
    tr.logger.logging = lambda s: None
    tr.train()
    torch.cuda.synchronize()
    out = {}
    for name, p in tr.model_mm.named_parameters():
        out["p/" + name] = p.detach().cpu().clone()
        st = tr.optimizer.state.get(p) if hasattr(tr.optimizer, "state") else None
        if st is not None and isinstance(st, tuple):
            out["m/" + name], out["v/" + name] = st[0].detach().cpu().clone(), st[1].detach().cpu().clone()
    if tr._fused:
        out["E_u"], out["E_i"] = tr._fused.E_u.detach().cpu().clone(), tr._fused.E_i.detach().cpu().clone()
    return out


def _diff(a, b):
    keys = sorted(set(a) | set(b))
    bad = []
    for k in keys:
        if k not in a or k not in b:
            bad.append((k, "missing"))
        elif not torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)):
            x, y = a[k].double(), b[k].double()
            bad.append((k, "rel L2 %.2e, %d of %d words differ" % (float((x - y).norm() / (y.norm() + 1e-300)),
                                                                  int((a[k].view(torch.int32) != b[k].view(torch.int32)).sum()), a[k].numel())))
    return bad


@pytest.mark.parametrize("case", ["nf_mid_lr", "ml_mid"])
def test_same_seed_same_bits_on_every_path(case, datasets, monkeypatch):
    states = {}
    for path in PATHS:
        if path in ("unfolded", "modular") and case != "nf_mid_lr":
            continue
        a = _state_of_one_epoch(case, path, datasets[case], monkeypatch)
        b = _state_of_one_epoch(case, path, datasets[case], monkeypatch)
        assert any(k.startswith("m/") for k in a), "no optimiser state captured"
        bad = _diff(a, b)
        print("[reproducible %s/%s] %d tensors (%d words): %s" % (case, path, len(a), sum(t.numel() for t in a.values()),
                                                                   "bit-identical" if not bad else bad[:4]))
        assert not bad, (path, bad[:6])
        states[path] = a
    ref = states[SAME_LAUNCHES[0]]
    for path in SAME_LAUNCHES[1:]:
        bad = _diff(ref, states[path])
        print("[reproducible %s] %s vs %s: %s" % (case, SAME_LAUNCHES[0], path, "bit-identical" if not bad else bad[:4]))
        assert not bad, (path, bad[:6])
