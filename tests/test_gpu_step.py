"""GPU parity of whole training steps and the epoch-end evaluation of the drop-in (main.py /
Models.py / utility.batch_test on the HIP kernels) against vectors captured from the unmodified
reference (tests/golden, oracle/make_golden.py), with the reference's own samples injected."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._dropin import load_dropin, golden_argv
from llmrec_amd.synth import DATASET_KEYS

TRAINABLE = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias",
             "user_trans.weight", "user_trans.bias", "item_trans.weight", "item_trans.bias",
             "user_id_embedding.weight", "item_id_embedding.weight"]
RTOL = 1e-4          # north_star: loss / embeddings within 1e-4 relative


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("path", ["fused", "modular", "graph", "fused_f32", "fused_reference_order", "fused_unfolded", "graph_unfolded"])
def test_training_steps_and_eval_match_reference(golden, path, monkeypatch):
    """path: fused = hand-written backward over preallocated buffers (llmrec_amd/fused.py, the
    default: item-side operands pre-propagated, (A F) W^T), modular = torch.autograd over the per-op Functions, graph = fused +
    HIP graph replay, fused_f32 = fused with the bit-exact fp32-MFMA projection instead of the default 3-term bf16 split,
    fused_reference_order = fused with projection then propagation, A (F W^T), as the reference orders them; *_unfolded = LLMREC_FOLD=0, the
    36-launch step of round 4 (separate sum-of-squares, AdamW-counter, loss-value, assembly, axpy and row clean-up launches)."""
    assert torch.cuda.is_available()
    monkeypatch.setenv("LLMREC_FUSED", "0" if path == "modular" else "1")
    monkeypatch.setenv("LLMREC_GRAPH", "1" if path.startswith("graph") else "0")
    monkeypatch.setenv("LLMREC_FOLD", "0" if path.endswith("_unfolded") else "1")
    monkeypatch.setenv("LLMREC_GEMM", "f32" if path == "fused_f32" else "bf16x3")
    monkeypatch.setenv("LLMREC_PREPROPAGATE", "0" if path == "fused_reference_order" else "1")
    m = load_dropin(golden_argv(golden))
    m.set_seed(golden.args["seed"])
    tr = m.Trainer(data_config={})
    keys = DATASET_KEYS[golden.dataset]
    detail = set(golden.detail_steps())
    bpr_log = []
    tr._on_bpr = lambda mf, emb: bpr_log.append((mf.detach(), emb.detach()))
    worst = {}
    for s in range(golden.n_steps):
        users, pos, neg = (torch.tensor(golden.z["step%d/%s" % (s, n)]).cuda() for n in ("users", "pos", "neg"))
        if s in detail:
            tr.model_mm.train()
            with torch.no_grad():
                out = tr.model_mm(tr.ui_graph, tr.iu_graph, tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph)
            named = dict(E_u=out[0], E_i=out[1], img_i=out[2], txt_i=out[3], img_u=out[4], txt_u=out[5], P_usr=out[6],
                         prof_u=out[8], prof_i=out[9])
            for nm, t in named.items():
                e = rel(t.cpu().numpy(), golden.z["step%d/%s" % (s, nm)])
                worst["fwd/" + nm] = max(worst.get("fwd/" + nm, 0), e)
                assert e < RTOL, (s, nm, e)
            for k in keys:
                assert rel(out[10][k].cpu().numpy(), golden.z["step%d/att_u/%s" % (s, k)]) < RTOL
                assert rel(out[11][k].cpu().numpy(), golden.z["step%d/att_i/%s" % (s, k)]) < RTOL
        bpr_log.clear()
        loss, mf, emb = tr.train_step(users, pos, neg)
        got = np.array([[float(a), float(b)] for a, b in bpr_log])
        gold = golden.z["step%d/bpr" % s]
        assert got.shape == gold.shape
        assert np.abs(got - gold).max() <= RTOL * np.abs(gold).max(), (s, got, gold)
        if s in detail:
            params = dict(tr.model_mm.named_parameters())
            for nm in TRAINABLE:
                e = rel(params[nm].grad.cpu().numpy(), golden.z["step%d/grad/%s" % (s, nm)])
                worst["grad/" + nm] = max(worst.get("grad/" + nm, 0), e)
                assert e < RTOL, (s, "grad", nm, e)
                e = rel(params[nm].detach().cpu().numpy(), golden.z["step%d/param/%s" % (s, nm)])
                assert e < RTOL, (s, "param", nm, e)
    print("worst relative errors:", {k: "%.2e" % v for k, v in sorted(worst.items())})

    # epoch-end evaluation (reference main.py:297-300)
    users_to_test = golden.z["eval/users"].tolist()
    ret = tr.test(users_to_test, is_val=False)
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(ret[k], golden.z["eval/" + k], rtol=0, atol=1e-12), (k, ret[k], golden.z["eval/" + k])
    # ranked lists from the reference's own final embeddings: bit-exact incl. order
    eu = torch.tensor(golden.z["eval/E_u"]).cuda(); ei = torch.tensor(golden.z["eval/E_i"]).cuda()
    _, idx = m.topk_lists(eu, ei, users_to_test)
    idx = idx.cpu().numpy()
    gold = golden.z["eval/topk"]
    for row in range(len(users_to_test)):
        assert [int(x) for x in idx[row] if x >= 0] == [int(x) for x in gold[row] if x >= 0], row


def test_device_sampler_epoch_runs(golden):
    """LLMREC_DEVICE_SAMPLER=1: one epoch trains and evaluates without the host sampler."""
    import os
    os.environ["LLMREC_DEVICE_SAMPLER"] = "1"
    try:
        m = load_dropin(golden_argv(golden))
        m.set_seed(1)
        tr = m.Trainer(data_config={})
        u, p, n = tr.sample_batch()
        assert u.is_cuda and u.numel() >= golden.args["batch_size"]
        loss, _, _ = tr.train_step(u, p, n)
        assert np.isfinite(float(loss))
    finally:
        os.environ.pop("LLMREC_DEVICE_SAMPLER", None)


def test_in_graph_sampler_steps_run_and_learn(golden, monkeypatch):
    """LLMREC_DEVICE_SAMPLER=1 + LLMREC_GRAPH=1: a step is one graph replay that samples its own batch
    (llmrec_sample_batch with the device step counter). Batches differ step to step, the loss is finite
    and the parameters move."""
    monkeypatch.setenv("LLMREC_DEVICE_SAMPLER", "1"); monkeypatch.setenv("LLMREC_GRAPH", "1"); monkeypatch.setenv("LLMREC_FUSED", "1")
    m = load_dropin(golden_argv(golden))
    m.set_seed(1)
    tr = m.Trainer(data_config={})
    before = tr.model_mm.item_id_embedding.weight.detach().clone()
    seen = []
    for _ in range(4):
        loss, mf, emb = tr.train_step_sampled()
        assert np.isfinite(float(loss)) and float(mf) > 0
        st = tr._fused_step().static
        nv = int(st["n_valid"])
        assert golden.args["batch_size"] <= nv <= tr._fused_step().b_max
        seen.append(st["users"][:nv].cpu().clone())
    assert int(tr._device_batcher().step_dev) == 4              # the capture call runs one real (warm-up) step, recording does not execute
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert not torch.equal(before, tr.model_mm.item_id_embedding.weight.detach())


def test_in_graph_sampler_epoch_sums_on_the_device(golden, monkeypatch):
    """Trainer.train_epoch_sampled: n steps as graph replays (graphs of four + single steps) with the logged scalars summed in double
    inside the graph - equal to the sum of the per-step scalars of the same steps run one replay at a time (same seed: the same
    batches), and the device step counter advanced by exactly n."""
    monkeypatch.setenv("LLMREC_DEVICE_SAMPLER", "1"); monkeypatch.setenv("LLMREC_GRAPH", "1"); monkeypatch.setenv("LLMREC_FUSED", "1")
    n = 7
    m = load_dropin(golden_argv(golden)); m.set_seed(1)
    tr = m.Trainer(data_config={})
    sums = tr.train_epoch_sampled(n).cpu().numpy()
    assert int(tr._device_batcher().step_dev) == n and tr._fused_step().graph_unroll == 4
    m2 = load_dropin(golden_argv(golden)); m2.set_seed(1)
    tr2 = m2.Trainer(data_config={})
    ref = np.zeros(3)
    for _ in range(n):
        ref += np.array([float(x) for x in tr2.train_step_sampled()], dtype=np.float64)
    assert np.all(np.isfinite(sums)) and np.allclose(sums, ref, rtol=2e-5, atol=0), (sums, ref)
    a, b = tr.model_mm.item_id_embedding.weight.detach(), tr2.model_mm.item_id_embedding.weight.detach()
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # a second epoch reuses the captured graphs and starts its sums from zero
    sums2 = tr.train_epoch_sampled(5).cpu().numpy()
    assert int(tr._device_batcher().step_dev) == n + 5 and np.all(sums2 < sums) and np.all(sums2 > 0)


def test_sharded_trainer_single_rank_matches_oracle():
    """llmrec_amd/dist.py on the HIP backend with a world of one rank (collectives are identities):
    the two-pass sharded BPR and the sharded SpMM operands must reproduce the oracle's steps."""
    import scipy.sparse as sp
    from oracle import oracle as O
    from llmrec_amd import dist as ld
    rng = np.random.default_rng(3)
    U, I, D, L, B, STEPS = 700, 500, 64, 2, 256, 3
    deg = rng.integers(1, 40, size=U); deg[5] = 300
    rows = np.repeat(np.arange(U), deg)
    cols = np.concatenate([rng.choice(I, size=c, replace=False) for c in deg])
    u_tab = (rng.standard_normal((U, D)) * 0.1).astype(np.float32)
    i_tab = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    batches = [(rng.integers(0, U, size=B), rng.integers(0, I, size=B), rng.integers(0, I, size=B)) for _ in range(STEPS)]
    # oracle
    R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(U, I))
    a_ui, a_iu = O.normalized_graphs(R)
    pu = torch.tensor(u_tab, requires_grad=True); pi = torch.tensor(i_tab, requires_grad=True)
    opt = torch.optim.AdamW([{"params": [pu, pi]}], lr=1e-3)
    cfg = O.Config(batch_size=B, decay=1e-5, prune_loss_drop_rate=0.71)
    ref_losses = []
    for us, ps, ns in batches:
        u, i = pu, pi
        ul, il = [u], [i]
        for l in range(L):
            u = torch.sparse.mm(a_ui, i)
            if l == L - 1: u = torch.softmax(u, -1)
            i = torch.sparse.mm(a_iu, u)
            if l == L - 1: i = torch.softmax(i, -1)
            ul.append(u); il.append(i)
        eu, ei = torch.mean(torch.stack(ul), 0), torch.mean(torch.stack(il), 0)
        mf, emb = O.bpr_loss(eu[torch.tensor(us)], ei[torch.tensor(ps)], ei[torch.tensor(ns)], cfg)
        opt.zero_grad(); (mf + emb).backward(); opt.step()
        ref_losses.append(float(mf + emb))
    # HIP, sharded code path, one rank
    comm, be = ld.Comm(), ld.HipBackend()
    g = ld.ShardedGraph.build(torch.tensor(rows).cuda(), torch.tensor(cols).cuda(), U, I, 0, comm, be)
    model = ld.ShardedIDModel(g, comm, be, D, L, U, seed=1)
    with torch.no_grad():
        model.user_id_embedding.copy_(torch.tensor(u_tab)); model.item_id_embedding.copy_(torch.tensor(i_tab))
    tr = ld.ShardedTrainer(model, 1e-3, B, 0.71, 1e-5, seed=1)
    for (us, ps, ns), want in zip(batches, ref_losses):
        loss, _ = tr.step((torch.tensor(us).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda()))
        assert abs(float(loss) - want) <= 1e-5 * abs(want)
    assert rel(model.user_id_embedding.detach().cpu().numpy(), pu.detach().numpy()) < RTOL
    assert rel(model.item_id_embedding.detach().cpu().numpy(), pi.detach().numpy()) < RTOL
    # the device sampler path of the trainer
    loss, _ = tr.step()
    assert np.isfinite(float(loss))


def test_sharded_full_model_single_rank_matches_oracle_and_eval():
    """llmrec_amd/dist.ShardedMMModel + ShardedMMTrainer + sharded_evaluate on the HIP backend with
    one rank, against the single-process oracle (same problem as the 2-rank gloo test)."""
    from tests.test_dist_cpu import _mm_problem, _mm_oracle_run, MM_KEYS, U, I, D, L, B_LOCAL, DROP, DECAY, LR
    from oracle import oracle as O
    from llmrec_amd import dist as ld
    from llmrec_amd.engine import Hyper
    ref_params, ref_losses, ref_eu, ref_ei, rows, cols = _mm_oracle_run()
    _, _, u_tab, i_tab, batches, feats, lin = _mm_problem()
    comm, be = ld.Comm(), ld.HipBackend()
    g = ld.ShardedGraph.build(torch.tensor(rows).cuda(), torch.tensor(cols).cuda(), U, I, 0, comm, be)
    item_feats = {k: torch.tensor(v).cuda() for k, v in feats.items() if k != "user"}
    model = ld.ShardedMMModel(g, comm, be, D, L, U, item_feats, torch.tensor(feats["user"]).cuda(), MM_KEYS, (0.02, 2.8, 0.005), seed=3)
    with torch.no_grad():
        for name, v in lin.items():
            mod, attr = name.split(".")
            getattr(getattr(model, mod), attr).copy_(torch.tensor(v))
        model.user_id_embedding.copy_(torch.tensor(u_tab)); model.item_id_embedding.copy_(torch.tensor(i_tab))
    hp = Hyper(batch_size=2 * B_LOCAL, decay=DECAY, prune_loss_drop_rate=DROP); hp.lr = LR
    tr = ld.ShardedMMTrainer(model, hp, 2 * B_LOCAL, I)
    for per_rank, want in zip(batches, ref_losses):
        us = np.concatenate([b[0] for b in per_rank]); ps = np.concatenate([b[1] for b in per_rank]); ns = np.concatenate([b[2] for b in per_rank])
        loss, _ = tr.step(torch.tensor(us).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda())
        assert abs(float(loss) - want) <= 2e-5 * abs(want)
    params = dict(model.named_parameters())
    assert rel(params["item_id_embedding"].detach().cpu().numpy(), ref_params["item_id_embedding.weight"]) < RTOL
    assert rel(params["user_id_embedding"].detach().cpu().numpy(), ref_params["user_id_embedding.weight"]) < RTOL
    assert rel(params["item_trans.weight"].detach().cpu().numpy(), ref_params["item_trans.weight"]) < RTOL
    # sharded evaluation == the oracle's evaluation of the same embeddings
    rng = np.random.default_rng(0)
    test_set = {u: sorted(rng.choice(I, size=2, replace=False).tolist()) for u in range(0, U, 2)}
    train_items = {u: cols[rows == u].tolist() for u in range(U)}
    trows = np.concatenate([np.full(len(v), u) for u, v in test_set.items()]); tcols = np.concatenate([v for v in test_set.values()])
    trp, tci, _ = be.ops.csr_from_coo(torch.tensor(trows).cuda(), torch.tensor(tcols).cuda(), None, U, I)
    got = ld.sharded_evaluate(model, comm, be, g, trp, tci, len(test_set), (10, 20, 50))
    with torch.no_grad():
        fw = model()
    want, _ = O.evaluate(fw["E_u"].cpu().numpy(), fw["E_i"].cpu().numpy(), sorted(test_set), train_items, test_set, (10, 20, 50), batch_size=64)
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(got[k], want[k], rtol=0, atol=1e-12), k


def _dp_replica(golden, cls, b_max, **kw):
    """A fresh drop-in Trainer (seeded: identical initialisation) wrapped in the given step class."""
    from llmrec_amd import ops
    m = load_dropin(golden_argv(golden))
    m.set_seed(golden.args["seed"])
    tr = m.Trainer(data_config={})
    graph = type("G", (), {"ui": ops.operand_from_sparse_tensor(tr.ui_graph), "iu": ops.operand_from_sparse_tensor(tr.iu_graph)})
    a = m.args
    step = cls(tr.model_mm, graph, tr.hyper, (a.model_cat_rate, a.user_cat_rate, a.item_cat_rate), tr.optimizer, b_max, **kw)
    return tr, step


@pytest.mark.parametrize("use_graph", [False, True])
def test_data_parallel_replicas_match_single_step_on_concatenated_batch(golden, use_graph):
    """llmrec_amd/dp.py: two batch-sharded replicas on ONE GPU, exchanging through a loop-back
    (gather blocks concatenated, gradient buckets summed by hand), must reproduce FusedStep on the
    concatenated batch - global prune threshold, global regulariser norms, feature regulariser
    counted once. The ranks hold different numbers of valid samples (device-side n_valid)."""
    from llmrec_amd.fused import FusedStep
    from llmrec_amd.dp import DataParallelStep
    W, STEPS = 2, 3
    z = golden.z
    cap = max(z["step%d/users" % s].size for s in range(golden.n_steps))
    cap = (cap + 1) // 2 + 3                                             # slots per rank (> what it needs)
    tr1, single = _dp_replica(golden, FusedStep, W * cap)
    reps = [_dp_replica(golden, DataParallelStep, cap, rank=r, world=W) for r in range(W)]
    for s in range(STEPS):
        u, p, n = (torch.tensor(z["step%d/%s" % (s, k)]).cuda() for k in ("users", "pos", "neg"))
        B = u.numel()
        cut = B // 2 - 1                                                 # rank 0 gets fewer samples than rank 1
        parts = [(0, cut), (cut, B)]
        want = [float(x) for x in single.step_eager(u, p, n)]
        loads = []
        for (lo, hi) in parts:
            pad = cap - (hi - lo)
            z64 = torch.zeros(pad, dtype=torch.int64, device="cuda")
            loads.append((torch.cat([u[lo:hi], z64]), torch.cat([p[lo:hi], z64]), torch.cat([n[lo:hi], z64]),
                          torch.tensor([hi - lo], dtype=torch.int32, device="cuda")))
        if use_graph and s == 0:
            # capture() runs one real (eager) step per replica; the loop-back needs the phases interleaved,
            # so capture on a throw-away exchange first and restore the state afterwards
            for (tr, st), ld in zip(reps, loads):
                snap = {k: v.detach().clone() for k, v in tr.model_mm.state_dict().items()}
                st.capture(*ld)
                tr.model_mm.load_state_dict(snap)
                for pstate in st.opt.state.values():
                    pstate[0].zero_(); pstate[1].zero_()
                if st.opt.dev_state is not None:
                    st.opt.dev_state.zero_()
        for (tr, st), ld in zip(reps, loads):
            if use_graph:
                st._load(*ld); st.graphs[0].replay()
            else:
                st.phase_a(*ld)
        g_all = torch.cat([st.g_local for _, st in reps])
        for (tr, st), ld in zip(reps, loads):
            st.g_all.copy_(g_all)
            if use_graph:
                st.graphs[1].replay()
            else:
                st.phase_b(*ld)
        total = reps[0][1].bucket + reps[1][1].bucket
        for (tr, st) in reps:
            st.bucket.copy_(total)
            if use_graph:
                st.graphs[2].replay()
            else:
                st.phase_c()
        for (tr, st) in reps:
            got = [float(st.scal[1]), float(st.scal[2]), float(st.scal[3])]
            for a_, b_ in zip(got, want):
                assert abs(a_ - b_) <= 2e-5 * abs(b_), (s, got, want)
    ref = dict(tr1.model_mm.named_parameters())
    for r, (tr, st) in enumerate(reps):
        mine = dict(tr.model_mm.named_parameters())
        for nm in TRAINABLE:
            e = rel(mine[nm].detach().cpu().numpy(), ref[nm].detach().cpu().numpy())
            assert e < RTOL, (r, nm, e)
    # replicas stay bit-identical
    a_, b_ = dict(reps[0][0].model_mm.named_parameters()), dict(reps[1][0].model_mm.named_parameters())
    for nm in TRAINABLE:
        assert torch.equal(a_[nm], b_[nm]), nm


def test_device_state_is_built_once_per_device(golden):
    """Data.device_state: "cuda" and "cuda:<current>" are one device - the cached CSRs are returned, not rebuilt at every other call (round 4:
    Trainer.test() in graph mode spent 80 ms per evaluation rebuilding them, and captured a new evaluation graph each time)."""
    m = load_dropin(golden_argv(golden))
    dg = m.data_generator
    a = dg.device_state(torch.device("cuda"))
    b = dg.device_state(torch.device("cuda", torch.cuda.current_device()))
    c = dg.device_state("cuda")
    assert a is b and b is c and a["train"] is c["train"]
    tr = m.Trainer(data_config={})
    users = list(dg.test_set.keys())
    tr.test(users, is_val=False); tr.test(users, is_val=False)
    fused = tr._fused_step()
    if fused:
        assert len(fused._eval_graphs) == 1                       # one captured evaluation, replayed


def test_weight_gradient_geometry_is_relaid_and_the_graph_recaptured_without_an_extra_step(golden, monkeypatch):
    """FusedStep.check_wgrad_geometry: when the row list's length has left the range the captured launch was laid out for, the
    launch is laid out again and the step graph RE-captured with no warm-up step - the trajectory goes on exactly where it was: every later
    step's 8 (mf, emb) pairs and the final parameters still match the reference's golden vectors."""
    monkeypatch.setenv("LLMREC_FUSED", "1"); monkeypatch.setenv("LLMREC_GRAPH", "1")
    m = load_dropin(golden_argv(golden))
    m.set_seed(golden.args["seed"])
    tr = m.Trainer(data_config={})
    bpr_log = []
    tr._on_bpr = lambda mf, emb: bpr_log.append((mf.detach(), emb.detach()))
    recaptured = 0
    for s in range(golden.n_steps):
        users, pos, neg = (torch.tensor(golden.z["step%d/%s" % (s, n)]).cuda() for n in ("users", "pos", "neg"))
        bpr_log.clear()
        tr.train_step(users, pos, neg)
        got = np.array([[float(a), float(b)] for a, b in bpr_log])
        gold = golden.z["step%d/bpr" % s]
        assert np.abs(got - gold).max() <= RTOL * np.abs(gold).max(), (s, recaptured)
        f = tr._fused_step()
        if s == 1 and f.wgrad_rows:
            assert f.graph_exec is not None and f.act_expected is not None
            assert f.check_wgrad_geometry() is False                     # the list length is what the launch was laid out for
            old_graph = f.graph_exec
            f.act_expected = f.act_expected * 4 + 1000                   # pretend the launch had been laid out for a far longer list
            assert f.check_wgrad_geometry() is True
            assert f.graph_exec is not old_graph and f.act_expected < 4 * f.U
            recaptured += 1
    params = dict(tr.model_mm.named_parameters())
    last = max(golden.detail_steps())
    if last == golden.n_steps - 1:
        for nm in TRAINABLE:
            e = rel(params[nm].detach().cpu().numpy(), golden.z["step%d/param/%s" % (last, nm)])
            assert e < RTOL, (nm, e)
    if tr._fused_step().wgrad_rows:
        assert recaptured == 1
