"""GPU parity at the shapes bench.py TIMES (BASELINE.json configs[1], Netflix shape U=13187, I=17366):
the kernel instantiations the small op tests cannot reach -

* ``linear_fwd_grouped_bf16x3_kernel<4,2,2>`` / ``linear_fwd_grouped_kernel<4,2,2>``: every K % 32 == 0 and
  >= 256 work units of 128 rows (llmrec_amd/csrc/dense.hip, linear_fwd_grouped_impl);
* ``linear_wgrad_bf16x3_kernel`` at M = 5 x 17366 / 13187, K = 1536 / 768 / 512: XCD-ordered slabs, the 4-slab
  LDS pre-reduction, ragged slab tails, dY as a column slice of the [I, 7d] buffer (ld = 448);
* the scoring + top-K kernel at I = 17366 (about five buffer drains per user) and at I = 10^6, with
  adversarial score orders;
* one whole fused + graph-replayed training step and one evaluation at the bench's exact shape against the oracle
  (reference Models.py:145-199, main.py:228-278, utility/batch_test.py:21-36).

References are fp64 products (torch) or the CPU oracle; tolerances are written at each assert."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as O

DEV = "cuda"
U_NF, I_NF, D = 13187, 17366, 64


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from llmrec_amd import ops as _ops
    return _ops


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


# ------------------------------------------------------------------------------------------
# R4: the grouped projection exactly as FusedStep._project_all launches it
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,tol", [("bf16x3", 3e-6), ("f32", 3e-6)])
def test_grouped_projection_at_bench_shape(ops, precision, tol):
    g = torch.Generator(device=DEV); g.manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    P_cat = torch.full((I_NF, 7 * D), 7.0, device=DEV)              # the 7 item-side streams as one [I, 7d] buffer
    P_usr = torch.full((U_NF, D), 7.0, device=DEV)
    shapes = [(I_NF, 1536)] * 5 + [(U_NF, 1536), (I_NF, 768), (I_NF, 512)]       # longest K first, as the step sorts them
    outs = [P_cat[:, (2 + k) * D:(3 + k) * D] for k in range(5)] + [P_usr, P_cat[:, D:2 * D], P_cat[:, 0:D]]
    W_item, b_item = rn(D, 1536) / 39.0, rn(D)
    jobs = []
    for i, ((M, K), out) in enumerate(zip(shapes, outs)):
        W, b = (W_item, b_item) if i < 5 else (rn(D, K) / K ** 0.5, rn(D))      # ONE shared item_trans for the 5 attribute keys
        jobs.append((rn(M, K), W, b, out))
    assert sum((M + 127) // 128 for M, _ in shapes) >= 256 and all(K % 32 == 0 for _, K in shapes)   # -> the <4, 2, 2> instantiation
    ops.linear_fwd_grouped(jobs, D, precision=precision)
    for X, W, b, out in jobs:
        want = X.double() @ W.double().t() + b.double()
        e = relmax(out, want)
        assert e < tol, (precision, tuple(X.shape), e)
    # ragged K (not a multiple of 32) in the same launch geometry -> the <4, 2, 1> instantiation
    jobs2 = [(rn(I_NF, 1540), rn(D, 1540) / 39.0, rn(D), torch.empty(I_NF, D, device=DEV)) for _ in range(2)] + jobs[5:]
    ops.linear_fwd_grouped(jobs2, D, precision=precision)
    for X, W, b, out in jobs2[:2]:
        assert relmax(out, X.double() @ W.double().t() + b.double()) < tol


# ------------------------------------------------------------------------------------------
# R4: the step's four weight-gradient launches
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,tol", [("bf16x3", 3e-6), ("f32", 3e-6)])
def test_weight_gradients_at_bench_shape(ops, precision, tol):
    g = torch.Generator(device=DEV); g.manual_seed(12)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    dP_cat = rn(I_NF, 7 * D)                                       # dY of stream s = columns [s d, (s+1) d): ld = 448
    dP_usr = rn(U_NF, D)
    feats = [rn(I_NF, 1536) for _ in range(5)]
    cases = [("item_trans x5", [(dP_cat[:, (2 + k) * D:(3 + k) * D], feats[k]) for k in range(5)], 1536),
             ("user_trans", [(dP_usr, rn(U_NF, 1536))], 1536),
             ("text_trans", [(dP_cat[:, D:2 * D], rn(I_NF, 768))], 768),
             ("image_trans", [(dP_cat[:, 0:D], rn(I_NF, 512))], 512)]
    for name, pairs, K in cases:
        dW = torch.full((D, K), 7.0, device=DEV); db = torch.full((D,), 7.0, device=DEV)
        ops.linear_wgrad_grouped(pairs, dW, db, False, precision=precision)
        want = sum(dy.double().t() @ x.double() for dy, x in pairs)
        want_b = sum(dy.double().sum(0) for dy, _ in pairs)
        e, eb = relmax(dW, want), relmax(db, want_b)
        assert e < tol and eb < tol, (precision, name, e, eb)
        dW2 = dW.clone()
        ops.linear_wgrad_grouped(pairs, dW2, db, True, precision=precision)                      # accumulate
        assert relmax(dW2, 2 * want) < 2 * tol, (precision, name)
        ops.linear_wgrad_grouped(pairs, dW2, db, False, precision=precision)                     # deterministic (fixed reduction tree)
        assert torch.equal(dW2, dW), (precision, name)


def test_multi_target_weight_gradient_at_bench_shape(ops):
    """llmrec_linear_wgrad_multi_bf16x3: item_trans' (5 pairs), text's and image's gradients in one launch - each target against
    the fp64 product (3e-6), against its single-target launch (other slab length: equal up to rounding), accumulate,
    determinism, ragged problem sizes, and the refusal of shapes outside the fast path."""
    g = torch.Generator(device=DEV); g.manual_seed(13)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    for n_items in (I_NF, 1000, 33):
        dP_cat = rn(n_items, 7 * D)
        feats = [rn(n_items, 1536) for _ in range(5)]
        text, image = rn(n_items, 768), rn(n_items, 512)
        pairs = [[(dP_cat[:, (2 + k) * D:(3 + k) * D], feats[k]) for k in range(5)], [(dP_cat[:, D:2 * D], text)], [(dP_cat[:, 0:D], image)]]
        Ks = [1536, 768, 512]
        dWs = [torch.full((D, K), 7.0, device=DEV) for K in Ks]
        dbs = [torch.full((D,), 7.0, device=DEV) for _ in Ks]
        targets = [(pairs[t], dWs[t], dbs[t], False) for t in range(3)]
        assert ops.linear_wgrad_multi_workspace(targets) > 0
        ops.linear_wgrad_multi(targets)
        for t in range(3):
            want = sum(dy.double().t() @ x.double() for dy, x in pairs[t])
            want_b = sum(dy.double().sum(0) for dy, _ in pairs[t])
            e, eb = relmax(dWs[t], want), relmax(dbs[t], want_b)
            assert e < 3e-6 and eb < 3e-6, (n_items, t, e, eb)
            single = torch.empty_like(dWs[t]); single_b = torch.empty_like(dbs[t])
            ops.linear_wgrad_grouped(pairs[t], single, single_b, False, precision="bf16x3")
            assert relmax(dWs[t], single.double()) < 3e-6, (n_items, t)
        first = [w.clone() for w in dWs]
        ops.linear_wgrad_multi([(pairs[t], dWs[t], dbs[t], True) for t in range(3)])           # accumulate
        for t in range(3):
            assert relmax(dWs[t], 2 * first[t].double()) < 1e-6, (n_items, t)
        ops.linear_wgrad_multi(targets)                                                          # deterministic
        for t in range(3):
            assert torch.equal(dWs[t], first[t]), (n_items, t)
    bad = [([(rn(64, D), rn(64, 100))], torch.empty(D, 100, device=DEV), None, False)]          # K % 64 != 0
    assert ops.linear_wgrad_multi_workspace(bad) == -1
    with pytest.raises(RuntimeError):
        ops.linear_wgrad_multi(bad)


# ------------------------------------------------------------------------------------------
# R9: scoring + masked top-K at Netflix width and at 10^6 items, adversarial orders
# ------------------------------------------------------------------------------------------
def _check_topk(ops, eu, ei, q, train_rows, K, what):
    """Lists against oracle.rank_topk_np on the kernel's own (bit-exact, separately tested) scores."""
    from llmrec_amd.ops import Csr, SpmmPlan
    n_items = ei.shape[0]
    rp = np.zeros(eu.shape[0] + 1, dtype=np.int64)
    for u, items in train_rows.items():
        rp[u + 1] = len(items)
    rp = np.cumsum(rp)
    ci = np.concatenate([np.sort(np.asarray(train_rows[u], dtype=np.int64)) for u in sorted(train_rows)]) if train_rows else np.zeros(0, dtype=np.int64)
    train = Csr(eu.shape[0], n_items, torch.tensor(rp, dtype=torch.int32, device=DEV), torch.tensor(ci, dtype=torch.int32, device=DEV),
                None, None, None, SpmmPlan())
    qd = torch.tensor(q, dtype=torch.int64, device=DEV)
    idx, sc = ops.score_topk(eu, ei, qd, train, K)
    S = ops.scores(eu, ei, qd).cpu().numpy()
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for r, u in enumerate(q):
        want = O.rank_topk_np(S[r], train_rows.get(u, []), K)
        got = idx[r][idx[r] >= 0]
        assert got.tolist() == want.tolist(), (what, u, got[:8], want[:8])
        assert np.array_equal(sc[r][: len(want)], S[r][want]), (what, u)
        assert (idx[r][len(want):] == -1).all(), (what, u)


@pytest.mark.parametrize("K", [50, 64])
def test_topk_netflix_width_random_and_adversarial(ops, K):
    rng = np.random.default_rng(K)
    n_users, n_items = 96, I_NF
    eu = torch.tensor(rng.standard_normal((n_users, D)).astype(np.float32), device=DEV)
    ei_np = rng.standard_normal((n_items, D)).astype(np.float32)
    train = {u: rng.choice(n_items, size=int(rng.integers(0, 60)), replace=False).tolist() for u in range(n_users)}
    train[3] = rng.choice(n_items, size=n_items - (K - 1), replace=False).tolist()      # fewer than K candidates left
    train[4] = list(range(n_items))                                                       # nothing left
    q = list(range(n_users))
    _check_topk(ops, eu, torch.tensor(ei_np, device=DEV), q, train, K, "random")
    # scores strictly increasing with the item id for every user: each candidate beats the running K-th, the filter
    # never rejects, every round appends - maximal buffer drains
    eu1 = torch.zeros(n_users, D, device=DEV); eu1[:, 0] = torch.tensor(rng.uniform(0.5, 2.0, n_users).astype(np.float32))
    ei1 = torch.zeros(n_items, D, device=DEV); ei1[:, 0] = torch.arange(n_items, device=DEV, dtype=torch.float32) / 8.0
    _check_topk(ops, eu1, ei1, q, train, K, "ascending")
    # strictly decreasing: the first K items win and nothing after passes (the opposite extreme)
    ei2 = ei1.clone(); ei2[:, 0] = -ei1[:, 0]
    _check_topk(ops, eu1, ei2, q, train, K, "descending")
    # all scores equal: the tie rule alone (ascending item id) decides
    ei3 = torch.zeros(n_items, D, device=DEV); ei3[:, 1] = 1.0
    eu3 = torch.zeros(n_users, D, device=DEV); eu3[:, 1] = 0.25
    _check_topk(ops, eu3, ei3, q, train, K, "all-equal")
    # few distinct values: long runs of exact ties across tile and wave boundaries
    ei4 = torch.zeros(n_items, D, device=DEV); ei4[:, 0] = torch.tensor(rng.integers(0, 4, n_items).astype(np.float32), device=DEV)
    _check_topk(ops, eu1, ei4, q, train, K, "four-values")


def test_topk_one_million_items(ops):
    rng = np.random.default_rng(7)
    n_users, n_items, K = 40, 1_000_000, 50
    eu = torch.tensor(rng.standard_normal((n_users, D)).astype(np.float32), device=DEV)
    ei = torch.randn(n_items, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    train = {u: rng.choice(n_items, size=int(rng.integers(1, 3000)), replace=False).tolist() for u in range(n_users)}
    q = list(range(0, n_users, 3)) + [1, 1]                                               # repeated query user
    _check_topk(ops, eu, ei, q, train, K, "1M random")
    eu1 = torch.zeros(n_users, D, device=DEV); eu1[:, 5] = 1.0
    ei1 = torch.zeros(n_items, D, device=DEV); ei1[:, 5] = torch.arange(n_items, device=DEV, dtype=torch.float32)
    _check_topk(ops, eu1, ei1, q[:6], train, K, "1M ascending")


# ------------------------------------------------------------------------------------------
# the whole step and evaluation at the bench's shape (fused + HIP graph replay) against the oracle
# ------------------------------------------------------------------------------------------
def test_fused_graph_step_and_eval_at_netflix_shape_match_oracle():
    import bench
    dev = torch.device("cuda")
    w = bench.NetflixShaped("nf", 0, dev)
    rep = bench.parity_check(w, n_eval_users=192)
    print(rep)
    assert rep["forward_max_rel"] < 1e-4 and rep["bpr_max_rel"] < 1e-4 and rep["loss_rel"] < 1e-4, rep
    assert rep["grad_max_rel"] < 1e-4 and rep["adamw_given_gpu_grads_max_rel"] < 1e-5 and rep["embeddings_after_steps_max_rel"] < 1e-4, rep
    assert rep["topk_lists_equal"] == rep["topk_lists_checked"] > 0, rep
    assert rep["ok"], rep


def test_fused_graph_step_and_eval_at_movielens_shape_match_oracle():
    """BASELINE.json configs[2]: MovieLens shape (U = 12495, I = 10322), THREE propagation layers, text + visual fusion,
    prune loss - the fused + graph-replayed step and one evaluation against the oracle (reference Models.py:169-186 with
    --weight_size [64,64,64], main.py:228-278)."""
    import bench
    w = bench.NetflixShaped("ml", 0, torch.device("cuda"))
    assert w.fused.L == 3 and (w.sh.n_users, w.sh.n_items) == (12495, 10322)
    rep = bench.parity_check(w, n_eval_users=192)
    print(rep)
    assert rep["forward_max_rel"] < 1e-4 and rep["bpr_max_rel"] < 1e-4 and rep["loss_rel"] < 1e-4, rep       # stage-wise
    assert rep["grad_max_rel"] < 1e-4 and rep["adamw_given_gpu_grads_max_rel"] < 1e-5, rep
    assert rep["embeddings_after_steps_max_rel"] < 1e-4, rep                                                 # end to end, after 2 AdamW steps
    assert rep["topk_lists_equal"] == rep["topk_lists_checked"] > 0, rep
    assert rep["ok"], rep


def test_multi_step_graph_equals_single_step_replays():
    """FusedStep.run_steps: graphs of four steps + single-step graphs for the remainder run the same steps as single replays - the
    device sampler's counter, AdamW's step counter and the row stamps all advance inside the graph. Seven steps each way from
    identical seeds: the sampled batch of the last step and the optimiser's step counter are identical, parameters agree to the rounding
    of the backward's atomic adds (as two runs of either form do)."""
    import bench
    dev = torch.device("cuda")
    res = []
    for multi in (False, True):
        w = bench.NetflixShaped("ml", 0, dev)
        w.UNROLL = 4
        if multi:
            w.run_steps(7)                                        # capture (1 step) + one 4-step graph + 2 single replays
            assert w.fused.graph_multi is not None and w.fused.graph_unroll == 4
        else:
            for _ in range(7):
                w.step()
        torch.cuda.synchronize()
        st = w.fused.static
        res.append(([st[k].clone() for k in ("users", "pos", "neg")], [p.detach().clone() for p in w.opt.params],
                    int(w.opt.dev_state[:1].view(torch.int32)[0]), [float(x) for x in w.fused.scal[1:4]]))
    (b0, p0, t0, l0), (b1, p1, t1, l1) = res
    assert all(torch.equal(x, y) for x, y in zip(b0, b1)) and t0 == t1 == 7
    for x, y in zip(p0, p1):
        assert float((x - y).abs().max()) <= 2e-5 * float(x.abs().max()) + 1e-9      # (Adam turns atomic-order rounding of tiny gradients into ~1e-3 lr)
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(l0, l1))


# ------------------------------------------------------------------------------------------
# R2 in its HBM-bound regime: >= 10 M edges, d = 64 and d = 128 (BASELINE.json configs[3] / [4] operand widths), every
# direction the steps run, sampled rows + the longest rows against fp64
# ------------------------------------------------------------------------------------------
def _expected_rows(csr, X, rows_sel):
    """fp64 rows of diag(row_scale) P diag(col_scale) X for the listed rows (device, vectorised)."""
    rp = csr.rowptr.long()
    starts, lens = rp[rows_sel], rp[rows_sel + 1] - rp[rows_sel]
    owner = torch.repeat_interleave(torch.arange(rows_sel.numel(), device=X.device), lens)
    offs = torch.arange(int(lens.sum()), device=X.device) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
    cols = csr.colidx.long()[torch.repeat_interleave(starts, lens) + offs]
    vals = X[cols].double()
    if csr.col_scale is not None:
        vals = vals * csr.col_scale[cols].double()[:, None]
    out = torch.zeros(rows_sel.numel(), X.shape[1], dtype=torch.float64, device=X.device)
    out.index_add_(0, owner, vals)
    if csr.row_scale is not None:
        out = out * csr.row_scale[rows_sel].double()[:, None]
    return out


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_at_10M_edges_sampled_rows_match_fp64(ops, d):
    from llmrec_amd import synth
    n_users, n_items, n_edges = 600_000, 300_000, 10_000_000
    rows, cols = synth.bipartite_edges_device(n_users, n_items, n_edges, 3, DEV)
    assert rows.numel() == n_edges                                        # the generator's count is exact
    g = ops.BipartiteGraph.from_edges(rows, cols, n_users, n_items)
    assert g.ui.fwd.nnz >= ops.SPMM_LATENCY_NNZ                           # the throughput policy (128 / 512 nnz pieces), not the Netflix one
    gen = torch.Generator(device=DEV); gen.manual_seed(d)
    Xi = torch.randn(n_items, d, generator=gen, device=DEV); Xu = torch.randn(n_users, d, generator=gen, device=DEV)
    for name, a, X in (("ui.fwd", g.ui.fwd, Xi), ("iu.fwd", g.iu.fwd, Xu), ("ui.bwd (per-edge weight)", g.ui.bwd, Xu), ("iu.bwd (per-edge weight)", g.iu.bwd, Xi)):
        Y = ops.spmm_raw(a, X)
        deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
        sel = torch.unique(torch.cat([torch.randint(0, a.n_rows, (3000,), generator=gen, device=DEV), torch.topk(deg, 8).indices,
                                      torch.nonzero(deg == 0)[:4].reshape(-1), torch.tensor([0, a.n_rows - 1], device=DEV)]))
        want = _expected_rows(a, X, sel)
        e = float((Y[sel].double() - want).abs().max() / want.abs().max())
        assert e < 1e-5, (name, d, e)                                    # fp32 summation of up to 10^4 terms per row
        assert torch.equal(Y, ops.spmm_raw(a, X)), name                   # deterministic at this size too
    # the forms the row-sharded step runs: softmax epilogue on the finished row, and the PATTERN operand gathering from a
    # pre-scaled tensor with an output scale (A^T g = R (s . g), llmrec_amd/dist_fused.py)
    a = g.ui.fwd
    Y = ops.spmm_raw(a, Xi, epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX))
    sel = torch.randint(0, a.n_rows, (2000,), generator=gen, device=DEV)
    want = torch.softmax(_expected_rows(a, Xi, sel), dim=-1)
    assert float((Y[sel].double() - want).abs().max() / want.abs().max()) < 3e-6
    pat = ops.Csr(a.n_rows, a.n_cols, a.rowptr, a.colidx, None, None, None, a.plans)
    Z = torch.randn(n_users, d, generator=gen, device=DEV)
    Y = ops.spmm_raw(pat, Xi * g.s_i[:, None], epilogue=ops.spmm_epilogue(ops.EPI_NONE, 0.25, Z, None, g.s_u))
    want = (0.25 * Z[sel].double() + _expected_rows(pat, Xi * g.s_i[:, None], sel)) * g.s_u[sel].double()[:, None]
    assert float((Y[sel].double() - want).abs().max() / want.abs().max()) < 3e-6


@pytest.mark.parametrize("n_users", [U_NF, 300, 4096 + 16 * 57])
def test_topk_split_left_over_tiles_equal_the_unsplit_sweep(ops, n_users):
    """llmrec_score_topk_ws_f32 re-lays the item table in MFMA fragment order, cuts the user tiles beyond the last full round of one
    tile per CU into item parts and merges their lists: the result is the workspace-free kernel's, bit for bit (ids and scores),
    including users whose train rows leave fewer than K candidates in a part; I_NF = 17366 is not a multiple of the 32-item tile
    (the packed table's padded last tile)."""
    from llmrec_amd import _lib
    from llmrec_amd.ops import _p, _ld
    g = torch.Generator(device=DEV); g.manual_seed(n_users)
    K = 50
    eu = torch.randn(n_users, D, generator=g, device=DEV)
    ei = torch.randn(I_NF, D, generator=g, device=DEV)
    ei[::7] = ei[3]                                                       # exact ties across the item parts
    q = torch.arange(n_users, dtype=torch.int64, device=DEV)
    deg = torch.randint(0, 12, (n_users,), generator=g, device=DEV)
    deg[-5] = I_NF - 20                                                   # fewer than K candidates in total
    rp = torch.zeros(n_users + 1, dtype=torch.int64, device=DEV); rp[1:] = torch.cumsum(deg, 0)
    ci = torch.empty(int(rp[-1]), dtype=torch.int32, device=DEV)
    rp_h, deg_h = rp.cpu().numpy(), deg.cpu().numpy()
    rng = np.random.default_rng(n_users)
    ci_h = np.empty(int(rp_h[-1]), dtype=np.int32)
    for u in range(n_users):
        if deg_h[u]:
            ci_h[rp_h[u]:rp_h[u + 1]] = np.sort(rng.choice(I_NF, size=int(deg_h[u]), replace=False))
    ci.copy_(torch.from_numpy(ci_h))
    rp32 = rp.to(torch.int32)
    need = _lib.query("llmrec_score_topk_workspace_bytes", n_users, I_NF, D)
    assert need > -(-I_NF // 32) * 2 * (D // 16) * 64 * 16, "these shapes leave user tiles over (more than the packed item table)"
    out = []
    for ws in (None, torch.empty(need, dtype=torch.uint8, device=DEV)):
        idx = torch.empty(n_users, K, dtype=torch.int32, device=DEV)
        sc = torch.empty(n_users, K, dtype=torch.float32, device=DEV)
        _lib.call("llmrec_score_topk_ws_f32", n_users, _p(q), _p(eu), _ld(eu), _p(ei), _ld(ei), I_NF, D, _p(rp32), _p(ci), K, _p(idx), _p(sc),
                  _p(ws), need if ws is not None else 0, None)
        torch.cuda.synchronize()
        out.append((idx, sc))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert int((out[0][0][n_users - 5] >= 0).sum()) == 20
    with pytest.raises(RuntimeError):                                      # a workspace that is too small is refused
        _lib.call("llmrec_score_topk_ws_f32", n_users, _p(q), _p(eu), _ld(eu), _p(ei), _ld(ei), I_NF, D, _p(rp32), _p(ci), K, _p(out[0][0]), _p(out[0][1]),
                  _p(torch.empty(need, dtype=torch.uint8, device=DEV)), need - 16, None)
