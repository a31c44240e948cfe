"""GPU parity: every HIP entry point (called through the C ABI via llmrec_amd.ops) against the
CPU oracle (oracle/oracle.py) or the ATen op the reference calls, on the same seeded inputs.
Tolerances: fp32 results 1e-4 relative (north_star) - asserted tighter where the arithmetic
allows; index lists bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import oracle as O


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from llmrec_amd import ops as _ops
    return _ops


DEV = "cuda"


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rand_graph(rng, n_rows, n_cols, degs):
    rows = np.repeat(np.arange(n_rows), degs)
    cols = np.concatenate([rng.choice(n_cols, size=d, replace=False) for d in degs]) if rows.size else np.zeros(0, dtype=np.int64)
    return rows.astype(np.int64), cols.astype(np.int64)


def degree_mix(rng, n_rows, n_cols):
    """Empty rows, short rows, rows just around the long-row / segment boundaries, hubs."""
    degs = rng.integers(0, 12, size=n_rows)
    special = [0, 1, 127, 128, 129, 255, 256, 257, 511, 513, min(n_cols, 1500)]
    for k, d in enumerate(special):
        degs[(k * 7) % n_rows] = min(d, n_cols)
    return degs


# ------------------------------------------------------------------------------------------
# R1 graph ingest
# ------------------------------------------------------------------------------------------
def test_csr_build_matches_scipy(ops):
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    n_rows, n_cols = 300, 2000
    rows, cols = rand_graph(rng, n_rows, n_cols, degree_mix(rng, n_rows, n_cols))
    perm = rng.permutation(rows.size)                    # unsorted COO in
    rows, cols = rows[perm], cols[perm]
    vals = rng.standard_normal(rows.size).astype(np.float32)
    rp, ci, v = ops.csr_from_coo(torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV), torch.from_numpy(vals).to(DEV), n_rows, n_cols)
    ref = sp.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))
    ref.sort_indices()
    assert np.array_equal(rp.cpu().numpy(), ref.indptr)
    assert np.array_equal(ci.cpu().numpy(), ref.indices)
    assert np.array_equal(v.cpu().numpy(), ref.data)
    # pattern-only, with duplicate edges kept, and the transpose via swapped arguments
    rows2 = np.concatenate([rows, rows[:50]]); cols2 = np.concatenate([cols, cols[:50]])
    rp2, ci2, _ = ops.csr_from_coo(torch.from_numpy(cols2).to(DEV), torch.from_numpy(rows2).to(DEV), None, n_cols, n_rows)
    order = np.lexsort((rows2, cols2))
    assert np.array_equal(ci2.cpu().numpy(), rows2[order])
    assert np.array_equal(rp2.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(cols2, minlength=n_cols))]))
    # empty graph
    rp3, ci3, _ = ops.csr_from_coo(torch.zeros(0, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV), None, 5, 5)
    assert rp3.cpu().tolist() == [0] * 6 and ci3.numel() == 0


def test_degree_scale_matches_reference_formula(ops):
    degs = np.array([0, 1, 2, 3, 7, 100, 12345], dtype=np.int64)
    rp = torch.tensor(np.concatenate([[0], np.cumsum(degs)]), dtype=torch.int32, device=DEV)
    got = ops.degree_scale(rp).cpu().numpy()
    want = np.power(degs.astype(np.float32) + 1e-8, -0.5).astype(np.float32)     # reference main.py:115-116 on fp32 sums
    want[degs == 0] = 0
    assert np.allclose(got, want, rtol=1.3e-7, atol=0)


# ------------------------------------------------------------------------------------------
# R2 SpMM forward / backward
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [64, 16, 128, 448, 20, 4])
def test_spmm_forward_backward_vs_torch_sparse(ops, d):
    rng = np.random.default_rng(d)
    n_rows, n_cols = 257, 1600
    rows, cols = rand_graph(rng, n_rows, n_cols, degree_mix(rng, n_rows, n_cols))
    deg = np.bincount(rows, minlength=n_rows)
    s = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0).astype(np.float32)
    vals = s[rows]                                           # the reference's diag(s) R
    A_cpu = torch.sparse_coo_tensor(torch.tensor(np.vstack([rows, cols])), torch.tensor(vals), (n_rows, n_cols))
    X_cpu = torch.tensor(rng.standard_normal((n_cols, d)).astype(np.float32), requires_grad=True)
    Y_cpu = torch.sparse.mm(A_cpu, X_cpu)
    G = torch.tensor(rng.standard_normal((n_rows, d)).astype(np.float32))
    Y_cpu.backward(G)

    A = A_cpu.to(DEV)
    op = ops.operand_from_sparse_tensor(A)
    assert op.fwd.val is None and op.fwd.row_scale is not None      # row-constant values detected
    assert op.fwd.plan.n_long >= 5
    X = X_cpu.detach().to(DEV).requires_grad_(True)
    Y = ops.spmm(A, X)
    Y.backward(G.to(DEV))
    assert rel_err(Y.detach().cpu(), Y_cpu.detach()) < 2e-6
    assert rel_err(X.grad.cpu(), X_cpu.grad) < 2e-6
    # deterministic run to run (no float atomics in the SpMM)
    assert torch.equal(ops.spmm(A, X).detach(), Y.detach())


def test_spmm_row_buckets_split_rows_and_epilogues(ops):
    """Every row bucket of the plan (lane group <= 32 nnz, wavefront <= 512, block <= 16384, split segments beyond) and the
    fused epilogues: Y = alpha Z + A X, row softmax (reference Models.py:176-177) and its backward, each against torch."""
    rng = np.random.default_rng(77)
    n_rows, n_cols = 600, 45000
    degs = rng.integers(0, 40, size=n_rows)
    for k, dg in enumerate([0, 1, 32, 33, 511, 512, 513, 4095, 4096, 4097, 16384, 16385, 20000, 44000]):
        degs[(k * 13 + 1) % n_rows] = dg
    rows, cols = rand_graph(rng, n_rows, n_cols, degs)
    deg = np.bincount(rows, minlength=n_rows)
    s = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0).astype(np.float32)
    A_cpu = torch.sparse_coo_tensor(torch.tensor(np.vstack([rows, cols])), torch.tensor(s[rows]), (n_rows, n_cols))
    A = A_cpu.to(DEV)
    op = ops.operand_from_sparse_tensor(A)
    sw, pl = op.fwd.plan_for(64)
    assert sw == 0 and (pl.t_wave, pl.t_block) == (128, 2048) and pl.n_wave >= 1 and pl.n_block >= 3 and pl.n_split == 7
    assert op.fwd.plan_for(448)[0] == 64                                              # wide operands go slice by slice
    for d in (64, 128, 448, 20):
        X_cpu = torch.tensor(rng.standard_normal((n_cols, d)).astype(np.float32))
        Z_cpu = torch.tensor(rng.standard_normal((n_rows, d)).astype(np.float32))
        want = torch.sparse.mm(A_cpu.double(), X_cpu.double()).float()     # rows of up to 44000 terms: fp64 reference
        X, Z = X_cpu.to(DEV), Z_cpu.to(DEV)
        Y = ops.spmm_raw(op.fwd, X)
        assert rel_err(Y.cpu(), want) < 4e-6, d
        for _ in range(3):
            assert torch.equal(ops.spmm_raw(op.fwd, X), Y)                   # deterministic, whichever wave finishes a row
        _, key = ops.spmm_shape(d, op.fwd.nnz)
        keep = op.fwd.plans[key]
        for alt in ((512, 16384, 16384), (32, 32, 64), (64, 256, 4096)):      # the HBM-regime thresholds; everything split; mixed
            op.fwd.plans[key] = ops.SpmmPlan.build(op.fwd.rowptr, *alt)
            assert rel_err(ops.spmm_raw(op.fwd, X).cpu(), want) < 4e-6, (d, alt)
        op.fwd.plans[key] = keep
        # round 6: the plan's PERMUTED CSR (rows stored by descending length class, slot -> row map, lists in slots): the same bits as the
        # operand's own CSR - with every bucket populated (lane-group rows incl. empty ones, wavefront, block and split rows), with the
        # "+ Z" and softmax epilogues, with column slices, and with the XCD-contiguous block -> row map
        assert op.fwd.val is None and keep.slot_row is not None          # (the default plan of a pattern-only operand is the permuted one)
        for alt in (key, (32, 32, 64)):
            perm = ops.SpmmPlan.build(op.fwd.rowptr, *alt, colidx=op.fwd.colidx, order_rows=True)
            assert perm.slot_row is not None and sorted(perm.slot_row.tolist()) == list(range(n_rows))
            plain = ops.SpmmPlan.build(op.fwd.rowptr, *alt)
            for epi in (lambda: None, lambda: ops.spmm_epilogue(ops.EPI_NONE, 0.25, Z), lambda: ops.spmm_epilogue(ops.EPI_SOFTMAX),
                        lambda: ops.spmm_epilogue(ops.EPI_SOFTMAX_BWD, 0.5, Z, torch.softmax(Z, dim=-1)),
                        lambda: ops.spmm_epilogue(xcd_contiguous=True)):
                op.fwd.plans[key] = plain
                a_ = ops.spmm_raw(op.fwd, X, epilogue=epi())
                op.fwd.plans[key] = perm
                b_ = ops.spmm_raw(op.fwd, X, epilogue=epi())
                assert torch.equal(a_.view(torch.int32), b_.view(torch.int32)), (d, alt)
        op.fwd.plans[key] = keep
        Y2 = ops.spmm_raw(op.fwd, X, epilogue=ops.spmm_epilogue(ops.EPI_NONE, 0.25, Z))
        assert rel_err(Y2.cpu(), 0.25 * Z_cpu + want) < 4e-6, d
        Y3 = Z.clone(); ops.spmm_raw(op.fwd, X, out=Y3, accumulate=True)
        assert rel_err(Y3.cpu(), Z_cpu + want) < 4e-6, d
        Y4 = ops.spmm_raw(op.fwd, X, epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX))
        sm = torch.softmax(want, dim=-1)
        assert rel_err(Y4.cpu(), sm) < 5e-6, d
        # backward of a softmax layer: t = 0.5 Z + A X is the incoming gradient, S the forward output
        S_cpu = torch.softmax(torch.tensor(rng.standard_normal((n_rows, d)).astype(np.float32)), dim=-1)
        t = 0.5 * Z_cpu + want
        want_b = S_cpu * (t - (t * S_cpu).sum(-1, keepdim=True))
        Y5 = ops.spmm_raw(op.fwd, X, epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX_BWD, 0.5, Z, S_cpu.to(DEV)))
        assert rel_err(Y5.cpu(), want_b) < 1e-5, d
    # the transposed operand (hub COLUMNS become many short rows gathering with col_scale)
    G_cpu = torch.tensor(rng.standard_normal((n_rows, 64)).astype(np.float32))
    assert rel_err(ops.spmm_raw(op.bwd, G_cpu.to(DEV)).cpu(), torch.sparse.mm(A_cpu.t(), G_cpu)) < 2e-6


@pytest.mark.parametrize("d", [64, 128, 20])
def test_spmm_nontemporal_policy_is_bit_identical(ops, d):
    """llmrec_spmm_epilogue_t.x_nt_from_row (round-5 cache-policy experiment, profiles/experiments/r05_spmm_nt.md): gathered rows from that
    index on are loaded non-temporally - a hint, so every row bucket (lane group, wavefront, block, split segments) must give the SAME
    BITS as the default policy, whatever the threshold, with and without an init term."""
    rng = np.random.default_rng(500 + d)
    n_rows, n_cols = 600, 45000
    degs = rng.integers(0, 40, size=n_rows)
    for k, dg in enumerate([0, 1, 32, 33, 511, 512, 513, 4095, 4097, 16385, 20000]):
        degs[(k * 13 + 1) % n_rows] = dg
    rows, cols = rand_graph(rng, n_rows, n_cols, degs)
    deg = np.bincount(rows, minlength=n_rows)
    s = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0).astype(np.float32)
    A = torch.sparse_coo_tensor(torch.tensor(np.vstack([rows, cols])), torch.tensor(s[rows]), (n_rows, n_cols)).to(DEV)
    a = ops.operand_from_sparse_tensor(A).fwd
    assert a.val is None                                                 # pattern-only: the products the policy applies to
    X = torch.tensor(rng.standard_normal((n_cols, d)).astype(np.float32)).to(DEV)
    Z = torch.tensor(rng.standard_normal((n_rows, d)).astype(np.float32)).to(DEV)
    want = ops.spmm_raw(a, X)
    want_z = ops.spmm_raw(a, X, epilogue=ops.spmm_epilogue(ops.EPI_NONE, 0.5, Z))
    for H in (1, 100, 20000, n_cols, n_cols + 7):
        got = ops.spmm_raw(a, X, epilogue=ops.spmm_epilogue(ops.EPI_NONE, x_nt_from_row=H))
        assert torch.equal(got, want), H
        got_z = ops.spmm_raw(a, X, epilogue=ops.spmm_epilogue(ops.EPI_NONE, 0.5, Z, x_nt_from_row=H))
        assert torch.equal(got_z, want_z), H


def test_spmm_general_values_and_strided_operands(ops):
    rng = np.random.default_rng(5)
    n_rows, n_cols, d = 120, 900, 64
    rows, cols = rand_graph(rng, n_rows, n_cols, degree_mix(rng, n_rows, n_cols))
    vals = rng.standard_normal(rows.size).astype(np.float32)
    A_cpu = torch.sparse_coo_tensor(torch.tensor(np.vstack([rows, cols])), torch.tensor(vals), (n_rows, n_cols))
    big = torch.tensor(rng.standard_normal((n_cols, 3 * d)).astype(np.float32))
    X_cpu = big[:, d:2 * d].clone().requires_grad_(True)
    Y_cpu = torch.sparse.mm(A_cpu, X_cpu)
    Y_cpu.sum().backward()
    op = ops.operand_from_sparse_tensor(A_cpu.to(DEV))
    assert op.fwd.val is not None
    Xg = big.to(DEV)[:, d:2 * d]                              # ld = 3 d view, no copy
    Xg.requires_grad_(True)
    Y = ops.spmm(op, Xg)
    Y.sum().backward()
    assert rel_err(Y.detach().cpu(), Y_cpu.detach()) < 2e-6
    assert rel_err(Xg.grad.cpu(), X_cpu.grad) < 2e-6


def test_bipartite_graph_matches_oracle_normalisation(ops):
    import scipy.sparse as sp
    rng = np.random.default_rng(9)
    U, I, d = 150, 400, 64
    rows, cols = rand_graph(rng, U, I, rng.integers(0, 30, size=U))
    R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(U, I))
    a_ui, a_iu = O.normalized_graphs(R)
    g = ops.BipartiteGraph.from_edges(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), U, I)
    Xi = torch.tensor(rng.standard_normal((I, d)).astype(np.float32))
    Xu = torch.tensor(rng.standard_normal((U, d)).astype(np.float32))
    assert rel_err(ops.spmm(g.ui, Xi.to(DEV)).cpu(), torch.sparse.mm(a_ui, Xi)) < 2e-6
    assert rel_err(ops.spmm(g.iu, Xu.to(DEV)).cpu(), torch.sparse.mm(a_iu, Xu)) < 2e-6
    assert rel_err(ops.spmm_raw(g.ui.bwd, Xu.to(DEV)).cpu(), torch.sparse.mm(a_ui.t(), Xu)) < 2e-6
    assert rel_err(ops.spmm_raw(g.iu.bwd, Xi.to(DEV)).cpu(), torch.sparse.mm(a_iu.t(), Xi)) < 2e-6


# ------------------------------------------------------------------------------------------
# R4 projection
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(1000, 512, 64), (333, 24, 64), (97, 56, 16), (70000, 64, 64), (50, 1536, 64), (129, 40, 128)])
def test_linear_forward_and_weight_grad(ops, M, K, N):
    rng = np.random.default_rng(M + K)
    X = torch.tensor(rng.standard_normal((M, K)).astype(np.float32))
    W = torch.tensor((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), requires_grad=True)
    b = torch.tensor(rng.standard_normal(N).astype(np.float32), requires_grad=True)
    G = torch.tensor(rng.standard_normal((M, N)).astype(np.float32))
    Y = F.linear(X, W, b)
    Y.backward(G)
    Wg = W.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True)
    Yg = ops.linear(X.to(DEV), Wg, bg)
    Yg.backward(G.to(DEV))
    assert rel_err(Yg.detach().cpu(), Y.detach()) < 5e-6
    assert rel_err(Wg.grad.cpu(), W.grad) < 2e-5
    assert rel_err(bg.grad.cpu(), b.grad) < 2e-5


def test_linear_grouped_matches_single_launches(ops):
    """llmrec_linear_fwd_grouped_f32: several projections (different M, K, strided outputs) in one launch."""
    rng = np.random.default_rng(8)
    for N in (64, 16, 48):
        shapes = [(700, 512), (700, 768), (530, 1536), (129, 24), (1, 40), (300, 36)]
        big = torch.zeros(700, 3 * N, device=DEV)
        jobs, refs = [], []
        for i, (M, K) in enumerate(shapes):
            X = torch.tensor(rng.standard_normal((M, K)).astype(np.float32))
            W = torch.tensor((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
            b = torch.tensor(rng.standard_normal(N).astype(np.float32))
            out = big[:M, (i % 3) * N:(i % 3 + 1) * N] if i < 2 else torch.empty(M, N, device=DEV)
            jobs.append((X.to(DEV), W.to(DEV), b.to(DEV), out))
            refs.append(F.linear(X, W, b))
        ops.linear_fwd_grouped(jobs, N)
        for (X, W, b, out), ref in zip(jobs, refs):
            assert rel_err(out.cpu(), ref) < 5e-6
        # split-precision variant: fp32-roundoff-class error against an fp64 product
        for _, _, _, out in jobs:
            out.zero_()
        ops.linear_fwd_grouped(jobs, N, precision="bf16x3")
        for (X, W, b, out) in jobs:
            ref64 = X.double().cpu() @ W.double().cpu().t() + b.double().cpu()
            assert rel_err(out.cpu(), ref64) < 2e-6


def test_linear_wgrad_grouped(ops):
    rng = np.random.default_rng(12)
    for (N, K) in ((64, 1536), (64, 192), (16, 40), (64, 36)):
        Ms = [700, 333, 48, 1]
        Xs = [torch.tensor(rng.standard_normal((M, K)).astype(np.float32)) for M in Ms]
        big = torch.tensor(rng.standard_normal((700, 3 * N)).astype(np.float32))
        dYs = [big[:Ms[0], N:2 * N]] + [torch.tensor(rng.standard_normal((M, N)).astype(np.float32)) for M in Ms[1:]]
        ref_w = sum(dy.t() @ x for dy, x in zip(dYs, Xs)); ref_b = sum(dy.sum(0) for dy in dYs)
        bigg = big.to(DEV)
        pairs = [(bigg[:Ms[0], N:2 * N], Xs[0].to(DEV))] + [(dy.to(DEV), x.to(DEV)) for dy, x in zip(dYs[1:], Xs[1:])]
        dW = torch.full((N, K), 7.0, device=DEV); db = torch.full((N,), -3.0, device=DEV)
        ops.linear_wgrad_grouped(pairs, dW, db, False)
        assert rel_err(dW.cpu(), ref_w) < 2e-5 and rel_err(db.cpu(), ref_b) < 2e-5
        ops.linear_wgrad_grouped(pairs[:2], dW, db, True)          # accumulate
        ref_w2 = ref_w + sum(dy.t() @ x for dy, x in zip(dYs[:2], Xs[:2])); ref_b2 = ref_b + sum(dy.sum(0) for dy in dYs[:2])
        assert rel_err(dW.cpu(), ref_w2) < 2e-5 and rel_err(db.cpu(), ref_b2) < 2e-5


def test_linear_wgrad_bf16x3_matches_fp64(ops):
    """llmrec_linear_wgrad_grouped_bf16x3: grouped (dY, X) pairs with ragged row counts (slab tails, a one-row
    problem) against the fp64 product; error must be of fp32-roundoff class, like the fp32 MFMA kernel's."""
    rng = np.random.default_rng(21)
    N, K = 64, 320
    Ms = [1000, 1, 333, 64]
    pairs_cpu = [(torch.tensor(rng.standard_normal((m, N)).astype(np.float32)), torch.tensor(rng.standard_normal((m, K)).astype(np.float32) * 3)) for m in Ms]
    want = sum(dy.double().t() @ x.double() for dy, x in pairs_cpu)
    want_b = sum(dy.double().sum(0) for dy, _ in pairs_cpu)
    big = torch.zeros(max(Ms), 3 * N, device=DEV)
    pairs = []
    for dy, x in pairs_cpu:                                    # dY as a strided view (ld = 3 N), as the fused step passes it
        holder = torch.zeros(dy.shape[0], 3 * N, device=DEV); holder[:, N:2 * N] = dy.to(DEV)
        pairs.append((holder[:, N:2 * N], x.to(DEV)))
    for precision in ("bf16x3", "f32"):
        dW = torch.full((N, K), 7.0, device=DEV); db = torch.full((N,), 7.0, device=DEV)
        ops.linear_wgrad_grouped(pairs, dW, db, False, precision=precision)
        e = float((dW.double().cpu() - want).abs().max() / want.abs().max())
        eb = float((db.double().cpu() - want_b).abs().max() / want_b.abs().max())
        assert e < 3e-6 and eb < 3e-6, (precision, e, eb)
        ops.linear_wgrad_grouped(pairs, dW, db, True, precision=precision)       # accumulate
        assert float((dW.double().cpu() - 2 * want).abs().max() / want.abs().max()) < 6e-6


def test_linear_multi_shares_one_weight(ops):
    rng = np.random.default_rng(3)
    M, K, N = 300, 56, 64
    Xs = [torch.tensor(rng.standard_normal((M, K)).astype(np.float32)) for _ in range(5)]
    W = torch.tensor(rng.standard_normal((N, K)).astype(np.float32), requires_grad=True)
    b = torch.tensor(rng.standard_normal(N).astype(np.float32), requires_grad=True)
    Gs = [torch.tensor(rng.standard_normal((M, N)).astype(np.float32)) for _ in range(5)]
    sum((F.linear(x, W, b) * g).sum() for x, g in zip(Xs, Gs)).backward()
    Wg = W.detach().to(DEV).requires_grad_(True); bg = b.detach().to(DEV).requires_grad_(True)
    outs = ops.linear_multi(Wg, bg, [x.to(DEV) for x in Xs])
    sum((o * g.to(DEV)).sum() for o, g in zip(outs, Gs)).backward()
    assert rel_err(Wg.grad.cpu(), W.grad) < 2e-5
    assert rel_err(bg.grad.cpu(), b.grad) < 2e-5


# ------------------------------------------------------------------------------------------
# R3 / R6 row ops
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [64, 16, 128, 20])
def test_softmax_rows(ops, d):
    rng = np.random.default_rng(d)
    Z = torch.tensor((rng.standard_normal((211, d)) * 3).astype(np.float32), requires_grad=True)
    G = torch.tensor(rng.standard_normal((211, d)).astype(np.float32))
    Y = torch.softmax(Z, dim=-1); Y.backward(G)
    Zg = Z.detach().to(DEV).requires_grad_(True)
    Yg = ops.softmax_rows(Zg); Yg.backward(G.to(DEV))
    assert rel_err(Yg.detach().cpu(), Y.detach()) < 2e-6
    assert rel_err(Zg.grad.cpu(), Z.grad) < 1e-5


@pytest.mark.parametrize("d", [64, 16, 128])
def test_fuse_mean_normalize_add(ops, d):
    rng = np.random.default_rng(d + 1)
    rows = 203
    means = [torch.tensor(rng.standard_normal((rows, d)).astype(np.float32), requires_grad=True) for _ in range(3)]
    norms = [torch.tensor(rng.standard_normal((rows, d)).astype(np.float32), requires_grad=True) for _ in range(8)]
    with torch.no_grad():
        norms[2][5] = 0.0                                   # a zero row: F.normalize's eps clamp
    rates = [0.02, 0.02, 2.8] + [0.005] * 5
    G = torch.tensor(rng.standard_normal((rows, d)).astype(np.float32))
    out = torch.mean(torch.stack(means), dim=0)
    for r, t in zip(rates, norms):
        out = out + r * F.normalize(t, p=2, dim=1)
    out.backward(G)
    mg = [m.detach().to(DEV).requires_grad_(True) for m in means]
    ng = [n.detach().to(DEV).requires_grad_(True) for n in norms]
    og = ops.fuse(mg, ng, rates)
    og.backward(G.to(DEV))
    assert rel_err(og.detach().cpu(), out.detach()) < 2e-6
    for a, b in zip(mg + ng, means + norms):
        assert rel_err(a.grad.cpu(), b.grad) < 2e-5


def test_sumsq_and_adamw(ops):
    rng = np.random.default_rng(2)
    Xs = [torch.tensor(rng.standard_normal(s).astype(np.float32), requires_grad=True) for s in [(100, 64), (333, 64), (7, 16)]]
    coef = 1e-5 * 0.5 / 80
    ref = sum(coef * (x ** 2).sum() for x in Xs); ref.backward()
    Xg = [x.detach().to(DEV).requires_grad_(True) for x in Xs]
    got = ops.sumsq(coef, Xg)[0]; got.backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    for a, b in zip(Xg, Xs):
        assert rel_err(a.grad.cpu(), b.grad) < 1e-6
    # AdamW: 5 steps against torch.optim.AdamW with the reference's construction (lr only)
    p_ref = torch.nn.Parameter(torch.tensor(rng.standard_normal((50, 64)).astype(np.float32)))
    p_gpu = torch.nn.Parameter(p_ref.detach().clone().to(DEV))
    o_ref = torch.optim.AdamW([{'params': [p_ref]}], lr=1e-4)
    o_gpu = ops.FusedAdamW([p_gpu], lr=1e-4)
    for step in range(5):
        g = torch.tensor(rng.standard_normal((50, 64)).astype(np.float32)) * (10.0 ** (step - 2))
        p_ref.grad = g.clone(); p_gpu.grad = g.to(DEV)
        o_ref.step(); o_gpu.step()
        assert rel_err(p_gpu.detach().cpu(), p_ref.detach()) < 1e-6


# ------------------------------------------------------------------------------------------
# R7 BPR + prune
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("drop,B,d", [(0.71, 1126, 64), (0.0, 1024, 64), (0.71, 35, 16), (0.5, 2048, 128), (0.999, 40, 64)])
def test_bpr_prune_forward_backward(ops, drop, B, d):
    rng = np.random.default_rng(B)
    U, I = 300, 500
    Eu = torch.tensor((rng.standard_normal((U, d)) * 0.3).astype(np.float32), requires_grad=True)
    Ei = torch.tensor((rng.standard_normal((I, d)) * 0.3).astype(np.float32), requires_grad=True)
    users = torch.tensor(rng.integers(0, U, size=B)); pos = torch.tensor(rng.integers(0, I, size=B)); neg = torch.tensor(rng.integers(0, I, size=B))
    cfg = O.Config(batch_size=1024, decay=1e-5, prune_loss_drop_rate=drop)
    mf, emb = O.bpr_loss(Eu[users], Ei[pos], Ei[neg], cfg)
    (0.7 * mf + 1.3 * emb).backward()
    Eug = Eu.detach().to(DEV).requires_grad_(True); Eig = Ei.detach().to(DEV).requires_grad_(True)
    out = ops.bpr_prune(Eug, Eig, users.to(DEV), pos.to(DEV), neg.to(DEV), drop, 1e-5, 1024)
    (0.7 * out[0] + 1.3 * out[1]).backward()
    if int((1 - drop) * B) == 0:
        assert np.isnan(float(out[0])) and np.isnan(float(mf))
        return
    assert abs(float(out[0]) - float(mf)) < 2e-6 * abs(float(mf))
    assert abs(float(out[1]) - float(emb)) < 2e-6 * abs(float(emb))
    assert rel_err(Eug.grad.cpu(), Eu.grad) < 2e-5
    assert rel_err(Eig.grad.cpu(), Ei.grad) < 2e-5


def test_bpr_device_batch_count(ops):
    """n_valid on the device (graph-replay path): padding entries beyond it are ignored."""
    rng = np.random.default_rng(4)
    U, I, d, B, Bmax = 100, 120, 64, 70, 96
    Eu = torch.tensor(rng.standard_normal((U, d)).astype(np.float32) * 0.2)
    Ei = torch.tensor(rng.standard_normal((I, d)).astype(np.float32) * 0.2)
    idx = [torch.tensor(rng.integers(0, n, size=Bmax)) for n in (U, I, I)]
    cfg = O.Config(batch_size=64, decay=1e-5, prune_loss_drop_rate=0.71)
    mf, emb = O.bpr_loss(Eu[idx[0][:B]], Ei[idx[1][:B]], Ei[idx[2][:B]], cfg)
    nv = torch.tensor([B], dtype=torch.int32, device=DEV)
    out = ops.bpr_prune(Eu.to(DEV), Ei.to(DEV), idx[0].to(DEV), idx[1].to(DEV), idx[2].to(DEV), 0.71, 1e-5, 64, n_valid=nv)
    assert abs(float(out[0]) - float(mf)) < 2e-6 * abs(float(mf))
    assert abs(float(out[1]) - float(emb)) < 2e-6 * abs(float(emb))


def test_bpr_multi_sharded_matches_oracle_on_concatenated_batch(ops):
    """llmrec_bpr_multi_fwd_sharded_f32 (phase 1 -> gather blocks -> phase 2) + llmrec_bpr_multi_bwd_f32
    on three ranks with ragged valid counts (one rank empty), two problems over one batch: the sum of
    the ranks' mf shares, the emb value and the summed gradients equal the oracle's BPR + prune on
    the concatenation of the valid samples."""
    import ctypes
    from llmrec_amd import _lib
    rng = np.random.default_rng(11)
    U, I, d, cap, P, W = 200, 260, 64, 96, 2, 3
    valid = [70, 0, 96]
    tabs = [(torch.tensor((rng.standard_normal((U, d)) * 0.3).astype(np.float32)), torch.tensor((rng.standard_normal((I, d)) * 0.3).astype(np.float32)))
            for _ in range(P)]
    idx = [[torch.tensor(rng.integers(0, n, size=cap)) for n in (U, I, I)] for _ in range(W)]
    cat = [torch.cat([idx[r][k][:valid[r]] for r in range(W)]) for k in range(3)]
    cfg = O.Config(batch_size=64, decay=1e-5, prune_loss_drop_rate=0.71)
    want = []
    for Eu, Ei in tabs:
        a = Eu.clone().requires_grad_(True); b = Ei.clone().requires_grad_(True)
        mf, emb = O.bpr_loss(a[cat[0]], b[cat[1]], b[cat[2]], cfg)
        (0.7 * mf + 1.3 * emb).backward()
        want.append((float(mf.detach()), float(emb.detach()), a.grad, b.grad))
    gsz = P * cap + 4 * P + 1
    dev_t = [(Eu.to(DEV), Ei.to(DEV)) for Eu, Ei in tabs]
    grads = [(torch.zeros(U, d, device=DEV), torch.zeros(I, d, device=DEV)) for _ in range(P)]
    arr = (ops.BprProblem * P)()
    for i in range(P):
        arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = dev_t[i][0].data_ptr(), d, dev_t[i][1].data_ptr(), d
        arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = grads[i][0].data_ptr(), d, grads[i][1].data_ptr(), d
        arr[i].g_mf, arr[i].g_emb = 0.7, 1.3
    st = torch.cuda.current_stream().cuda_stream
    dev_idx = [[x.to(DEV) for x in idx[r]] for r in range(W)]
    nv = [torch.tensor([valid[r]], dtype=torch.int32, device=DEV) for r in range(W)]
    saved = [torch.zeros(P * ops.bpr_saved_floats(cap), device=DEV) for _ in range(W)]
    outs = [torch.zeros(P, 2, device=DEV) for _ in range(W)]
    blocks = torch.zeros(W, gsz, device=DEV)
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())

    def call(phase, r):
        _lib.call("llmrec_bpr_multi_fwd_sharded_f32", P, arr, d, p_(dev_idx[r][0]), p_(dev_idx[r][1]), p_(dev_idx[r][2]), cap, p_(nv[r]),
                  1 - 0.71, 1e-5, 64.0, phase, p_(blocks[r]), p_(blocks), W, gsz, r, p_(outs[r]), p_(saved[r]), st)
    for r in range(W):
        call(1, r)
    assert [int(blocks[r, -1]) for r in range(W)] == valid
    for r in range(W):
        call(2, r)
        plan = ops.bpr_scatter_plan(dev_idx[r][0], dev_idx[r][1], dev_idx[r][2], nv[r])
        _lib.call("llmrec_bpr_multi_bwd_f32", P, arr, d, p_(dev_idx[r][0]), p_(dev_idx[r][1]), p_(dev_idx[r][2]), cap, p_(nv[r]),
                  1e-5, 64.0, p_(saved[r]), p_(plan), st)
    for i in range(P):
        mf = sum(float(outs[r][i, 0]) for r in range(W))
        assert abs(mf - want[i][0]) < 3e-6 * abs(want[i][0])
        for r in range(W):
            assert abs(float(outs[r][i, 1]) - want[i][1]) < 3e-6 * abs(want[i][1])
        assert rel_err(grads[i][0].cpu(), want[i][2]) < 2e-5
        assert rel_err(grads[i][1].cpu(), want[i][3]) < 2e-5


@pytest.mark.parametrize("drop,cap,valid,d", [(0.71, 1126, 1126, 64), (0.0, 1024, 1000, 64), (0.5, 2048, 1500, 128), (0.71, 96, 0, 64), (0.999, 40, 40, 16)])
def test_bpr_select_and_backward_in_one_launch(ops, drop, cap, valid, d):
    """scores -> llmrec_bpr_multi_select_bwd_f32 -> llmrec_bpr_multi_losses_f32 (the fused step's loss path) against
    llmrec_bpr_multi_fwd_f32 + llmrec_bpr_multi_bwd_f32: `saved`, `out` AND the gradient rows bit for bit (valid < capacity included,
    an empty batch included; round 6: the scatter is deterministic - one owner per destination row adds the duplicates in slot order),
    and against the oracle's autograd."""
    import ctypes
    from llmrec_amd import _lib
    rng = np.random.default_rng(5)
    U, I, P = 300, 420, 3
    tabs = [(torch.tensor((rng.standard_normal((U, d)) * 0.3).astype(np.float32)), torch.tensor((rng.standard_normal((I, d)) * 0.3).astype(np.float32)))
            for _ in range(P)]
    idx = [torch.tensor(rng.integers(0, n, size=cap)).to(DEV) for n in (U, I, I)]
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    dev_t = [(a.to(DEV), b.to(DEV)) for a, b in tabs]
    w = [(1.0, 1.0), (0.7, 0.0), (0.02, 1.3)]
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream().cuda_stream

    def run(fused):
        grads = [(torch.zeros(U, d, device=DEV), torch.zeros(I, d, device=DEV)) for _ in range(P)]
        arr = (ops.BprProblem * P)()
        for i in range(P):
            arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = dev_t[i][0].data_ptr(), d, dev_t[i][1].data_ptr(), d
            arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = grads[i][0].data_ptr(), d, grads[i][1].data_ptr(), d
            arr[i].g_mf, arr[i].g_emb = w[i]
        saved = torch.full((P * ops.bpr_saved_floats(cap),), 7.0, device=DEV)
        out = torch.zeros(P, 2, device=DEV)
        flag_u, flag_i = torch.full((U,), 42, dtype=torch.uint8, device=DEV), torch.zeros(I, dtype=torch.uint8, device=DEV)   # 42: a stale stamp
        stamp = torch.tensor([41 + 255 * 3], dtype=torch.int32, device=DEV)     # the scores launch advances it: stamp value (42 + 765) % 255 + 1 = 43
        common = (P, arr, d, p_(idx[0]), p_(idx[1]), p_(idx[2]), cap, p_(nv))
        plan = ops.bpr_scatter_plan(idx[0], idx[1], idx[2], nv)
        if fused:
            _lib.call("llmrec_bpr_multi_scores_f32", *common, p_(saved), p_(stamp), st)
            _lib.call("llmrec_bpr_multi_select_bwd_f32", *common, 1 - drop, 1e-5, 64.0, p_(saved), p_(flag_u), p_(flag_i), p_(stamp), p_(plan), st)
            _lib.call("llmrec_bpr_multi_losses_f32", P, cap, p_(nv), 1 - drop, 1e-5, 64.0, p_(out), p_(saved), st)
        else:
            _lib.call("llmrec_bpr_multi_fwd_f32", *common, 1 - drop, 1e-5, 64.0, p_(out), p_(saved), st)
            _lib.call("llmrec_bpr_multi_bwd_f32", *common, 1e-5, 64.0, p_(saved), p_(plan), st)
        torch.cuda.synchronize()
        if fused:                                            # the rows of the valid samples carry the step's stamp, no other row changes
            assert int(stamp[0]) == 42 + 255 * 3
            want_u = torch.full((U,), 42, dtype=torch.uint8); want_u[idx[0][:valid].cpu()] = 43
            want_i = torch.zeros(I, dtype=torch.uint8); want_i[idx[1][:valid].cpu()] = 43; want_i[idx[2][:valid].cpu()] = 43
            assert torch.equal(flag_u.cpu(), want_u) and torch.equal(flag_i.cpu(), want_i)
            keep = [(a.clone(), b.clone()) for a, b in grads]
            _lib.call("llmrec_bpr_multi_zero_rows_f32", *common, st)
            torch.cuda.synchronize()
            assert all(float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0 for a, b in grads)
            grads = keep
        return saved.cpu(), out.cpu(), [(a.cpu(), b.cpu()) for a, b in grads]
    s0, o0, g0 = run(False)
    s1, o1, g1 = run(True)
    stride = ops.bpr_saved_floats(cap)
    for i in range(P):                                       # every slot the contract defines: coefficients, norm sums + k, kept values, norms
        a, b = s0[i * stride:(i + 1) * stride], s1[i * stride:(i + 1) * stride]
        assert torch.equal(a[:cap + 4].view(torch.int32), b[:cap + 4].view(torch.int32))
        lo = cap + 4
        for slot in (0, 1, 2, 3, 4):
            assert torch.equal(a[lo + slot * cap:lo + slot * cap + valid].view(torch.int32), b[lo + slot * cap:lo + slot * cap + valid].view(torch.int32)), slot
    assert torch.equal(o0.view(torch.int32), o1.view(torch.int32))
    for i in range(P):
        assert torch.equal(g1[i][0].view(torch.int32), g0[i][0].view(torch.int32)) and torch.equal(g1[i][1].view(torch.int32), g0[i][1].view(torch.int32))
    s2, o2, g2 = run(True)                                   # and again: the same bits (no float atomics anywhere in the scatter)
    assert all(torch.equal(g2[i][k].view(torch.int32), g1[i][k].view(torch.int32)) for i in range(P) for k in (0, 1))
    if valid and int((1 - drop) * valid) > 0:
        cfg = O.Config(batch_size=64, decay=1e-5, prune_loss_drop_rate=drop)
        for i, (Eu, Ei) in enumerate(tabs):
            a = Eu.clone().requires_grad_(True); b = Ei.clone().requires_grad_(True)
            u, pp, q = (x[:valid].cpu() for x in idx)
            mf, emb = O.bpr_loss(a[u], b[pp], b[q], cfg)
            (w[i][0] * mf + w[i][1] * emb).backward()
            assert abs(float(o1[i, 0]) - float(mf)) < 3e-6 * abs(float(mf))
            assert rel_err(g1[i][0], a.grad) < 2e-5 and rel_err(g1[i][1], b.grad) < 2e-5


@pytest.mark.parametrize("d,weighted,n_rows", [(448, False, 30000), (448, True, 30000), (64, False, 150000), (192, True, 25000), (20, False, 70000)])
def test_spmm_pipelined_short_rows_equal_one_task_per_lane_group(ops, d, weighted, n_rows):
    """Round 6: the lane-group bucket's tasks as software pipelines (a lane group takes up to 8 tasks and keeps the next task's indices and
    the one after's row pointers in flight) against one task per lane group (epilogue.no_pipeline): the same bits - plain and with the
    "+ Z" epilogue, pattern-only and weighted (col_scale: the transposed products of the step's backward), sliced and unsliced operands,
    float4 and scalar rows (d = 20), with and without the plan's permuted CSR, empty rows and every long-row bucket included."""
    rng = np.random.default_rng(d + int(weighted))
    n_cols = 5000
    degs = rng.integers(0, 34, size=n_rows)
    for k, dg in enumerate([0, 0, 1, 31, 32, 33, 100, 129, 600, 2049, 4000]):
        degs[(k * 17 + 3) % n_rows] = dg
    rows = np.repeat(np.arange(n_rows), degs)
    cols = rng.integers(0, n_cols, size=rows.size)
    key64 = np.unique(rows.astype(np.int64) * n_cols + cols)                   # (distinct (row, col) pairs)
    rows, cols = key64 // n_cols, key64 % n_cols
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, n_rows, n_cols)
    rs = torch.tensor(rng.random(n_rows).astype(np.float32)).to(DEV)
    cs = torch.tensor(rng.random(n_cols).astype(np.float32)).to(DEV) if weighted else None
    a = ops.Csr(n_rows, n_cols, rp, ci, None, rs, cs, {})
    X = torch.tensor(rng.standard_normal((n_cols, d)).astype(np.float32)).to(DEV)
    Z = torch.tensor(rng.standard_normal((n_rows, d)).astype(np.float32)).to(DEV)
    sw, pl = a.plan_for(d)
    assert pl.slot_row is not None and n_rows * (d // sw if sw else 1) > 2 * 32 * 1024         # enough tasks for several per lane group
    _, key = ops.spmm_shape(d, a.nnz)
    for plan in (pl, ops.SpmmPlan.build(rp, *key)):
        a.plans[key] = plan
        for z in (None, Z):
            mk = lambda off: ops.spmm_epilogue(ops.EPI_NONE, 0.5 if z is not None else 0.0, z, no_pipeline=off)
            y_pipe = ops.spmm_raw(a, X, epilogue=mk(False))
            y_task = ops.spmm_raw(a, X, epilogue=mk(True))
            assert torch.equal(y_pipe.view(torch.int32), y_task.view(torch.int32)), (d, weighted, plan is pl, z is not None)
    A64 = torch.sparse_coo_tensor(torch.tensor(np.vstack([rows, cols])), torch.ones(len(rows), dtype=torch.float64), (n_rows, n_cols))
    want = torch.sparse.mm(A64, X.cpu().double() * (cs.cpu().double()[:, None] if weighted else 1.0)) * rs.cpu().double()[:, None]
    assert rel_err(ops.spmm_raw(a, X).cpu(), want.float()) < 4e-6


def test_bpr_scatter_plan_layout(ops):
    """llmrec_bpr_scatter_plan: sorted (id << 32 | slot) keys per side, the unused slots (id 0xffffffff) at the end, and the run
    lengths at the first position of every run."""
    rng = np.random.default_rng(2)
    for cap, valid, with_count in ((37, 29, True), (37, 37, False), (1126, 1100, True), (16, 0, True)):
        u, p, q = (torch.tensor(rng.integers(0, n, size=cap)) for n in (6, 9, 9))
        nv = torch.tensor([valid], dtype=torch.int32, device=DEV) if with_count else None
        plan = ops.bpr_scatter_plan(u.to(DEV), p.to(DEV), q.to(DEV), nv).cpu()
        keys = plan[:3 * cap].numpy().astype(np.uint64)
        runlen = plan[3 * cap:].view(torch.int32)[:3 * cap].numpy()
        none = 0xFFFFFFFF
        ku = sorted((int(u[b]) << 32) | b for b in range(valid)) + [(none << 32) | b for b in range(valid, cap)]
        ki = sorted([(int(p[b]) << 32) | b for b in range(valid)] + [(int(q[b]) << 32) | (cap + b) for b in range(valid)]) + \
            sorted([(none << 32) | b for b in range(valid, cap)] + [(none << 32) | (cap + b) for b in range(valid, cap)])
        assert [int(x) for x in keys[:cap]] == ku and [int(x) for x in keys[cap:]] == ki
        for side, want in ((runlen[:cap], ku), (runlen[cap:], ki)):
            ids = [k >> 32 for k in want]
            for j, i_ in enumerate(ids):
                head = i_ != none and (j == 0 or ids[j - 1] != i_)
                assert side[j] == (ids.count(i_) if head else 0), (cap, valid, j)


@pytest.mark.parametrize("d", [64, 24, 200])
def test_bpr_backward_is_deterministic_on_heavy_duplicates_and_shared_targets(ops, d):
    """The loss backward's scatter (llmrec_bpr_multi_bwd_f32 through llmrec_bpr_scatter_plan): (a) a batch in which every sample hits
    ONE user, two positives and three negatives (runs of hundreds of duplicates) against the oracle's index_put backward; (b) problems
    that share one dEu buffer (the step's five attribute problems all scatter into d prof_u) = the sum of the same problems run into
    buffers of their own; (c) two runs: the same bits."""
    import ctypes
    from llmrec_amd import _lib
    rng = np.random.default_rng(17)
    U, I, P, cap, valid = 40, 50, 4, 700, 613
    tabs = [(torch.tensor((rng.standard_normal((U, d)) * 0.3).astype(np.float32)), torch.tensor((rng.standard_normal((I, d)) * 0.3).astype(np.float32)))
            for _ in range(P)]
    tabs = [(tabs[0][0], tabs[0][1])] + [(tabs[1][0], t[1]) for t in tabs[1:]]         # problems 1.. share their USER table (as prof_u)
    idx = [torch.tensor(rng.choice(c, size=cap)) for c in ([7], [3, 11], [5, 6, 49])]
    dev = [x.to(DEV) for x in idx]
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    dev_t = [(a.to(DEV), b.to(DEV)) for a, b in tabs]
    w = [(1.0, 1.0)] + [(0.012, 0.0)] * (P - 1)
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    plan = ops.bpr_scatter_plan(dev[0], dev[1], dev[2], nv)

    def run(shared):
        gu = [torch.zeros(U, d, device=DEV) for _ in range(P)]
        if shared:
            gu = [gu[0]] + [gu[1]] * (P - 1)
        gi = [torch.zeros(I, d, device=DEV) for _ in range(P)]
        arr = (ops.BprProblem * P)()
        for i in range(P):
            arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = dev_t[i][0].data_ptr(), d, dev_t[i][1].data_ptr(), d
            arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = gu[i].data_ptr(), d, gi[i].data_ptr(), d
            arr[i].g_mf, arr[i].g_emb = w[i]
        saved = torch.zeros(P * ops.bpr_saved_floats(cap), device=DEV)
        out = torch.zeros(P, 2, device=DEV)
        common = (P, arr, d, p_(dev[0]), p_(dev[1]), p_(dev[2]), cap, p_(nv))
        _lib.call("llmrec_bpr_multi_fwd_f32", *common, 1 - 0.71, 1e-5, 64.0, p_(out), p_(saved), st)
        _lib.call("llmrec_bpr_multi_bwd_f32", *common, 1e-5, 64.0, p_(saved), p_(plan), st)
        torch.cuda.synchronize()
        return [g.cpu() for g in gu], [g.cpu() for g in gi]
    gu_own, gi_own = run(False)
    gu_sh, gi_sh = run(True)
    gu_sh2, gi_sh2 = run(True)
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(gu_sh + gi_sh, gu_sh2 + gi_sh2))
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(gi_own, gi_sh))          # item targets: unchanged
    assert torch.equal(gu_own[0].view(torch.int32), gu_sh[0].view(torch.int32))
    assert rel_err(gu_sh[1], sum(gu_own[1:])) < 2e-5                            # (613 x 3 addends in another order)
    cfg = O.Config(batch_size=64, decay=1e-5, prune_loss_drop_rate=0.71)
    for i, (Eu, Ei) in enumerate(tabs):
        a = Eu.clone().requires_grad_(True); b = Ei.clone().requires_grad_(True)
        u, pp, q = (x[:valid] for x in idx)
        mf, emb = O.bpr_loss(a[u], b[pp], b[q], cfg)
        (w[i][0] * mf + w[i][1] * emb).backward()
        assert rel_err(gu_own[i], a.grad) < 2e-5 and rel_err(gi_own[i], b.grad) < 2e-5


# ------------------------------------------------------------------------------------------
# R9/R10 scoring + top-K
# ------------------------------------------------------------------------------------------
def _train_csr(ops, train_items, U, I):
    rows = np.concatenate([np.full(len(v), u) for u, v in train_items.items()] + [np.zeros(0)]).astype(np.int64)
    cols = np.concatenate([np.asarray(v) for v in train_items.values()] + [np.zeros(0)]).astype(np.int64)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, U, I)
    return ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())


@pytest.mark.parametrize("U,I,d,K", [(150, 1000, 64, 50), (70, 130, 16, 50), (33, 64, 64, 20), (200, 777, 128, 64), (10, 45, 64, 50)])
def test_score_topk_lists_bit_exact(ops, U, I, d, K):
    rng = np.random.default_rng(U + I)
    Eu = rng.standard_normal((U, d)).astype(np.float32)
    Ei = rng.standard_normal((I, d)).astype(np.float32)
    train_items = {u: sorted(rng.choice(I, size=int(rng.integers(0, min(I - 1, 40))), replace=False).tolist()) for u in range(U)}
    train_items[0] = sorted(rng.choice(I, size=I - 7, replace=False).tolist())        # fewer than K candidates
    train_items[1] = []
    users = rng.permutation(U)[: U - 3]
    Eug, Eig = torch.tensor(Eu).to(DEV), torch.tensor(Ei).to(DEV)
    q = torch.tensor(users).to(DEV)
    S = ops.scores(Eug, Eig, q).cpu().numpy()
    # the MFMA scores are the documented k-ordered fp32 fma chain
    chain = O.scores_fma_chain(Eu[users], Ei, order="mfma16x16x4")
    assert np.array_equal(S, chain), "MFMA score arithmetic differs from the documented fma chain"
    assert rel_err(S, Eu[users].astype(np.float64) @ Ei.astype(np.float64).T) < 1e-5
    idx, sc = ops.score_topk(Eug, Eig, q, _train_csr(ops, train_items, U, I), K)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for row, u in enumerate(users):
        want = O.rank_topk_np(S[row], train_items[u], K)
        got = idx[row][idx[row] >= 0]
        assert np.array_equal(got, want), (row, u)
        assert np.array_equal(sc[row][: len(want)], S[row][want])
        assert np.all(idx[row][len(want):] == -1)


def test_score_topk_exact_ties_follow_reference_rule(ops):
    """Integer-valued embeddings: every score is exact in any summation order, so ties are real
    and the list must equal the reference's heapq ranking (score desc, item id asc)."""
    rng = np.random.default_rng(11)
    U, I, d, K = 64, 500, 64, 50
    Eu = rng.integers(-2, 3, size=(U, d)).astype(np.float32)
    Ei = rng.integers(-1, 2, size=(I, d)).astype(np.float32)
    train_items = {u: sorted(rng.choice(I, size=20, replace=False).tolist()) for u in range(U)}
    S_ref = (torch.tensor(Eu) @ torch.tensor(Ei).t()).numpy()                     # the reference's GEMM, exact here
    idx, _ = ops.score_topk(torch.tensor(Eu).to(DEV), torch.tensor(Ei).to(DEV), torch.arange(U).to(DEV),
                            _train_csr(ops, train_items, U, I), K)
    idx = idx.cpu().numpy()
    for u in range(U):
        assert idx[u].tolist() == O.rank_topk(S_ref[u], train_items[u], K), u


def test_export_candidates_matches_torch_topk(ops):
    rng = np.random.default_rng(21)
    U, I, d = 130, 900, 64
    Eu = torch.tensor(rng.standard_normal((U, d)).astype(np.float32)); Ei = torch.tensor(rng.standard_normal((I, d)).astype(np.float32))
    got = ops.export_candidates(Eu.to(DEV), Ei.to(DEV), 10).cpu()
    S = ops.scores(Eu.to(DEV), Ei.to(DEV), torch.arange(U).to(DEV)).cpu()
    want = torch.topk(S, k=10, dim=1).indices                      # no exact ties in random fp32 data
    assert torch.equal(got, want)


def test_topk_hits(ops):
    rng = np.random.default_rng(1)
    U, I, K = 40, 300, 50
    test = {u: sorted(rng.choice(I, size=int(rng.integers(1, 5)), replace=False).tolist()) for u in range(U)}
    topk = np.stack([rng.permutation(I)[:K] for _ in range(U)]).astype(np.int32)
    topk[3, 40:] = -1
    csr = _train_csr(ops, test, U, I)
    hits = ops.topk_hits(torch.tensor(topk).to(DEV), torch.arange(U).to(DEV), csr.rowptr, csr.colidx).cpu().numpy()
    want = np.array([[1 if (i >= 0 and i in set(test[u])) else 0 for i in topk[u]] for u in range(U)])
    assert np.array_equal(hits, want)


# ------------------------------------------------------------------------------------------
# R11 device sampler
# ------------------------------------------------------------------------------------------
def test_device_sampler_properties(ops):
    rng = np.random.default_rng(6)
    U, I, B = 500, 300, 200
    train_items = {u: sorted(rng.choice(I, size=int(rng.integers(1, 60)), replace=False).tolist()) for u in range(0, U, 2)}
    csr = _train_csr(ops, train_items, U, I)
    exist = torch.tensor(sorted(train_items), dtype=torch.int64, device=DEV)
    seen_users = set()
    for step in range(4):
        u, p, n = (t.cpu().tolist() for t in ops.sample_bpr(2022, step, exist, I, csr, B))
        assert len(set(u)) == B                                # without replacement (rd.sample)
        for uu, pp, nn_ in zip(u, p, n):
            assert uu in train_items and pp in train_items[uu] and nn_ not in train_items[uu] and 0 <= nn_ < I
        seen_users.update(u)
    assert len(seen_users) >= len(train_items) - 5             # 4 x 200 draws from 250 users covers ~all of them
    a = ops.sample_bpr(2022, 1, exist, I, csr, B); b = ops.sample_bpr(2022, 1, exist, I, csr, B)
    assert all(torch.equal(x, y) for x, y in zip(a, b))       # counter-based: reproducible
    # B > n_exist -> with replacement
    u, _, _ = ops.sample_bpr(1, 0, exist[:50], I, csr, 128)
    assert u.numel() == 128 and set(u.cpu().tolist()) <= set(exist[:50].cpu().tolist())


def test_sample_batch_one_launch_with_augmented_triples(ops):
    """llmrec_sample_batch: slice of the global batch == llmrec_sample_bpr's triples; augmented triples are
    distinct users of the slice with both ids valid, kept pairs first; device step counter advances; two ranks'
    slices tile the global batch."""
    rng = np.random.default_rng(8)
    U, I, B, W = 700, 300, 96, 2
    train_items = {u: sorted(rng.choice(I, size=int(rng.integers(1, 40)), replace=False).tolist()) for u in range(U)}
    csr = _train_csr(ops, train_items, U, I)
    exist = torch.arange(U, dtype=torch.int64, device=DEV)
    aug_pos = torch.tensor(rng.integers(0, int(I * 1.3), size=U), device=DEV)      # ~23 % of the ids are out of range
    aug_neg = torch.tensor(rng.integers(0, int(I * 1.3), size=U), device=DEV)
    n_aug = 31
    for step in (0, 5):
        want = [t.cpu() for t in ops.sample_bpr(77, step, exist, I, csr, B * W)]
        for r in range(W):
            step_dev = torch.tensor([step], dtype=torch.int64, device=DEV)
            u, p, n = (torch.full((B + n_aug,), -7, dtype=torch.int64, device=DEV) for _ in range(3))
            nv = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.sample_batch(77, step_dev, exist, I, csr, B * W, r * B, B, n_aug, aug_pos, aug_neg, u, p, n, nv)
            assert int(step_dev) == step + 1
            u, p, n = u.cpu(), p.cpu(), n.cpu()
            for got, ref in zip((u, p, n), want):
                assert torch.equal(got[:B], ref[r * B:(r + 1) * B])
            kept = int(nv) - B
            assert 0 < kept < n_aug
            au, ap, an = u[B:B + kept].tolist(), p[B:B + kept].tolist(), n[B:B + kept].tolist()
            assert len(set(au)) == kept and set(au) <= set(u[:B].tolist())
            ap_all, an_all = aug_pos.cpu(), aug_neg.cpu()
            for uu, pp, nn_ in zip(au, ap, an):
                assert pp == int(ap_all[uu]) and nn_ == int(an_all[uu]) and pp < I and nn_ < I
            # every drawn user whose pair is valid was kept: the dropped ones are exactly the invalid pairs
            assert torch.equal(u[B + kept:], torch.zeros(n_aug - kept, dtype=torch.int64))
    # run-to-run determinism
    a = ops.sample_bpr(77, 5, exist, I, csr, B)
    sd = torch.tensor([5], dtype=torch.int64, device=DEV)
    u2 = torch.empty(B, dtype=torch.int64, device=DEV); p2 = torch.empty_like(u2); n2 = torch.empty_like(u2)
    ops.sample_batch(77, sd, exist, I, csr, B, 0, B, 0, None, None, u2, p2, n2, torch.zeros(1, dtype=torch.int32, device=DEV))
    assert torch.equal(a[0], u2) and torch.equal(a[1], p2) and torch.equal(a[2], n2)


def test_score_topk_exact_ties_are_ordered_by_item_id(ops):
    """Duplicate item rows give bit-equal scores in different item quarters (different waves' buffers, different
    drains): the lists must hold them in ascending item id, as the reference's (score, id) ordering does."""
    rng = np.random.default_rng(123)
    U, I, d, K = 48, 900, 64, 50
    Eu = rng.standard_normal((U, d)).astype(np.float32)
    base = rng.standard_normal((60, d)).astype(np.float32)
    Ei = base[rng.integers(0, 60, size=I)]                       # only 60 distinct rows: every score value repeats ~15 times
    Eug, Eig = torch.tensor(Eu).to(DEV), torch.tensor(Ei).to(DEV)
    q = torch.arange(U).to(DEV)
    train_items = {u: sorted(rng.choice(I, size=20, replace=False).tolist()) for u in range(U)}
    S = ops.scores(Eug, Eig, q).cpu().numpy()
    idx, sc = ops.score_topk(Eug, Eig, q, _train_csr(ops, train_items, U, I), K)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for u in range(U):
        want = O.rank_topk_np(S[u], train_items[u], K)
        assert np.array_equal(idx[u], want), u
        assert np.array_equal(sc[u], S[u][want])


def test_topk_metrics_match_host_formulae(ops):
    """llmrec_topk_metrics == utility.metrics.metrics_from_hit_matrix (the drop-in's vectorised host formulae,
    themselves checked against the reference's scalar functions in tests/test_host_cpu.py)."""
    from utility.metrics import metrics_from_hit_matrix
    rng = np.random.default_rng(31)
    U, I, K, Ks = 300, 400, 50, (10, 20, 50)
    test = {u: sorted(rng.choice(I, size=int(rng.integers(1, 9)), replace=False).tolist()) for u in range(U)}
    topk = np.stack([rng.permutation(I)[:K] for _ in range(U)]).astype(np.int32)
    for u in range(0, U, 3):                                   # make hits frequent
        topk[u, rng.integers(0, K, size=3)] = rng.choice(test[u], size=3)
    topk[5, 30:] = -1; topk[6, 3:] = -1; topk[7, :] = -1       # short lists
    csr = _train_csr(ops, test, U, I)
    q = torch.arange(U).to(DEV)
    idx = torch.tensor(topk).to(DEV)
    hits = ops.topk_hits(idx, q, csr.rowptr, csr.colidx)
    got = ops.topk_metrics(idx, hits, q, csr.rowptr, Ks).cpu().numpy()
    n_pos = np.array([len(test[u]) for u in range(U)], dtype=np.float64)
    want = metrics_from_hit_matrix(hits.cpu().numpy(), n_pos, Ks, (topk >= 0).sum(1))
    for j, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
        assert np.allclose(got[:, j, :], want[k], rtol=0, atol=1e-13), k
    # round 6: llmrec_topk_eval_sums - hits, per-user metrics and their sums over the users in two launches, into device memory or straight
    # into pinned host memory; a permuted / partial query list; deterministic
    sums = ops.topk_eval_sums(idx, q, csr.rowptr, csr.colidx, Ks)
    assert np.allclose(sums.cpu().numpy(), got.sum(0), rtol=1e-14, atol=1e-12)
    pinned = torch.zeros(4, len(Ks), dtype=torch.float64).pin_memory()
    ops.topk_eval_sums(idx, q, csr.rowptr, csr.colidx, Ks, out=pinned)
    torch.cuda.synchronize()
    assert np.array_equal(pinned.numpy(), sums.cpu().numpy())
    sel = torch.tensor(rng.permutation(U)[:77]).to(DEV)
    part = ops.topk_eval_sums(idx[sel].contiguous(), sel, csr.rowptr, csr.colidx, Ks).cpu().numpy()
    assert np.allclose(part, got[sel.cpu().numpy()].sum(0), rtol=1e-14, atol=1e-12)


def test_fuse_bwd_source_mode_and_zero_rows(ops):
    """llmrec_fuse_bwd_src_f32: d_terms = source (or 0) + the term's gradient, identical to scatter-then-accumulate with
    llmrec_fuse_bwd_f32; llmrec_bpr_multi_zero_rows_f32 clears exactly the rows the multi-problem backward touched."""
    import ctypes as C
    from llmrec_amd import _lib
    from llmrec_amd.ops import _p, _ld, BprProblem
    g = torch.Generator(device=DEV); g.manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    rows, d, T = 301, 64, 4
    dout = rn(rows, d)
    cat = rn(rows, 3 * d); prof = rn(rows, d)
    cat[7] = 0.0                                                            # a zero row: the clamp branch
    terms = [cat[:, 0:d], cat[:, d:2 * d], prof, cat[:, 2 * d:3 * d]]
    rates = (C.c_float * T)(0.3, 0.2, 0.5, 0.1)
    tab = lambda ts: ((C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts]),
                      (C.c_int64 * len(ts))(*[_ld(t) if t is not None else 0 for t in ts]))
    src_cat = torch.zeros(rows, 3 * d, device=DEV); src_cat[::5] = rn((rows + 4) // 5, 3 * d)   # what a scatter would have left
    srcs = [src_cat[:, 0:d], src_cat[:, d:2 * d], None, src_cat[:, 2 * d:3 * d]]
    # reference: accumulate into a copy of the sources (profile term: into zeros)
    ref_cat = src_cat.clone(); ref_prof = torch.zeros(rows, d, device=DEV)
    d_ref = [ref_cat[:, 0:d], ref_cat[:, d:2 * d], ref_prof, ref_cat[:, 2 * d:3 * d]]
    npt, nl = tab(terms); dp, dl = tab(d_ref)
    _lib.call("llmrec_fuse_bwd_f32", rows, d, _p(dout), _ld(dout), T, npt, nl, rates, dp, dl, 1, 2, 0.01, None)
    out_cat = torch.full((rows, 3 * d), 9.0, device=DEV); out_prof = torch.full((rows, d), 9.0, device=DEV)   # garbage: must be overwritten
    d_out = [out_cat[:, 0:d], out_cat[:, d:2 * d], out_prof, out_cat[:, 2 * d:3 * d]]
    dp2, dl2 = tab(d_out); sp, sl = tab(srcs)
    _lib.call("llmrec_fuse_bwd_src_f32", rows, d, _p(dout), _ld(dout), T, npt, nl, rates, dp2, dl2, sp, sl, 2, 0.01, None)
    torch.cuda.synchronize()
    assert torch.equal(out_cat, ref_cat) and torch.equal(out_prof, ref_prof)

    # zero_rows: two problems sharing index vectors, n_valid < B
    U, I, B, nv = 50, 70, 32, 20
    Eu, Ei = rn(U, d), rn(I, d)
    dEu = [torch.ones(U, d, device=DEV) for _ in range(2)]; dEi = [torch.ones(I, 2 * d, device=DEV) for _ in range(2)]
    users = torch.randperm(U, generator=g, device=DEV)[:B].to(torch.int64)
    pos = torch.randint(0, I, (B,), generator=g, device=DEV); neg = torch.randint(0, I, (B,), generator=g, device=DEV)
    n_valid = torch.tensor([nv], dtype=torch.int32, device=DEV)
    probs = (BprProblem * 2)()
    for i in range(2):
        probs[i].Eu, probs[i].ldu, probs[i].Ei, probs[i].ldi = Eu.data_ptr(), d, Ei.data_ptr(), d
        tgt = dEi[i][:, d:2 * d]                                             # a column slice: ld = 2 d
        probs[i].dEu, probs[i].lddu, probs[i].dEi, probs[i].lddi = dEu[i].data_ptr(), d, tgt.data_ptr(), 2 * d
        probs[i].g_mf, probs[i].g_emb = 1.0, 1.0
    _lib.call("llmrec_bpr_multi_zero_rows_f32", 2, probs, d, _p(users), _p(pos), _p(neg), B, _p(n_valid), None)
    torch.cuda.synchronize()
    for i in range(2):
        want_u = torch.ones(U, d, device=DEV); want_u[users[:nv]] = 0
        want_i = torch.ones(I, 2 * d, device=DEV); want_i[pos[:nv], d:] = 0; want_i[neg[:nv], d:] = 0
        assert torch.equal(dEu[i], want_u) and torch.equal(dEi[i], want_i)


def test_fuse_pair_launches_equal_the_single_launches(ops):
    """llmrec_fuse_fwd_multi_f32 / llmrec_fuse_bwd_src_multi_f32 (user side + item side in one launch) against the
    per-side entry points, bit for bit; different row counts and term counts per side, a NULL source term."""
    import ctypes as C
    from llmrec_amd import _lib
    from llmrec_amd.ops import _p, _ld, FuseFwdProblem, FuseBwdProblem
    g = torch.Generator(device=DEV); g.manual_seed(15)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    d = 64
    tab = lambda ts: ((C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts]),
                      (C.c_int64 * len(ts))(*[_ld(t) if t is not None else 0 for t in ts]))
    sides = []
    for rows, n_mean, n_norm in ((777, 3, 8), (4101, 4, 5)):
        cat = rn(rows, n_norm * d); cat[3] = 0.0
        sides.append({"rows": rows, "means": [rn(rows, d) for _ in range(n_mean)], "norms": [cat[:, k * d:(k + 1) * d] for k in range(n_norm)],
                      "rates": (C.c_float * n_norm)(*[0.1 * (k + 1) for k in range(n_norm)]), "dout": rn(rows, d),
                      "srcs": [rn(rows, d) if k % 3 else None for k in range(n_norm)]})
    keep = []
    # forward
    fwd = (FuseFwdProblem * 2)(); outs_multi, outs_single = [], []
    for pr, sd in zip(fwd, sides):
        mp, ml = tab(sd["means"]); npt, nl = tab(sd["norms"]); keep += [mp, ml, npt, nl]
        out = torch.empty(sd["rows"], d, device=DEV); ref = torch.empty(sd["rows"], d, device=DEV)
        outs_multi.append(out); outs_single.append(ref)
        pr.rows, pr.mean_scale, pr.n_mean, pr.n_norm = sd["rows"], 1.0 / len(sd["means"]), len(sd["means"]), len(sd["norms"])
        pr.mean_terms, pr.mean_ld = C.cast(mp, C.c_void_p), C.cast(ml, C.c_void_p)
        pr.norm_terms, pr.norm_ld, pr.rates = C.cast(npt, C.c_void_p), C.cast(nl, C.c_void_p), C.cast(sd["rates"], C.c_void_p)
        pr.out, pr.ldo = out.data_ptr(), d
        _lib.call("llmrec_fuse_fwd_f32", sd["rows"], d, 1.0 / len(sd["means"]), len(sd["means"]), mp, ml, len(sd["norms"]), npt, nl, sd["rates"], _p(ref), d, None)
    _lib.call("llmrec_fuse_fwd_multi_f32", 2, fwd, d, None)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs_multi, outs_single))
    # backward (source mode)
    bwd = (FuseBwdProblem * 2)(); d_multi, d_single = [], []
    for pr, sd in zip(bwd, sides):
        n = len(sd["norms"])
        dm = [torch.full((sd["rows"], d), 7.0, device=DEV) for _ in range(n)]; ds = [torch.full((sd["rows"], d), 7.0, device=DEV) for _ in range(n)]
        d_multi.append(dm); d_single.append(ds)
        npt, nl = tab(sd["norms"]); dp, dl = tab(dm); dps, dls = tab(ds); sp, sl = tab(sd["srcs"]); keep += [npt, nl, dp, dl, dps, dls, sp, sl]
        pr.rows, pr.dOut, pr.lddo, pr.n_norm = sd["rows"], sd["dout"].data_ptr(), d, n
        pr.norm_terms, pr.norm_ld, pr.rates = C.cast(npt, C.c_void_p), C.cast(nl, C.c_void_p), C.cast(sd["rates"], C.c_void_p)
        pr.d_terms, pr.d_ld, pr.src_terms, pr.src_ld = C.cast(dp, C.c_void_p), C.cast(dl, C.c_void_p), C.cast(sp, C.c_void_p), C.cast(sl, C.c_void_p)
        pr.n_reg_terms, pr.reg_two_coef = 2, 0.01
        _lib.call("llmrec_fuse_bwd_src_f32", sd["rows"], d, _p(sd["dout"]), d, n, npt, nl, sd["rates"], dps, dls, sp, sl, 2, 0.01, None)
    _lib.call("llmrec_fuse_bwd_src_multi_f32", 2, bwd, d, None)
    torch.cuda.synchronize()
    for dm, ds in zip(d_multi, d_single):
        assert all(torch.equal(a, b) for a, b in zip(dm, ds))
    # row flags: rows whose flag is 0 have zero dOut and zero sources (as the loss backward leaves them) - the flagged launch
    # skips their reads and must write the same bits (regulariser term on the first two streams, +0 elsewhere)
    flags = []
    for sd in sides:
        touched = torch.rand(sd["rows"], generator=g, device=DEV) < 0.1
        touched[3] = True                                                    # (the all-zero cat row stays on the general path once)
        sd["dout"][~touched] = 0.0
        for t in sd["srcs"]:
            if t is not None:
                t[~touched] = 0.0
        flags.append(touched.to(torch.uint8).contiguous())
    _lib.call("llmrec_fuse_bwd_src_multi_f32", 2, bwd, d, None)
    torch.cuda.synchronize()
    plain = [[t.clone() for t in dm] for dm in d_multi]
    for dm in d_multi:
        for t in dm:
            t.fill_(7.0)
    for pr, fl in zip(bwd, flags):
        pr.row_flags = fl.data_ptr()
    _lib.call("llmrec_fuse_bwd_src_multi_f32", 2, bwd, d, None)
    torch.cuda.synchronize()
    for dm, pm in zip(d_multi, plain):
        assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(dm, pm))
    # stamped flags: active = LLMREC_ROW_STAMP(counter) = counter % 255 + 1; every other byte value (stale stamps) is "not touched"
    counter = torch.tensor([6 + 255], dtype=torch.int32, device=DEV)
    stamped = [torch.where(fl != 0, torch.full_like(fl, 7), torch.full_like(fl, 3)) for fl in flags]
    for dm in d_multi:
        for t in dm:
            t.fill_(7.0)
    for pr, fl in zip(bwd, stamped):
        pr.row_flags, pr.row_stamp = fl.data_ptr(), counter.data_ptr()
    _lib.call("llmrec_fuse_bwd_src_multi_f32", 2, bwd, d, None)
    torch.cuda.synchronize()
    for dm, pm in zip(d_multi, plain):
        assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(dm, pm))


def test_weighted_column_sums_in_groups(ops):
    """llmrec_weighted_colsum_f32: out_g[j] (+)= sum_r w[r] X[r][gw g + j]; groups sharing a destination are summed; deterministic."""
    import ctypes as C
    from llmrec_amd import _lib
    from llmrec_amd.ops import _p, _ld
    g = torch.Generator(device=DEV); g.manual_seed(21)
    for rows, n_groups, gw in ((13187, 7, 64), (301, 3, 16), (5, 8, 64), (0, 2, 64)):
        big = torch.randn(max(rows, 1), n_groups * gw + 8, generator=g, device=DEV)[:rows]
        X = big[:, 4:4 + n_groups * gw]                                      # a column slice (ld > d, 16-byte misaligned)
        w = torch.rand(rows, generator=g, device=DEV)
        shared = torch.full((gw,), 3.0, device=DEV)                          # groups 1.. share one destination (the 5 attribute streams)
        first = torch.full((gw,), 5.0, device=DEV)
        outs = [first] + [shared] * (n_groups - 1)
        gp = (C.c_void_p * n_groups)(*[t.data_ptr() for t in outs])
        ws = torch.empty(_lib.query("llmrec_weighted_colsum_workspace_bytes", n_groups * gw), dtype=torch.uint8, device=DEV)
        _lib.call("llmrec_weighted_colsum_f32", rows, n_groups, gw, _p(X), _ld(X), _p(w), gp, 0, _p(ws), ws.numel(), None)
        want = (w.double()[:, None] * X.double()).sum(0).reshape(n_groups, gw)
        assert torch.allclose(first.double(), want[0], rtol=0, atol=2e-6 * max(1.0, float(want.abs().max())))
        assert torch.allclose(shared.double(), want[1:].sum(0), rtol=0, atol=2e-6 * max(1.0, float(want.abs().max())))
        a = shared.clone()
        first.fill_(5.0); shared.fill_(3.0)
        _lib.call("llmrec_weighted_colsum_f32", rows, n_groups, gw, _p(X), _ld(X), _p(w), gp, 0, _p(ws), ws.numel(), None)
        assert torch.equal(a, shared)                                        # run-to-run identical
        _lib.call("llmrec_weighted_colsum_f32", rows, n_groups, gw, _p(X), _ld(X), None, gp, 1, _p(ws), ws.numel(), None)   # accumulate, w = ones
        assert torch.allclose(first.double(), want[0] + X.double().sum(0).reshape(n_groups, gw)[0], rtol=0, atol=4e-6 * max(1.0, float(want.abs().max())))


def test_projection_of_a_pre_propagated_operand(ops):
    """llmrec_linear_problem_t.bias_scale: (A F) W^T + (A 1) b^T equals A (F W^T + 1 b^T) - the identity the fused step's
    pre-propagated item-side operands rest on (reference Models.py:145-157)."""
    g = torch.Generator(device=DEV); g.manual_seed(31)
    U_, I_, K, d = 700, 900, 256, 64
    rows = torch.randint(0, U_, (6000,), generator=g, device=DEV); cols = torch.randint(0, I_, (6000,), generator=g, device=DEV)
    key = torch.unique(rows * I_ + cols); rows, cols = key // I_, key % I_
    gr = ops.BipartiteGraph.from_edges(rows, cols, U_, I_)
    F_ = torch.randn(I_, K, generator=g, device=DEV); W = torch.randn(d, K, generator=g, device=DEV) / 16; b = torch.randn(d, generator=g, device=DEV)
    AF = torch.empty(U_, K, device=DEV)
    for c0 in range(0, K, 64):
        ops.spmm_raw(gr.ui.fwd, F_[:, c0:c0 + 64], out=AF[:, c0:c0 + 64])
    c_u = ops.spmm_raw(gr.ui.fwd, torch.ones(I_, 4, device=DEV))[:, 0].contiguous()
    for precision in ("bf16x3", "f32"):
        out = torch.empty(U_, d, device=DEV)
        ops.linear_fwd_grouped([(AF, W, b, out, c_u)], d, precision=precision)
        P = F_.double() @ W.double().t() + b.double()
        A = torch.zeros(U_, I_, dtype=torch.float64, device=DEV); A[rows, cols] = 1.0
        want = gr.s_u.double()[:, None] * (A @ P)
        assert float((out.double() - want).abs().max() / want.abs().max()) < 3e-6, precision


def test_weight_gradient_with_row_weighted_bias_gradient(ops):
    """llmrec_wgrad_problem_t.db_row_weight: dW as ever, db = sum_r w[r] dY[r] (the bias gradient of a projection with bias_scale),
    several problems per target with different weights, ragged M; and the refusal where the organisation does not serve it."""
    g = torch.Generator(device=DEV); g.manual_seed(41)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    relmax = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    d, K = 64, 256
    Ms = [1301, 517, 64]
    dY = rn(sum(Ms), 3 * d)
    pairs, want_w, want_b = [], torch.zeros(d, K, dtype=torch.float64, device=DEV), torch.zeros(d, dtype=torch.float64, device=DEV)
    r0 = 0
    for j, M in enumerate(Ms):
        X = rn(M, K); w = torch.rand(M, generator=g, device=DEV) if j != 1 else None
        dy = dY[r0:r0 + M, d:2 * d]; r0 += M                                       # a column slice, ld = 3 d
        pairs.append((dy, X, w))
        want_w += dy.double().t() @ X.double()
        want_b += (dy.double() * (w.double()[:, None] if w is not None else 1.0)).sum(0)
    dW = torch.empty(d, K, device=DEV); db = torch.empty(d, device=DEV)
    ops.linear_wgrad_multi([(pairs, dW, db, False)])
    assert relmax(dW, want_w) < 3e-6 and relmax(db, want_b) < 3e-6
    a = db.clone()
    ops.linear_wgrad_multi([(pairs, dW, db, False)])
    assert torch.equal(a, db)
    ops.linear_wgrad_grouped(pairs, dW, db, False, precision="bf16x3")                 # the single-target entry routes to the same launch
    assert relmax(db, want_b) < 3e-6
    with pytest.raises(RuntimeError):                                                  # the exact-fp32 kernels sum dY unweighted: refused
        ops.linear_wgrad_grouped(pairs, dW, db, False, precision="f32")


def test_weight_gradient_launch_with_the_adamw_update_inside(ops):
    """llmrec_linear_wgrad_multi_adamw_bf16x3 = llmrec_linear_wgrad_multi_bf16x3 followed by llmrec_adamw_multi_f32 on (W, b) of every
    target: gradients, parameters and both moments bit for bit, over two steps (non-zero moments in the second), one target without a
    bias gradient, one with a row-weighted one."""
    N = 64
    shapes = [(3, 700, 256), (1, 900, 128), (1, 333, 384)]            # (pairs, M, K)
    rngs = np.random.default_rng(22)
    data = []
    for n_pairs, M, K in shapes:
        data.append([(torch.tensor((rngs.standard_normal((M, N)) * 1e-3).astype(np.float32), device=DEV),
                      torch.tensor(rngs.standard_normal((M, K)).astype(np.float32), device=DEV),
                      torch.tensor(rngs.random(M).astype(np.float32), device=DEV)) for _ in range(n_pairs)])

    def run(fused):
        torch.manual_seed(0)
        rng2 = np.random.default_rng(23)
        lin = []
        for t, (n_pairs, M, K) in enumerate(shapes):
            W = torch.tensor((rng2.standard_normal((N, K)) * 0.05).astype(np.float32), device=DEV)
            b = None if t == 1 else torch.tensor((rng2.standard_normal(N) * 0.05).astype(np.float32), device=DEV)
            W.grad = torch.zeros_like(W)
            if b is not None:
                b.grad = torch.zeros_like(b)
            lin.append((W, b))
        params = [p_ for W, b in lin for p_ in (W, b) if p_ is not None]
        opt = ops.FusedAdamW(params, lr=1e-3)
        for step in range(2):
            targets = []
            for t, (W, b) in enumerate(lin):
                pairs = [(dY * (1.0 + step), X, w) if t == 2 else (dY * (1.0 + step), X) for dY, X, w in data[t]]
                targets.append((pairs, W.grad, b.grad if b is not None else None, False))
            opt.advance()
            if fused:
                ops.linear_wgrad_multi(targets, update=(opt, lin))
            else:
                ops.linear_wgrad_multi(targets)
                opt.step_params(params)
        torch.cuda.synchronize()
        out = []
        for p_ in params:
            out += [p_.detach().cpu(), p_.grad.cpu(), opt.state[p_][0].cpu(), opt.state[p_][1].cpu()]
        return out
    a, b = run(False), run(True)
    assert len(a) == len(b) == 4 * 5
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    assert float(a[2].abs().max()) > 0 and float(a[3].abs().max()) > 0   # (moments were written)



@pytest.mark.parametrize("d,weighted", [(64, False), (64, True), (128, False), (448, True)])
def test_spmm_operand_row_mask_and_output_flags(ops, d, weighted):
    """llmrec_spmm_epilogue_t operand sparsity: with the promised zeros in place (rows of X that are not active are all-zero) the masked
    product equals the unmasked one; rows that are not active are NOT READ (poisoning them changes nothing); y_row_flag marks exactly
    the rows with an active neighbour or a flagged Z row in the short / wavefront buckets and every row of the long-row buckets; stale
    byte values count as not active. All row buckets (hub rows past the split threshold), the softmax-backward epilogue included."""
    import scipy.sparse as sp
    rng = np.random.default_rng(41)
    torch.manual_seed(41)                                      # (the torch.rand / randn operands below: the same data whatever ran before)
    n_rows, n_cols, stamp = 3000, 2500, 77
    degs = rng.integers(0, 30, size=n_rows); degs[5] = 2400; degs[6] = 0; degs[7] = 700; degs[8] = 150
    rows, cols = rand_graph(rng, n_rows, n_cols, degs)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, n_rows, n_cols)
    cs = (torch.rand(n_cols, device=DEV) + 0.5) if weighted else None
    a = ops.Csr(n_rows, n_cols, rp, ci, None, torch.rand(n_rows, device=DEV) + 0.5, cs, {})
    act = torch.rand(n_cols, device=DEV) < 0.02
    mask = torch.where(act, torch.full((n_cols,), stamp, dtype=torch.uint8, device=DEV),
                       torch.randint(0, 60, (n_cols,), device=DEV).to(torch.uint8))          # stale values everywhere else
    X = torch.randn(n_cols, d, device=DEV); X[~act] = 0.0
    Z = torch.zeros(n_rows, d, device=DEV)
    zrows = torch.tensor([6, 11, 2999], device=DEV); Z[zrows] = torch.randn(3, d, device=DEV)
    zflag = torch.randint(0, 60, (n_rows,), device=DEV).to(torch.uint8); zflag[zrows] = stamp
    S = torch.softmax(torch.randn(n_rows, d, device=DEV), dim=1)
    whole = d <= 128                                                        # (the softmax epilogues need the whole row)
    op = ops.EPI_SOFTMAX_BWD if whole else ops.EPI_NONE
    ref = ops.spmm_raw(a, X, epilogue=ops.spmm_epilogue(op, 0.5, Z, S if whole else None))
    Xp = X.clone(); Xp[~act] = float("nan")                                 # never read
    yflag = zflag.clone()                                                   # (aliased with z_row_flag, as the row-sharded step uses it)
    got = ops.spmm_raw(a, Xp, epilogue=ops.spmm_epilogue(op, 0.5, Z, S if whole else None, x_row_mask=mask, x_mask_active=stamp,
                                                          y_row_flag=yflag, z_row_flag=yflag))
    assert bool(torch.isfinite(got).all()) and torch.equal(got, ref)
    A = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n_rows, n_cols))
    hit = np.asarray(A @ act.cpu().numpy().astype(np.float64)).ravel() > 0
    hit[zrows.cpu().numpy()] = True
    yf = yflag.cpu().numpy()
    assert set(np.unique(yf)) <= {0, stamp}
    sw, pl = a.plan_for(d, whole_row=whole and op != ops.EPI_NONE)
    short = degs <= pl.t_wave                                               # lane-group and wavefront rows: exact flags
    assert np.array_equal(yf[short] == stamp, hit[short])
    assert np.all(yf[~short] == stamp) and np.all(yf[hit] == stamp)         # long rows: always flagged; no active row is ever missed
    assert np.abs(got.cpu().numpy()[yf == 0]).max() == 0.0                  # an unflagged row is a zero row
    # the same product behind a row GATE built from the active columns' adjacency (llmrec_mark_neighbours_u8 on the transposed CSR)
    # plus the flagged Z rows (llmrec_mark_rows_u8): gated-out rows are written as zeros unread, the result does not change
    import ctypes
    from llmrec_amd import _lib
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    rpT, ciT, _ = ops.csr_from_coo(torch.tensor(cols).to(DEV), torch.tensor(rows).to(DEV), None, n_cols, n_rows)
    gate = torch.randint(0, 60, (n_rows,), device=DEV).to(torch.uint8)
    act_ids = torch.nonzero(act).flatten().to(torch.int64)
    ids = torch.cat([act_ids, torch.tensor([-1], device=DEV)])               # (negative ids are skipped)
    _lib.call("llmrec_mark_rows_u8", zrows.numel(), p_(zrows), stamp, p_(gate), st)
    _lib.call("llmrec_mark_neighbours_u8", ids.numel(), p_(ids), p_(rpT), p_(ciT), stamp, p_(gate), st)
    torch.cuda.synchronize()
    assert np.array_equal(gate.cpu().numpy() == stamp, hit)
    got2 = ops.spmm_raw(a, Xp, epilogue=ops.spmm_epilogue(op, 0.5, Z, S if whole else None, x_row_mask=mask, x_mask_active=stamp,
                                                           z_row_flag=gate, y_row_gate=gate))
    assert torch.equal(got2, ref)


def test_mark_rows_and_neighbours_at_launch_sizes_beyond_one_grid_cap(ops):
    """llmrec_mark_rows_u8 / llmrec_mark_neighbours_u8 with many ids (more blocks than the row kernels' usual grid cap), duplicate and
    negative ids, hub rows and empty rows: exactly the listed rows / their columns are marked, nothing else changes."""
    import ctypes
    import scipy.sparse as sp
    from llmrec_amd import _lib
    rng = np.random.default_rng(43)
    n_rows, n_cols = 20000, 30000
    degs = rng.integers(0, 12, size=n_rows); degs[17] = 25000; degs[18] = 0
    rows, cols = rand_graph(rng, n_rows, n_cols, degs)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, n_rows, n_cols)
    ids_np = np.concatenate([rng.integers(0, n_rows, size=9000), [17, 18, -1, -1, 17]])
    ids = torch.tensor(ids_np, dtype=torch.int64, device=DEV)
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    flags = torch.full((n_cols,), 9, dtype=torch.uint8, device=DEV)
    _lib.call("llmrec_mark_neighbours_u8", ids.numel(), p_(ids), p_(rp), p_(ci), 200, p_(flags), st)
    rflags = torch.full((n_rows,), 9, dtype=torch.uint8, device=DEV)
    _lib.call("llmrec_mark_rows_u8", ids.numel(), p_(ids), 201, p_(rflags), st)
    torch.cuda.synchronize()
    sel = np.zeros(n_rows, dtype=bool); sel[ids_np[ids_np >= 0]] = True
    A = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(n_rows, n_cols))
    want = np.asarray(A.T @ sel.astype(np.float64)).ravel() > 0
    assert np.array_equal(flags.cpu().numpy(), np.where(want, 200, 9).astype(np.uint8))
    assert np.array_equal(rflags.cpu().numpy(), np.where(sel, 201, 9).astype(np.uint8))


@pytest.mark.parametrize("d", [64, 128, 20])
def test_softmax_backward_of_listed_rows(ops, d):
    """llmrec_softmax_rows_bwd_listed_f32 against the dense launches it replaces in the listed rows (axpy, softmax backward, row scale);
    rows that are not listed keep their contents; negative and duplicate ids."""
    import ctypes
    from llmrec_amd import _lib
    g = torch.Generator(device=DEV); g.manual_seed(7)
    n = 5000
    Y = torch.softmax(torch.randn(n, d, generator=g, device=DEV), dim=1)
    dY = torch.randn(n, d, generator=g, device=DEV) * 1e-3
    ps = torch.rand(n, generator=g, device=DEV) + 0.5
    ids = torch.cat([torch.randint(0, n, (700,), generator=g, device=DEV), torch.tensor([-1, 3, 3], device=DEV)]).to(torch.int64)
    out = torch.full((n, d), 5.0, device=DEV)
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("llmrec_softmax_rows_bwd_listed_f32", ids.numel(), p_(ids), d, 0.25, p_(Y), d, p_(dY), d, p_(ps), p_(out), d,
              torch.cuda.current_stream().cuda_stream)
    gd = (0.25 * dY).double(); yd = Y.double()
    want = (ps.double()[:, None] * (yd * (gd - (gd * yd).sum(1, keepdim=True)))).float()
    sel = torch.zeros(n, dtype=torch.bool, device=DEV); sel[ids[ids >= 0]] = True
    assert rel_err(out[sel].cpu(), want[sel].cpu()) < 2e-6
    assert bool((out[~sel] == 5.0).all())


def test_spmm_of_listed_rows_and_of_needed_rows(ops):
    """ops.spmm_listed (a plan over an explicit row list + rows_listed_only): exactly the listed rows of A X are written - short rows,
    wavefront / block rows and a hub past the split threshold - equal to the full product's rows to summation-order rounding; y_row_needed: the short-row
    range skips the rows that are not marked (their previous contents stay), the forward softmax epilogue included."""
    rng = np.random.default_rng(47)
    torch.manual_seed(47)
    n_rows, n_cols, d = 4000, 3000, 64
    degs = rng.integers(0, 30, size=n_rows); degs[5] = 2900; degs[6] = 0; degs[7] = 900; degs[8] = 200; degs[9] = 40
    rows, cols = rand_graph(rng, n_rows, n_cols, degs)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, n_rows, n_cols)
    a = ops.Csr(n_rows, n_cols, rp, ci, None, torch.rand(n_rows, device=DEV) + 0.5, None, {})
    X = torch.randn(n_cols, d, device=DEV)
    full = ops.spmm_raw(a, X)
    listed = torch.unique(torch.cat([torch.tensor([5, 6, 7, 8, 9, 0, n_rows - 1], device=DEV), torch.randint(0, n_rows, (300,), device=DEV)]))
    out = torch.full((n_rows, d), 3.0, device=DEV)
    ops.spmm_listed(a, X, listed, out)
    sel = torch.zeros(n_rows, dtype=torch.bool, device=DEV); sel[listed] = True
    # (listed short rows run in the wavefront bucket: another summation order than the short-row range of the full product)
    assert rel_err(out[sel].cpu(), full[sel].cpu()) < 2e-6 and bool((out[~sel] == 3.0).all())
    # needed rows, softmax epilogue
    stamp = 9
    need = torch.where(torch.rand(n_rows, device=DEV) < 0.3, torch.full((n_rows,), stamp, dtype=torch.uint8, device=DEV),
                       torch.randint(10, 50, (n_rows,), device=DEV).to(torch.uint8))
    ref = ops.spmm_raw(a, X, epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX))
    got = torch.full((n_rows, d), 3.0, device=DEV)
    ops.spmm_raw(a, X, out=got, epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX, y_row_needed=need, x_mask_active=stamp))
    nd = (need == stamp)
    sw, pl = a.plan_for(d, whole_row=True)
    short = torch.tensor(degs <= 32, device=DEV)                      # LLMREC_SPMM_LONG_ROW: the rows of the short-row range
    assert torch.equal(got[nd], ref[nd])
    assert bool((got[~nd & short] == 3.0).all()) and torch.equal(got[~nd & ~short], ref[~nd & ~short])


def test_batch_reach_rows_lists_the_batch_users_and_their_items_neighbours(ops):
    """llmrec_batch_reach_rows: ascending list of {batch users} U {users adjacent to a batch item}, device count, scratch left all-zero;
    invalid (negative / out-of-range) ids and the entries past n_valid are ignored; an empty batch gives an empty list."""
    rng = np.random.default_rng(8)
    U_, I_ = 5000, 700
    rows = rng.integers(0, U_, size=9000); cols = (rng.zipf(1.6, size=9000) % I_)
    key = np.unique(rows * I_ + cols); rows, cols = key // I_, key % I_
    gr = ops.BipartiteGraph.from_edges(torch.tensor(rows, device=DEV), torch.tensor(cols, device=DEV), U_, I_)
    by_item = gr.iu.fwd                                                       # rows = items, columns = users
    adj = {}
    for u, i in zip(rows.tolist(), cols.tolist()):
        adj.setdefault(i, set()).add(u)
    flags = torch.zeros(U_, dtype=torch.uint8, device=DEV)
    lst = torch.full((U_ + 32,), -7, dtype=torch.int32, device=DEV); n = torch.zeros(1, dtype=torch.int32, device=DEV)
    for B, nv in ((64, 64), (200, 137), (16, 0), (1, 1)):
        users = rng.integers(0, U_, size=B); pos = rng.integers(0, I_, size=B); neg = rng.integers(0, I_, size=B)
        if B >= 64:
            users[3] = -1; pos[5] = -1; neg[7] = I_ + 3                        # skipped
        nvt = torch.tensor([nv], dtype=torch.int32, device=DEV)
        ops.batch_reach_rows(torch.tensor(users, device=DEV), torch.tensor(pos, device=DEV), torch.tensor(neg, device=DEV), nvt, by_item, flags, lst, n)
        want = set()
        for b in range(nv):
            if 0 <= users[b] < U_: want.add(int(users[b]))
            for it in (pos[b], neg[b]):
                if 0 <= it < I_: want |= adj.get(int(it), set())
        got_n = int(n)
        assert got_n == len(want), (B, nv, got_n, len(want))
        assert lst[:got_n].cpu().tolist() == sorted(want)
        assert int(flags.max()) == 0
        pad = lst[got_n:(got_n + 15) // 16 * 16 + 16].cpu()
        assert bool((pad == 0).all())                                         # what a 16-wide tile may read past the end is defined


def test_weight_gradient_over_a_row_list_equals_the_dense_launch(ops):
    """llmrec_wgrad_problem_t.row_list: dY is zero outside the listed rows, and the launch that streams only the listed rows gives the
    dense launch's dW / db (row-weighted) up to fp32 summation order; list lengths 0, 1, ragged, all rows; a length far from
    rows_expected (the geometry hint) changes nothing; unlisted rows of dY are never read (poisoned with NaN here)."""
    g = torch.Generator(device=DEV); g.manual_seed(43)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    relmax = lambda a, b: float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))
    d, K, M = 64, 384, 3001
    Xs = [rn(M, K) for _ in range(3)]
    Xd = rn(1200, 256)                                                        # a dense target in the same launch
    w = torch.rand(M, generator=g, device=DEV)
    for n_act, expected in ((0, 0), (1, 0), (517, 600), (1300, 100), (M, 0), (2000, M)):
        ids = torch.sort(torch.randperm(M, generator=g, device=DEV)[:n_act]).values.to(torch.int32)
        lst = torch.zeros(M + 32, dtype=torch.int32, device=DEV); lst[:n_act] = ids
        n = torch.tensor([n_act], dtype=torch.int32, device=DEV)
        dY = torch.zeros(M, 3 * d, device=DEV)
        dY[ids.long()] = rn(n_act, 3 * d) * 1e-3
        dYd = rn(1200, d) * 1e-3
        dense = [(dY[:, k * d:(k + 1) * d], Xs[k], w) for k in range(3)]
        dW0, db0 = torch.empty(d, K, device=DEV), torch.empty(d, device=DEV)
        dWd0, dbd0 = torch.empty(d, 256, device=DEV), torch.empty(d, device=DEV)
        ops.linear_wgrad_multi([(dense, dW0, db0, False), ([(dYd, Xd)], dWd0, dbd0, False)])
        poisoned = dY.clone()
        mask = torch.ones(M, dtype=torch.bool, device=DEV); mask[ids.long()] = False
        poisoned[mask] = float("nan")
        listed = [(poisoned[:, k * d:(k + 1) * d], Xs[k], w, (lst, n, expected)) for k in range(3)]
        dW1, db1 = torch.full((d, K), 7.0, device=DEV), torch.full((d,), 7.0, device=DEV)
        dWd1, dbd1 = torch.empty(d, 256, device=DEV), torch.empty(d, device=DEV)
        ops.linear_wgrad_multi([(listed, dW1, db1, False), ([(dYd, Xd)], dWd1, dbd1, False)])
        want_w = sum(dY[:, k * d:(k + 1) * d].double().t() @ Xs[k].double() for k in range(3))
        want_b = sum((dY[:, k * d:(k + 1) * d].double() * w.double()[:, None]).sum(0) for k in range(3))
        if n_act == 0:
            assert float(dW1.abs().max()) == 0.0 and float(db1.abs().max()) == 0.0
        else:
            assert relmax(dW1, want_w) < 3e-6 and relmax(db1, want_b) < 3e-6, (n_act, relmax(dW1, want_w), relmax(db1, want_b))
            assert relmax(dW1, dW0) < 2e-6 and relmax(db1, db0) < 2e-6
        assert torch.equal(dWd1, dWd0) or relmax(dWd1, dWd0) < 2e-6          # the dense target beside it
        a = dW1.clone()
        ops.linear_wgrad_multi([(listed, dW1, db1, False), ([(dYd, Xd)], dWd1, dbd1, False)])
        assert torch.equal(a, dW1)                                            # deterministic
    with pytest.raises(RuntimeError):                                         # the single-target exact-fp32 entry does not serve lists
        ops.linear_wgrad_grouped([(dY[:, :d], Xs[0], None, (lst, n, 0))], dW1, db1, False, precision="f32")
