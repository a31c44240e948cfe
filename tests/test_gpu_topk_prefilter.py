"""llmrec_score_topk_mode_f32: the bf16 sweep with exact re-ranking and verification (LLMREC_TOPK_MODE_PREFILTER; user tiles whose
verification fails are redone by the exact sweep) must return the SAME BITS - item lists and scores - as the exact-fp32 sweep
(LLMREC_TOPK_MODE_EXACT_SWEEP), which the other top-K tests hold to the oracle and to the reference's lists: random tables, score distributions that keep the filter loose (near-identical rows, softmax-shaped rows), exact ties
(integer-valued embeddings: ordered by item id), orders that keep the threshold rising, fewer than K candidates, every supported width,
the Netflix shape with its split user tiles, and 10^6 items."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from llmrec_amd import ops as _ops
    return _ops


def _train_csr(ops, U, I, rng, max_deg):
    degs = rng.integers(0, max_deg + 1, size=U)
    rows = np.repeat(np.arange(U), degs)
    cols = np.concatenate([rng.choice(I, size=int(dg), replace=False) for dg in degs] + [np.zeros(0, dtype=np.int64)]).astype(np.int64)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, U, I)
    return ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())


LAST = {}


def _both(ops, Eu, Ei, q, train, K):
    i0, s0 = ops.score_topk(Eu, Ei, q, train, K, mode="exact")
    LAST.clear()
    i1, s1 = ops.score_topk(Eu, Ei, q, train, K, mode="prefilter", stats=LAST)
    torch.cuda.synchronize()
    return i0, s0, i1, s1


def _assert_same(i0, s0, i1, s1, what):
    same_i = torch.equal(i0, i1)
    same_s = torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    if not (same_i and same_s):
        bad = (i0 != i1).nonzero()
        raise AssertionError("%s: lists differ at %d positions (first %s), scores equal: %s" % (what, bad.shape[0], bad[:3].tolist(), same_s))


@pytest.mark.parametrize("U,I,d,K", [(150, 1000, 64, 50), (70, 130, 16, 50), (33, 64, 64, 20), (200, 777, 128, 64), (10, 45, 64, 50),
                                     (300, 5000, 48, 50), (97, 3001, 80, 10), (130, 2500, 112, 50), (50, 4000, 32, 1), (2000, 9000, 64, 50)])
def test_prefilter_equals_exact_on_random_tables(ops, U, I, d, K):
    rng = np.random.default_rng(U * 7 + I + d)
    Eu = torch.tensor((rng.standard_normal((U, d)) * 0.4).astype(np.float32)).to(DEV)
    Ei = torch.tensor((rng.standard_normal((I, d)) * 0.4).astype(np.float32)).to(DEV)
    train = _train_csr(ops, U, I, rng, min(40, I // 2))
    q = torch.tensor(rng.permutation(U)).to(DEV)
    _assert_same(*_both(ops, Eu, Ei, q, train, K), what="random")
    if K <= 50 and I >= 1000:
        assert LAST["fallback_tiles"] == 0, LAST                   # well-separated scores: the verification holds everywhere
        # rows of at most 40 train items: no long row; a block whose 16 rows hold more than 192 items sweeps them ALL as bitmaps (tile-major slice)
        assert LAST["bitmap_rows"] % 16 == 0 and LAST["bitmap_rows"] > 0, LAST
        sparse = _train_csr(ops, U, I, rng, 6)                     # ~3 items per row: every block walks its rows
        _assert_same(*_both(ops, Eu, Ei, q, sparse, K), what="random, sparse train rows")
        assert LAST["bitmap_rows"] == 0, LAST
    _assert_same(*_both(ops, Eu, Ei, q, None, K), what="random, no mask")
    _assert_same(*_both(ops, Eu, Ei, q[:7], train, K), what="7 queries")


@pytest.mark.parametrize("kind", ["near_identical_rows", "softmax_rows", "tiny_spread", "huge_norm_outlier", "integers", "ascending", "few_candidates",
                                  "all_negative", "zeros_and_signs", "strictly_ascending"])
def test_prefilter_equals_exact_where_the_filter_is_loose_or_ties_abound(ops, kind):
    # fixed seeds (hash(str) changes from process to process; 35 of the 1000 seeds it produced leave the "integers" case without a single
    # tile for the exact sweep - all 1000 give bit-identical, reference-equal lists in both modes, swept on the GPU)
    rng = np.random.default_rng({"near_identical_rows": 101, "softmax_rows": 102, "tiny_spread": 103, "huge_norm_outlier": 104, "integers": 105,
                                 "ascending": 106, "few_candidates": 107, "all_negative": 108, "zeros_and_signs": 109, "strictly_ascending": 110}[kind])
    U, I, d, K = 200, 6000, 64, 50
    if kind == "near_identical_rows":        # scores within ~1e-4 relative of one another: the slack admits many false positives
        base = rng.standard_normal(d).astype(np.float32)
        Ei = base[None, :] + (rng.standard_normal((I, d)) * 1e-4).astype(np.float32)
        Eu = np.abs(rng.standard_normal((U, d))).astype(np.float32)
    elif kind == "softmax_rows":             # the trained shape: rows of a softmax layer output plus small normalised terms
        z = rng.standard_normal((I, d)).astype(np.float32) * 0.05
        Ei = (np.exp(z) / np.exp(z).sum(1, keepdims=True)).astype(np.float32)
        zu = rng.standard_normal((U, d)).astype(np.float32) * 0.05
        Eu = (np.exp(zu) / np.exp(zu).sum(1, keepdims=True)).astype(np.float32)
    elif kind == "tiny_spread":              # every score equal up to the last bits
        Ei = np.full((I, d), 0.125, dtype=np.float32); Ei += (rng.integers(0, 3, size=(I, d)) * 2.0 ** -24).astype(np.float32)
        Eu = np.full((U, d), 0.5, dtype=np.float32)
    elif kind == "huge_norm_outlier":        # one item with a norm 1e4 times the others': max ||i|| inflates every user's slack
        Ei = (rng.standard_normal((I, d)) * 0.1).astype(np.float32); Ei[17] *= 1e4
        Eu = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    elif kind == "integers":                 # exact arithmetic in every order: ties everywhere, ranked by item id
        Ei = rng.integers(-2, 3, size=(I, d)).astype(np.float32)
        Eu = rng.integers(-2, 3, size=(U, d)).astype(np.float32)
    elif kind == "ascending":                # item scores grow with the id: the threshold rises all sweep long
        Eu = np.abs(rng.standard_normal((U, d))).astype(np.float32)
        Ei = (np.abs(rng.standard_normal((I, d))) * np.linspace(0.1, 2.0, I)[:, None]).astype(np.float32)
    elif kind == "all_negative":             # every score (and every filter of the pool sweep) below zero: the order-preserving keys of negative floats
        Eu = np.abs(rng.standard_normal((U, d))).astype(np.float32)
        Ei = (-np.abs(rng.standard_normal((I, d))) * 0.3).astype(np.float32)
    elif kind == "zeros_and_signs":          # scores of both signs around exact zeros (+0 / -0 bounds; a third of the items and a few users are all-zero rows)
        Eu = (rng.standard_normal((U, d)) * 0.2).astype(np.float32); Eu[::9] = 0.0
        Ei = (rng.standard_normal((I, d)) * 0.2).astype(np.float32); Ei[::3] = 0.0; Ei[1::7] *= -0.0
    elif kind == "strictly_ascending":       # every item beats all items before it, for every user: every candidate passes every filter, the pools refill
        Eu = (np.abs(rng.standard_normal((U, d))) + 0.5).astype(np.float32)     # each round and a drain follows each round
        Ei = (np.full((I, d), 0.25) * (1.0 + np.arange(I)[:, None] * 1e-3)).astype(np.float32)
    else:                                    # few_candidates: most items are train items
        Ei = rng.standard_normal((I, d)).astype(np.float32); Eu = rng.standard_normal((U, d)).astype(np.float32)
    Eu, Ei = torch.tensor(Eu).to(DEV), torch.tensor(Ei).to(DEV)
    if kind == "few_candidates":
        I = 90; Ei = Ei[:I].contiguous()
        rows = np.repeat(np.arange(U), 60); cols = np.concatenate([rng.choice(I, size=60, replace=False) for _ in range(U)]).astype(np.int64)
        rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, U, I)
        train = ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())
    else:
        train = _train_csr(ops, U, I, rng, 30)
    q = torch.arange(U, device=DEV)
    i0, s0, i1, s1 = _both(ops, Eu, Ei, q, train, K)
    _assert_same(i0, s0, i1, s1, what=kind)
    if kind in ("near_identical_rows", "tiny_spread", "integers"):   # more than 14 items within the slack of the boundary: the exact sweep
        assert LAST["fallback_tiles"] > 0, (kind, LAST)              # must have redone tiles (the path is exercised, not just present)
    print(kind, LAST)
    if kind == "integers":                   # and the tie rule itself: (score desc, item id asc)
        sc, ids = s1.cpu().numpy(), i1.cpu().numpy()
        for r in range(0, U, 17):
            pairs = [(-float(a), int(b)) for a, b in zip(sc[r], ids[r]) if b >= 0]
            assert pairs == sorted(pairs)


def test_prefilter_equals_exact_at_the_netflix_shape_and_at_a_million_items(ops):
    rng = np.random.default_rng(1)
    U, I, d, K = 13187, 17366, 64, 50
    Eu = torch.tensor((rng.standard_normal((U, d)) * 0.2).astype(np.float32)).to(DEV)
    Ei = torch.tensor((rng.standard_normal((I, d)) * 0.2).astype(np.float32)).to(DEV)
    train = _train_csr(ops, U, I, rng, 12)
    q = torch.arange(U, device=DEV)
    _assert_same(*_both(ops, Eu, Ei, q, train, K), what="netflix shape")
    g = torch.Generator(device=DEV); g.manual_seed(3)
    I2, U2 = 1_000_000, 600
    Ei2 = torch.randn(I2, d, generator=g, device=DEV) * 0.3
    Eu2 = torch.randn(U2, d, generator=g, device=DEV) * 0.3
    _assert_same(*_both(ops, Eu2, Ei2, torch.arange(U2, device=DEV), None, K), what="10^6 items")
    Ei3 = torch.randn(200_000, 128, generator=g, device=DEV) * 0.3
    Eu3 = torch.randn(300, 128, generator=g, device=DEV) * 0.3
    _assert_same(*_both(ops, Eu3, Ei3, torch.arange(300, device=DEV), None, 64), what="d = 128")


@pytest.mark.parametrize("I,d", [(3000, 64), (520, 32), (20000, 64)])
def test_dense_train_rows_cross_the_staged_window(ops, I, d):
    """Train rows of 0, 1, 15, 16, 17, 33, hundreds and ALL items: the sweep stages 16 train items per (wavefront, user) in LDS and
    refills the window from inside the sweep; both modes against torch's masked top-K (values bit-exact against one another, lists
    against the fp32 reference: well-separated random scores)."""
    rng = np.random.default_rng(I + d)
    degs = np.array([0, 1, 15, 16, 17, 33, 64, 65, 200, min(1500, I - 60), I - 55, I, 2, 31, 32, 48, 100, 3] * 3)
    U, K = len(degs), 50
    rows = np.repeat(np.arange(U), degs)
    cols = np.concatenate([np.sort(rng.choice(I, size=int(dg), replace=False)) for dg in degs]).astype(np.int64)
    # a contiguous run of train items inside one item tile and across a window boundary
    run_user = 1
    rows = np.concatenate([rows, np.full(40, run_user)]); cols = np.concatenate([cols, np.arange(100, 140)])
    keep = np.unique(rows * I + cols); rows, cols = keep // I, keep % I
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, U, I)
    train = ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())
    Eu = torch.tensor(rng.standard_normal((U, d)).astype(np.float32)).to(DEV)
    Ei = torch.tensor(rng.standard_normal((I, d)).astype(np.float32)).to(DEV)
    q = torch.arange(U, device=DEV)
    i0, s0, i1, s1 = _both(ops, Eu, Ei, q, train, K)
    _assert_same(i0, s0, i1, s1, what="dense train rows")
    # the long rows went through a bitmap path (all 16 rows of a block as one tile-major slice, or - item parts, slices beyond the budget - up to
    # two per block with the others walked; 54 users = 4 user tiles, each swept by one block or, at 20 000 items, by eight blocks of an item range each)
    assert LAST["bitmap_rows"] >= 4, LAST
    S = (Eu.double() @ Ei.double().T)
    S[torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV)] = -float("inf")
    ref_s, ref_i = torch.topk(S, K, dim=1)
    for u in range(U):
        n_free = I - int((rows == u).sum())
        n = min(K, n_free)
        assert i0[u, :n].tolist() == ref_i[u, :n].tolist(), (u, int(degs[u % len(degs)]))
        assert (i0[u, n:] == -1).all(), (u, n_free)
        tr_u = set(cols[rows == u].tolist())
        assert not (set(i0[u, :n].tolist()) & tr_u)


@pytest.mark.parametrize("U,I,d,K,part", [(700, 9000, 64, 50, 1024), (333, 40000, 64, 50, 4096), (90, 20000, 128, 20, 2048), (50, 5000, 48, 50, 1024)])
def test_item_parts_of_the_bf16_sweep_equal_the_exact_sweep(ops, U, I, d, K, part):
    """Round 6: the bf16 sweep with EVERY user tile cut into item parts (part-major block ids: the plan for tables beyond the L2, forced here
    on small tables through llmrec_topk_set_part_items) - train masks that cross part boundaries, long train rows walked inside a part,
    a last part shorter than the others, exact ties that send tiles to the exact sweep: the same bits as the exact sweep."""
    rng = np.random.default_rng(U + I + part)
    Eu = torch.tensor((rng.standard_normal((U, d)) * 0.4).astype(np.float32)).to(DEV)
    Ei = torch.tensor((rng.standard_normal((I, d)) * 0.4).astype(np.float32)).to(DEV)
    train = _train_csr(ops, U, I, rng, 300)
    q = torch.tensor(rng.permutation(U)).to(DEV)
    try:
        ops.topk_set_part_items(part)
        _assert_same(*_both(ops, Eu, Ei, q, train, K), what="item parts")
        assert LAST["fallback_tiles"] == 0, LAST
        Ei_t = torch.round(Ei * 2)                                                          # integer-valued: exact ties everywhere (ordered by item id)
        Eu_t = torch.round(Eu * 2)
        _assert_same(*_both(ops, Eu_t, Ei_t, q, train, K), what="item parts, ties")
    finally:
        ops.topk_set_part_items(0)


def test_blocks_of_both_bitmap_forms_in_one_call(ops):
    """One call whose blocks choose differently: user tiles with dense train rows sweep all 16 rows as bitmaps (tile-major slice), tiles with sparse rows
    walk them - and turn their one long row into a bitmap of the two-row form INSIDE the same slice layout. (A stride mismatch between the two forms let
    the sparse tiles' slices land in the dense tiles' ones: one user of 13 187 lost an item - found by a timing script, not by the suite.)"""
    rng = np.random.default_rng(77)
    I, d, K = 3000, 64, 50
    degs = []
    for t in range(8):
        if t % 2 == 0:
            degs += list(rng.integers(20, 60, size=16))                        # dense tile: all rows as bitmaps
        else:
            row = list(rng.integers(0, 4, size=16)); row[int(rng.integers(0, 16))] = 110   # sparse tile with one long row (<= 192 items in all: walked)
            degs += row
    degs = np.array(degs + [2, 150, 1])                                        # a partial last tile
    U = len(degs)
    rows = np.repeat(np.arange(U), degs)
    cols = np.concatenate([np.sort(rng.choice(I, size=int(dg), replace=False)) for dg in degs]).astype(np.int64)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV), None, U, I)
    train = ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())
    Eu = torch.tensor((rng.standard_normal((U, d)) * 0.4).astype(np.float32)).to(DEV)
    Ei = torch.tensor((rng.standard_normal((I, d)) * 0.4).astype(np.float32)).to(DEV)
    q = torch.arange(U, device=DEV)
    for rep in range(3):                                                       # (the workspace is reused: stale slices of the call before)
        i0, s0, i1, s1 = _both(ops, Eu, Ei, q, train, K)
        _assert_same(i0, s0, i1, s1, what="mixed bitmap forms, call %d" % rep)
    assert LAST["bitmap_rows"] >= 4 * 16 + 4, LAST                             # four dense tiles + the long rows of the sparse ones
    S = Eu.double() @ Ei.double().T
    S[torch.tensor(rows).to(DEV), torch.tensor(cols).to(DEV)] = -float("inf")
    assert not torch.isinf(S.gather(1, i1.long())).any()                       # no train item in any list
