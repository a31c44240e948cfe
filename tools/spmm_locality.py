"""SpMM locality experiment: does an ingest-time row / column permutation move the HBM-bound SpMM?

    python tools/spmm_locality.py <variant> [iters]         one variant: builds the graph, times Y = A_ui X (d = 64), prints one line
    bash tools/spmm_locality.sh                              every variant under rocprofv3 --pmc passes -> a table

Graphs (2 M users x 1 M items x 40 M edges):
  std          the standard generator (power-law degrees, popularity ~ rank^-0.8, hot items spread over the id range)
  std_items    + items relabelled by descending degree (hot rows of X contiguous)
  std_both     + users ordered by their hottest neighbour (users that share hot items are processed together)
  comm         planted communities: 256 blocks of users x items, 90 % of a user's edges inside its block (ids contiguous per block)
  comm_shuf    the same graph with user and item ids shuffled (the structure is there, the order hides it)
  comm_reord   comm_shuf after the std_both reordering heuristic (what an ingest pass could recover without knowing the blocks)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops, synth

variant = sys.argv[1] if len(sys.argv) > 1 else "std"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
# round 6: LLMREC_XCD=1 -> the XCD-contiguous block -> row map (llmrec_spmm_epilogue_t.xcd_contiguous), LLMREC_DIR=iu -> Y = A_iu X
XCD = os.environ.get("LLMREC_XCD", "0") == "1"
DIR = os.environ.get("LLMREC_DIR", "ui")
dev = torch.device("cuda")
U, I, E, d = 2_000_000, 1_000_000, 40_000_000, 64


def planted(seed=0, C=256, p_in=0.9):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    u = torch.rand(U, generator=g, device=dev, dtype=torch.float64)
    raw = torch.clamp((1.0 - u) ** (-1.0 / 0.8), 1.0, 10_000.0)
    deg = torch.clamp((raw * (E / raw.sum())).floor(), min=1.0).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(U, device=dev), deg)
    n = rows.numel()
    ipc = I // C                                               # items per community
    a = 0.2
    r = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    inside = torch.rand(n, generator=g, device=dev) < p_in
    rank_in = ((r * ((ipc + 1.0) ** a - 1.0) + 1.0) ** (1.0 / a) - 1.0).floor().to(torch.int64).clamp_(0, ipc - 1)
    rank_gl = ((r * ((I + 1.0) ** a - 1.0) + 1.0) ** (1.0 / a) - 1.0).floor().to(torch.int64).clamp_(0, I - 1)
    comm = rows // (U // C)
    cols = torch.where(inside, comm.clamp_(max=C - 1) * ipc + rank_in, (rank_gl * 2654435761) % I)
    key = torch.unique(rows * I + cols)
    return key // I, key % I


def relabel_items_by_degree(rows, cols):
    deg = torch.bincount(cols, minlength=I)
    order = torch.argsort(deg, descending=True)                # new id -> old id
    new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(I, device=dev)
    return rows, new_of_old[cols]


def order_users_by_hottest_neighbour(rows, cols):
    hottest = torch.full((U,), I, dtype=torch.int64, device=dev)
    hottest.scatter_reduce_(0, rows, cols, reduce="amin")      # items are degree-ordered: the smallest id is the hottest neighbour
    order = torch.argsort(hottest, stable=True)
    new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(U, device=dev)
    return new_of_old[rows], cols


def order_users_by_degree(rows, cols):
    deg = torch.bincount(rows, minlength=U)
    order = torch.argsort(deg, descending=True, stable=True)
    new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(U, device=dev)
    return new_of_old[rows], cols


def order_users_by_bucket_then_hottest(rows, cols, two=False):
    """primary key: the degree's power-of-two bucket (descending: long rows first, equal lengths together - a wavefront of the short-row
    range takes as long as its longest row), secondary: the hottest neighbour (tertiary with two=True: the second hottest)."""
    deg = torch.bincount(rows, minlength=U)
    bucket = 63 - torch.floor(torch.log2(deg.clamp(min=1).double())).to(torch.int64)         # small = long rows
    hottest = torch.full((U,), I, dtype=torch.int64, device=dev)
    hottest.scatter_reduce_(0, rows, cols, reduce="amin")
    key = bucket * (I + 1) + hottest
    if two:
        c2 = torch.where(cols == hottest[rows], torch.full_like(cols, I), cols)
        second = torch.full((U,), I, dtype=torch.int64, device=dev)
        second.scatter_reduce_(0, rows, c2, reduce="amin")
        order = torch.argsort(second, stable=True)
        order = order[torch.argsort(key[order], stable=True)]
    else:
        order = torch.argsort(key, stable=True)
    new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(U, device=dev)
    return new_of_old[rows], cols


if variant.startswith("std"):
    rows, cols = synth.bipartite_edges_device(U, I, E, 0, dev)
else:
    rows, cols = planted()
if variant == "comm_shuf" or variant == "comm_reord":
    g = torch.Generator(device=dev); g.manual_seed(7)
    pu, pi = torch.randperm(U, generator=g, device=dev), torch.randperm(I, generator=g, device=dev)
    rows, cols = pu[rows], pi[cols]
if variant in ("std_items", "std_both", "comm_reord"):
    rows, cols = relabel_items_by_degree(rows, cols)
if variant in ("std_both", "comm_reord"):
    rows, cols = order_users_by_hottest_neighbour(rows, cols)
if variant == "std_udeg":
    rows, cols = order_users_by_degree(*relabel_items_by_degree(rows, cols))
if variant == "std_bucket_hot":
    rows, cols = order_users_by_bucket_then_hottest(*relabel_items_by_degree(rows, cols))
if variant == "std_bucket_hot2":
    rows, cols = order_users_by_bucket_then_hottest(*relabel_items_by_degree(rows, cols), two=True)
if variant == "std_hot2":
    rows, cols = relabel_items_by_degree(rows, cols)
    hottest = torch.full((U,), I, dtype=torch.int64, device=dev); hottest.scatter_reduce_(0, rows, cols, reduce="amin")
    c2 = torch.where(cols == hottest[rows], torch.full_like(cols, I), cols)
    second = torch.full((U,), I, dtype=torch.int64, device=dev); second.scatter_reduce_(0, rows, c2, reduce="amin")
    order = torch.argsort(second, stable=True); order = order[torch.argsort(hottest[order], stable=True)]
    new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(U, device=dev)
    rows = new_of_old[rows]
gr = ops.BipartiteGraph.from_edges(rows, cols, U, I)
nnz = gr.ui.fwd.nnz
del rows, cols
a = gr.ui.fwd if DIR == "ui" else gr.iu.fwd
X = torch.randn(a.n_cols, d, device=dev); Y = torch.empty(a.n_rows, d, device=dev)
epi = ops.spmm_epilogue(xcd_contiguous=True) if XCD else None
if XCD:                                                       # the other block -> row map: the same bits
    Y0 = ops.spmm_raw(a, X)
    ops.spmm_raw(a, X, out=Y, epilogue=epi)
    assert torch.equal(Y0.view(torch.int32), Y.view(torch.int32)), "xcd_contiguous changed the result"
    del Y0
for _ in range(2): ops.spmm_raw(a, X, out=Y, epilogue=epi)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): ops.spmm_raw(a, X, out=Y, epilogue=epi)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
alg = 4.0 * nnz + 8.0 * a.n_rows + 4.0 * d * (a.n_cols + a.n_rows)
print("LOCALITY %s dir %s xcd %d nnz %d ms %.4f Gedges/s %.2f frac_hbm_algorithmic %.3f gather_GBs %.0f alg_GB %.3f" % (
    variant, DIR, int(XCD), nnz, ms, nnz / ms / 1e6, alg / ms / 1e6 / 8000.0, nnz * 4.0 * d / ms / 1e6, alg / 1e9))
