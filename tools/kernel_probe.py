"""GPU probe: runs the hot kernels in isolation (for rocprofv3 --pmc passes and event timing).
Usage: python tools/kernel_probe.py [fwd|wgrad|spmm|topk|all] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops, synth
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
torch.manual_seed(0)
I, U, d = 17366, 13187, 64

def timeit(fn, n=iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

if which in ("fwd", "all"):
    feats = [torch.randn(I, 512, device=dev), torch.randn(I, 768, device=dev)] + [torch.randn(I, 1536, device=dev) for _ in range(5)] + [torch.randn(U, 1536, device=dev)]
    Ws = [torch.randn(d, x.shape[1], device=dev) * 0.02 for x in feats]
    b = torch.zeros(d, device=dev)
    outs = [torch.empty(x.shape[0], d, device=dev) for x in feats]
    jobs = [(x, w, b, o) for x, w, o in zip(feats, Ws, outs)]
    ms = timeit(lambda: ops.linear_fwd_grouped(jobs, d))
    fl = sum(2.0 * x.shape[0] * x.shape[1] * d for x in feats)
    print("fwd_grouped ms %.4f TF %.1f" % (ms, fl / ms / 1e9))
    ms3 = timeit(lambda: ops.linear_fwd_grouped(jobs, d, precision="bf16x3"))
    byts = sum(4.0 * (x.shape[0] * x.shape[1] + d * x.shape[1] + x.shape[0] * d) for x in feats)
    ref = [torch.empty_like(o) for o in outs]
    ops.linear_fwd_grouped([(x, w, b, o) for x, w, o in zip(feats, Ws, ref)], d)
    ops.linear_fwd_grouped(jobs, d, precision="bf16x3")
    err = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(outs, ref))
    print("fwd_grouped_bf16x3 ms %.4f equivalent-TF %.1f HBM GB/s %.0f max rel diff vs f32 kernel %.2e" % (ms3, fl / ms3 / 1e9, byts / ms3 / 1e6, err))
    ms = timeit(lambda: ops.linear_fwd_raw(feats[2], Ws[2], b, out=outs[2]))
    print("fwd_single(I x 1536) ms %.4f TF %.1f" % (ms, 2.0 * I * 1536 * d / ms / 1e9))
if which in ("wgrad", "all"):
    X = torch.randn(I, 1536, device=dev); dY = torch.randn(I, d, device=dev)
    dW = torch.empty(d, 1536, device=dev); db = torch.empty(d, device=dev)
    ms = timeit(lambda: ops.linear_wgrad_raw(dY, X, dW, db, False))
    print("wgrad(I x 1536) ms %.4f TF %.1f" % (ms, 2.0 * I * 1536 * d / ms / 1e9))
    X5 = torch.randn(5 * I, 1536, device=dev); dY5 = torch.randn(5 * I, d, device=dev)
    ms = timeit(lambda: ops.linear_wgrad_raw(dY5, X5, dW, db, False))
    print("wgrad(5I x 1536) ms %.4f TF %.1f" % (ms, 2.0 * 5 * I * 1536 * d / ms / 1e9))
    ref = dW.clone()
    ms = timeit(lambda: ops.linear_wgrad_grouped([(dY5, X5)], dW, db, False, precision="bf16x3"))
    print("wgrad_bf16x3(5I x 1536) ms %.4f equivalent-TF %.1f HBM GB/s %.0f max rel diff vs f32 kernel %.2e" % (
        ms, 2.0 * 5 * I * 1536 * d / ms / 1e9, 4.0 * 5 * I * (1536 + d) / ms / 1e6, float((dW - ref).abs().max() / ref.abs().max())))
if which in ("spmm", "all"):
    nu, ni, ne = 2_000_000, 1_000_000, 40_000_000
    rows, cols = synth.bipartite_edges_device(nu, ni, ne, 0, dev)
    g = ops.BipartiteGraph.from_edges(rows, cols, nu, ni)
    del rows, cols
    Xi = torch.randn(ni, d, device=dev); Xu = torch.randn(nu, d, device=dev)
    Yu = torch.empty(nu, d, device=dev); Yi = torch.empty(ni, d, device=dev)
    nnz = g.ui.fwd.nnz
    for name, a, X, Y in (("ui", g.ui.fwd, Xi, Yu), ("iu", g.iu.fwd, Xu, Yi), ("ui_bwd", g.ui.bwd, Xu, Yi), ("iu_bwd", g.iu.bwd, Xi, Yu)):
        ms = timeit(lambda: ops.spmm_raw(a, X, out=Y))
        alg = 4.0 * nnz + 8.0 * a.n_rows + 4.0 * d * (a.n_cols + a.n_rows)
        print("spmm %s nnz %d ms %.4f Gedges/s %.2f alg GB/s %.1f gather GB/s %.1f" % (name, nnz, ms, nnz / ms / 1e6, alg / ms / 1e6, (nnz * (4 + 4.0 * d) + 4.0 * d * a.n_rows) / ms / 1e6))
if which in ("topk", "all"):
    Eu = torch.randn(U, d, device=dev); Ei = torch.randn(I, d, device=dev)
    rows, cols = synth.bipartite_edges(U, I, 55146, seed=0)
    rp, ci, _ = ops.csr_from_coo(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev), None, U, I)
    tr = ops.Csr(U, I, rp, ci, None, None, None, ops.SpmmPlan())
    q = torch.arange(U, device=dev)
    for mode in ("exact", "prefilter"):                      # both sweeps (bit-identical lists): profiles/experiments/r05_topk.md
        ms = timeit(lambda: ops.score_topk(Eu, Ei, q, tr, 50, mode=mode))
        print("score_topk[%s] ms %.4f TF %.2f users/s %.0f" % (mode, ms, 2.0 * U * I * d / ms / 1e9, U / ms * 1e3))
