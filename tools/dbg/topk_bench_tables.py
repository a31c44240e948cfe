"""the top-K call of the bench's evaluation alone: the Netflix-shaped workload's fused embeddings and train rows (scratch tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from llmrec_amd import ops
w = bench.NetflixShaped("nf", 0, torch.device("cuda:0"))
for _ in range(3): w.step()
w.eval_once(); torch.cuda.synchronize()
f = w.fused
Eu, Ei = f.E_u.detach().clone(), f.E_i.detach().clone()
q = w._eval_q
tr = w.graph.by_user
deg = (tr.rowptr[1:] - tr.rowptr[:-1]).float()
print("train rows: mean %.1f max %d, > 48 items: %d; |Eu| %.3f |Ei| %.3f" % (deg.mean().item(), int(deg.max().item()), int((deg > 48).sum().item()), Eu.norm(dim=1).mean().item(), Ei.norm(dim=1).mean().item()))
for t in (None, tr):
    st = {}
    ops.score_topk(Eu, Ei, q, t, 50, mode="prefilter", stats=st); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): ops.score_topk(Eu, Ei, q, t, 50, mode="prefilter")
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 50)
    print("train" if t is not None else "no train", ["%.4f" % x for x in ts], st)
# the same rows capped at 16 items (no long row): how much of the train rows' cost is the few long ones?
rp = tr.rowptr.cpu().numpy().astype("int64"); ci = tr.colidx.cpu().numpy()
import numpy as np
deg2 = np.minimum(rp[1:] - rp[:-1], 16)
rp2 = np.zeros_like(rp); rp2[1:] = np.cumsum(deg2)
ci2 = np.concatenate([ci[rp[u]:rp[u] + deg2[u]] for u in range(len(deg2))]).astype(np.int32)
cap = ops.Csr(tr.n_rows, tr.n_cols, torch.tensor(rp2, dtype=torch.int32, device="cuda"), torch.tensor(ci2, device="cuda"), None, None, None, ops.SpmmPlan())
st = {}
ops.score_topk(Eu, Ei, q, cap, 50, mode="prefilter", stats=st); torch.cuda.synchronize()
ts = []
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): ops.score_topk(Eu, Ei, q, cap, 50, mode="prefilter")
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 50)
print("train rows capped at 16", ["%.4f" % x for x in ts], st)
