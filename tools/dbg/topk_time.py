"""time of llmrec_score_topk at the Netflix shape, without and with a train CSR (~50 sorted items per user, a few long rows) (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from llmrec_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topk_mode_ab import tables
g = torch.Generator(device="cuda"); g.manual_seed(0)
U, I, d, K = 13187, 17366, 64, 50
rng = np.random.default_rng(0)
deg = np.minimum(rng.geometric(1 / 50.0, U) + 1, 2000); deg[:8] = 1500
rp = np.zeros(U + 1, np.int64); rp[1:] = np.cumsum(deg)
ci = np.concatenate([np.sort(rng.choice(I, int(k), replace=False)) for k in deg]).astype(np.int32)
train = ops.Csr(U, I, torch.tensor(rp, dtype=torch.int32, device="cuda"), torch.tensor(ci, device="cuda"), None, None, None, ops.SpmmPlan())
for kind in ("random", "trained_shape"):
    Eu, Ei = tables(kind, U, I, d, g)
    q = torch.arange(U, device="cuda")
    ref = None
    for tr in (None, train):
        st = {}
        out = ops.score_topk(Eu, Ei, q, tr, K, mode="prefilter", stats=st); torch.cuda.synchronize()
        if tr is not None and kind == "random":
            ex = ops.score_topk(Eu, Ei, q, tr, K, mode="exact")
            assert os.environ.get("NOCHECK") or torch.equal(out[0], ex[0]) and torch.equal(out[1].view(torch.int32), ex[1].view(torch.int32)), "bf16 sweep != exact sweep"
        ts = []
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50): ops.score_topk(Eu, Ei, q, tr, K, mode="prefilter")
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 50)
        print(kind, "train" if tr is not None else "no train", ["%.4f" % t for t in ts], st)
