"""time of llmrec_score_topk at the Netflix shape with a train CSR (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmrec_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topk_mode_ab import tables
g = torch.Generator(device="cuda"); g.manual_seed(0)
for kind in ("random", "trained_shape"):
    U, I, d, K = 13187, 17366, 64, 50
    Eu, Ei = tables(kind, U, I, d, g)
    q = torch.arange(U, device="cuda")
    for mode in ("prefilter",):
        st = {}
        ops.score_topk(Eu, Ei, q, None, K, mode=mode, stats=st); torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50): ops.score_topk(Eu, Ei, q, None, K, mode=mode)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 50)
        print(kind, mode, ["%.4f" % t for t in ts], st)
