"""where do item parts start to pay? (scratch tool) 16 384 users, d = 64, K = 50; parts off (-1) / policy (0) / 16 384-item parts"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmrec_amd import ops
g = torch.Generator(device="cuda"); g.manual_seed(0)
U, d, K = 16384, 64, 50
for I in (32768, 65536, 131072, 262144, 1000000):
    Eu = torch.randn(U, d, generator=g, device="cuda") * 0.1; Ei = torch.randn(I, d, generator=g, device="cuda") * 0.1
    q = torch.arange(U, device="cuda")
    row = {}
    for part in (-1, 0, 16384):
        ops.topk_set_part_items(part)
        ops.score_topk(Eu, Ei, q, None, K, mode="prefilter"); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ops.score_topk(Eu, Ei, q, None, K, mode="prefilter")
        e.record(); torch.cuda.synchronize()
        row[part] = round(s.elapsed_time(e) / 5, 3)
    ops.topk_set_part_items(0)
    print(I, row)
