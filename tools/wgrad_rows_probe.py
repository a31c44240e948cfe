"""A/B of the row-listed weight-gradient launch at the Netflix shape: time vs list length (and vs the geometry hint), against the dense launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llmrec_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
U, d = 13187, 64
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
Xs = [rn(U, 1536) for _ in range(5)]
Xu, Xt, Xi = rn(U, 1536), rn(U, 768), rn(U, 512)
roww = torch.rand(U, generator=g, device=dev)
Wg = [torch.empty(d, 1536, device=dev), torch.empty(d, 1536, device=dev), torch.empty(d, 768, device=dev), torch.empty(d, 512, device=dev)]
bg = [torch.empty(d, device=dev) for _ in range(4)]
dYu, dYt, dYi = rn(U, d) * 1e-5, rn(U, d) * 1e-5, rn(U, d) * 1e-5

def run(n_act, expected, listed=True, iters=30):
    ids = torch.sort(torch.randperm(U, generator=g, device=dev)[:n_act]).values.to(torch.int32)
    lst = torch.zeros(U + 32, dtype=torch.int32, device=dev); lst[:n_act] = ids
    n = torch.tensor([n_act], dtype=torch.int32, device=dev)
    dY = torch.zeros(U, 7 * d, device=dev)
    dY[ids.long()] = rn(n_act, 7 * d) * 1e-5
    rows = (lst, n, expected) if listed else None
    item = [(dY[:, (2 + k) * d:(3 + k) * d], Xs[k], roww, rows) for k in range(5)]
    targets = [(item, Wg[0], bg[0], False), ([(dYu, Xu)], Wg[1], bg[1], False), ([(dYt, Xt, roww)], Wg[2], bg[2], False), ([(dYi, Xi, roww)], Wg[3], bg[3], False)]
    ws = torch.empty(max(ops.linear_wgrad_multi_workspace(targets), 16), dtype=torch.uint8, device=dev)
    for _ in range(5):
        ops.linear_wgrad_multi(targets, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear_wgrad_multi(targets, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print("dense (no list)            : %.1f us" % run(U, 0, listed=False))
for n_act, exp in ((U, 0), (U, U), (5700, 6300), (5700, 0), (5700, 3000), (2000, 2200), (64, 128), (0, 64)):
    print("listed n=%5d expected=%5d : %.1f us" % (n_act, exp, run(n_act, exp)))
# only the item target (no dense targets beside it)
def run_item_only(n_act, expected, listed=True, iters=30):
    ids = torch.sort(torch.randperm(U, generator=g, device=dev)[:n_act]).values.to(torch.int32)
    lst = torch.zeros(U + 32, dtype=torch.int32, device=dev); lst[:n_act] = ids
    n = torch.tensor([n_act], dtype=torch.int32, device=dev)
    dY = torch.zeros(U, 7 * d, device=dev); dY[ids.long()] = rn(n_act, 7 * d) * 1e-5
    rows = (lst, n, expected) if listed else None
    item = [(dY[:, (2 + k) * d:(3 + k) * d], Xs[k], roww, rows) for k in range(5)]
    targets = [(item, Wg[0], bg[0], False)]
    ws = torch.empty(max(ops.linear_wgrad_multi_workspace(targets), 16), dtype=torch.uint8, device=dev)
    for _ in range(5): ops.linear_wgrad_multi(targets, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.linear_wgrad_multi(targets, ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("item target only, dense    : %.1f us" % run_item_only(U, 0, listed=False))
for n_act, exp in ((U, U), (5700, 6300), (2000, 2200)):
    print("item only listed n=%5d   : %.1f us" % (n_act, run_item_only(n_act, exp)))
