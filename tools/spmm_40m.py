"""The product SpMM at bench.py's spmm_roofline graph (2 M x 1 M x 40 M edges exactly, d = 64), both directions, a few launches each:
the workload tools/pmc_spmm.sh profiles (one process per PMC pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llmrec_amd import ops, synth
dev = torch.device("cuda")
U, I, E, d = 2_000_000, 1_000_000, 40_000_000, 64
rows, cols = synth.bipartite_edges_device(U, I, E, 0, dev)
g = ops.BipartiteGraph.from_edges(rows, cols, U, I)
del rows, cols
Xi, Xu = torch.randn(I, d, device=dev), torch.randn(U, d, device=dev)
Yu, Yi = torch.empty(U, d, device=dev), torch.empty(I, d, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    ops.spmm_raw(g.ui.fwd, Xi, out=Yu)      # rows = users (2 M), gathers item rows: "ui"
torch.cuda.synchronize()
for _ in range(n):
    ops.spmm_raw(g.iu.fwd, Xu, out=Yi)      # rows = items (1 M), gathers user rows: "iu"
torch.cuda.synchronize()
print("done", g.ui.fwd.nnz)
