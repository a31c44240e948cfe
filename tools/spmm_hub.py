"""Hub-row staging experiment (tools/spmm_hub.hip): the standard 2 M x 1 M x 40 M graph with items relabelled by descending degree,
Y = A_ui X (d = 64): the product kernel, the experiment kernel without staging, and with the 64 ... 512 hottest rows in LDS. The LDS copy
bounds the blocks per CU (160 KB / (256 B x hub)), so each staging size is run at the occupancy it allows and compared with the
unstaged kernel at the same number of blocks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops, synth
dev = torch.device("cuda")
U, I, E, d = 2_000_000, 1_000_000, 40_000_000, 64
rows, cols = synth.bipartite_edges_device(U, I, E, 0, dev)
deg = torch.bincount(cols, minlength=I)
order = torch.argsort(deg, descending=True); new_of_old = torch.empty_like(order); new_of_old[order] = torch.arange(I, device=dev)
cols = new_of_old[cols]
top = torch.sort(deg, descending=True).values
print("share of the edges that go to the 128 / 256 / 512 / 4096 hottest items: %s" % [round(float(top[:h].sum()) / E, 3) for h in (128, 256, 512, 4096)])
gr = ops.BipartiteGraph.from_edges(rows, cols, U, I)
a = gr.ui.fwd
X = torch.randn(I, d, device=dev); Y = torch.empty(U, d, device=dev); Y2 = torch.empty(U, d, device=dev)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "spmm_hub.so"))
lib.spmm_hub.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

ms = timeit(lambda: ops.spmm_raw(a, X, out=Y))
print("product kernel            %.4f ms  %.2f G edges/s" % (ms, a.nnz / ms / 1e6))
for hub, blocks in ((0, 512), (64, 512), (128, 512), (256, 512), (0, 256), (512, 256)):
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.spmm_hub(U, a.rowptr.data_ptr(), a.colidx.data_ptr(), a.row_scale.data_ptr(), X.data_ptr(), Y2.data_ptr(), hub, blocks, st)
    assert fn() == 0
    torch.cuda.synchronize()
    err = float((Y2 - Y).abs().max() / Y.abs().max())
    ms = timeit(fn)
    print("experiment kernel hub=%-4d blocks=%d  %.4f ms  %.2f G edges/s  (max rel diff vs the product kernel %.1e)" % (hub, blocks, ms, a.nnz / ms / 1e6, err))
