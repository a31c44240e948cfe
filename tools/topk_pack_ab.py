"""A/B of the scoring + top-K sweep with the item table packed in fragment order (workspace call) against plain row loads
(llmrec_score_topk_f32, no workspace): python tools/topk_pack_ab.py [Q I d]..."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from llmrec_amd import _lib, ops   # noqa: E402

p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def run(Q, I, d, K=50, iters=5):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(1)
    Eu = torch.randn(Q, d, device=dev, generator=g); Ei = torch.randn(I, d, device=dev, generator=g)
    q = torch.arange(Q, device=dev, dtype=torch.int64)
    idx = torch.empty(Q, K, dtype=torch.int32, device=dev); sc = torch.empty(Q, K, device=dev)
    idx2 = torch.empty_like(idx); sc2 = torch.empty_like(sc)
    ws = ops.topk_workspace(Q, I, dev, d)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    packed = lambda: _lib.call("llmrec_score_topk_ws_f32", Q, p(q), p(Eu), d, p(Ei), d, I, d, None, None, K, p(idx), p(sc), p(ws), ws.numel(), st())
    plain = lambda: _lib.call("llmrec_score_topk_f32", Q, p(q), p(Eu), d, p(Ei), d, I, d, None, None, K, p(idx2), p(sc2), st())
    out = {}
    for name, fn in (("packed+split", packed), ("plain", plain), ("packed+split", packed), ("plain", plain)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        out.setdefault(name, []).append((time.perf_counter() - t0) / iters * 1e3)
    same = bool(torch.equal(idx, idx2)) and bool(torch.equal(sc, sc2))
    fl = 2.0 * Q * I * d
    print("Q %d I %d d %d: " % (Q, I, d) + "  ".join("%s %s ms (%.1f TF)" % (k, ["%.3f" % x for x in v], fl / min(v) / 1e9) for k, v in out.items()) + "  identical %s" % same)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(13187, 17366, 64), (65536, 1000000, 64), (65536, 5000000, 128)]
    for c in cases:
        run(*c)
