"""GPU probe of the weight-gradient launches at the Netflix shape: the multi-target launch (item_trans x5 + text + image) and
user_trans', timed with HIP events, checked against fp64. With LLMREC_LIB=<instrumented build> (python -m llmrec_amd.build --tools)
also the effective shader clock of the launch. (The LLMREC_WGRAD_KERNEL / LLMREC_WGRAD_ABL variants of round 3 are in
profiles/experiments/r03_dense_with_wgrad_variants.hip.txt.)
Usage: python tools/wgrad_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
torch.manual_seed(0)
I, U, d = 17366, 13187, 64
g = torch.Generator(device=dev); g.manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
dP = rn(I, 7 * d) * 1e-3                                      # gradient-sized values
feats = {"attr%d" % k: rn(I, 1536) for k in range(5)}
text, image, user = rn(I, 768), rn(I, 512), rn(U, 1536)
dPu = rn(U, d) * 1e-3
dW = {k: torch.empty(d, s, device=dev) for k, s in (("item", 1536), ("text", 768), ("image", 512), ("user", 1536))}
db = {k: torch.empty(d, device=dev) for k in dW}
item_pairs = [(dP[:, (2 + k) * d:(3 + k) * d], feats["attr%d" % k]) for k in range(5)]
targets = [(item_pairs, dW["item"], db["item"], False), ([(dP[:, d:2 * d], text)], dW["text"], db["text"], False),
           ([(dP[:, 0:d], image)], dW["image"], db["image"], False)]
ws = torch.empty(max(ops.linear_wgrad_multi_workspace(targets), 16), dtype=torch.uint8, device=dev)
wsu = torch.empty(ops._lib.query("llmrec_linear_wgrad_workspace_bytes", U, d, 1536), dtype=torch.uint8, device=dev)

def timeit(fn, n=iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

multi = lambda: ops.linear_wgrad_multi(targets, ws)
single = lambda: ops.linear_wgrad_grouped([(dPu, user)], dW["user"], db["user"], False, wsu, precision="bf16x3")
multi(); single(); torch.cuda.synchronize()
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
errs = {}
want = sum(dy.double().t() @ x.double() for dy, x in item_pairs); errs["item"] = rel(dW["item"], want)
errs["item_b"] = rel(db["item"], sum(dy.double().sum(0) for dy, _ in item_pairs))
errs["text"] = rel(dW["text"], dP[:, d:2 * d].double().t() @ text.double())
errs["image"] = rel(dW["image"], dP[:, 0:d].double().t() @ image.double())
errs["user"] = rel(dW["user"], dPu.double().t() @ user.double())
errs["user_b"] = rel(db["user"], dPu.double().sum(0))
a = dW["item"].clone(); multi(); torch.cuda.synchronize()
det = bool(torch.equal(a, dW["item"]))
ms_m, ms_s = timeit(multi), timeit(single)
bytes_m = 4.0 * (5 * I * 1536 + I * 768 + I * 512 + 7 * I * d + d * (1536 + 768 + 512))
bytes_s = 4.0 * (U * 1536 + U * d + d * 1536)
print("multi %.4f ms (%.0f GB/s, %.3f of 8 TB/s)  user %.4f ms (%.0f GB/s)  max rel err vs fp64 %s  deterministic %s" % (
    ms_m, bytes_m / ms_m / 1e6, bytes_m / ms_m / 1e6 / 8000, ms_s, bytes_s / ms_s / 1e6,
    {k: "%.1e" % v for k, v in errs.items()}, det))
assert max(errs.values()) < 3e-6 and det

if os.environ.get("LLMREC_LIB"):                              # instrumented build: the effective shader clock of the launch
    import ctypes
    lib = ops._lib.load()
    buf = (ctypes.c_ulonglong * 3)()
    lib.llmrec_tools_wgrad_clock(buf)
    multi(); torch.cuda.synchronize()
    lib.llmrec_tools_wgrad_clock(buf)
    v = list(buf)
    if v[2]:
        print("main loop per wave: %.0f k shader cycles, %.1f us by the 100 MHz counter -> %.2f GHz effective (%d waves)" % (
            v[0] / v[2] / 1e3, v[1] / v[2] / 100.0, v[0] / max(v[1], 1) / 10.0, v[2]))
