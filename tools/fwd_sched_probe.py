"""Is the grouped projection's time set by block-round quantisation? Times the launch with subsets of the Netflix-shape problems.
    python tools/fwd_sched_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops

DEV = "cuda"
g = torch.Generator(device=DEV); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
I, U, D = 17366, 13187, 64


def job(M, K):
    return (rn(M, K), rn(D, K) / K ** 0.5, rn(D), torch.empty(M, D, device=DEV))


def timeit(jobs, n=30):
    for _ in range(3):
        ops.linear_fwd_grouped(jobs, D, precision="bf16x3")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.linear_fwd_grouped(jobs, D, precision="bf16x3")
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


attrs = [job(I, 1536) for _ in range(5)]
user = job(U, 1536)
text, image = job(I, 768), job(I, 512)
byts = lambda jobs: sum(4.0 * (x.shape[0] * x.shape[1] + D * x.shape[1] + x.shape[0] * D) for x, _, _, _ in jobs)
cases = {
    "all 8 (784 long units + 136 + 136)": attrs + [user, text, image],
    "5 attrs + text + image (680 long)": attrs + [text, image],
    "5 attrs (680 long)": attrs,
    "4 attrs + user (648 long)": attrs[:4] + [user],
    "4 attrs (544 long)": attrs[:4],
    "3 attrs + user (512 long)": attrs[:3] + [user],
    "3 attrs (408 long)": attrs[:3],
    "2 attrs (272 long)": attrs[:2],
    "user alone (104 long)": [user],
    "text + image": [text, image],
}
for name, jobs in cases.items():
    jobs = sorted(jobs, key=lambda j: -j[0].shape[1])
    ms = timeit(jobs)
    print("%-40s %.4f ms  %6.0f GB/s" % (name, ms, byts(jobs) / ms / 1e6))
