"""Ceilings of the SpMM access pattern (tools/gatherbench.hip): rate of indexed row gathers by table size
(L2 / Infinity Cache / HBM), index distribution (uniform, the synthetic graphs' rank^-0.8 popularity), row width,
loads in flight, cache policy and with / without the streaming output row.

    python tools/gatherbench.py [out.json]
"""
import ctypes
import json
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "gatherbench.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "gatherbench.hip")):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "gatherbench.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def zipf_idx(n_rows, n, alpha=0.8, spread=True, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    r = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    a = 1.0 - alpha
    rank = ((r * ((n_rows + 1.0) ** a - 1.0) + 1.0) ** (1.0 / a) - 1.0).floor().to(torch.int64).clamp_(0, n_rows - 1)
    if spread:
        import math
        mult = 2654435761 % n_rows
        while math.gcd(mult, n_rows) != 1:
            mult += 1
        rank = (rank * mult) % n_rows
    return rank.to(torch.int32)


def uniform_idx(n_rows, n, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(0, n_rows, (n,), generator=g, device=dev, dtype=torch.int32)


results = []


def run(name, rowf, n_rows, idx, deg, write, unroll=8, xpol=0, ynt=0, blocks=None):
    n_out = idx.numel() // deg
    X = torch.empty(n_rows * rowf, device=dev).normal_()
    Y = torch.empty(max(n_out, 1) * rowf if write else 64, device=dev)
    lpr = rowf // 4
    full = (n_out * lpr + 255) // 256
    b = full if blocks is None else min(blocks, full)
    rc = [0]

    def f():
        rc[0] = lib.run_gather(rowf, unroll, xpol, ynt, p(X), p(idx), ctypes.c_int64(n_out), deg, p(Y), write, b, st())
    ms = timeit(f)
    assert rc[0] == 0, (name, rc[0])
    n_g = n_out * deg
    r = {"name": name, "rowf": rowf, "table_mb": n_rows * rowf * 4 / 2**20, "gathers": n_g, "deg": deg, "write": write, "unroll": unroll,
         "xpol": xpol, "ynt": ynt, "blocks": b, "ms": ms, "ggathers_per_s": n_g / ms / 1e6, "gather_gbs": n_g * rowf * 4 / ms / 1e6,
         "alg_gbs": (n_g * 4 + n_rows * rowf * 4 + (n_out * rowf * 4 if write else 0)) / ms / 1e6}
    results.append(r)
    print("%-58s %8.3f ms  %6.2f G gathers/s  gather %7.0f GB/s  algorithmic %6.0f GB/s" % (name, ms, r["ggathers_per_s"], r["gather_gbs"], r["alg_gbs"]), flush=True)
    del X, Y


N = 36_000_000 // 18 * 18
# A: uniform indices, table size sweep (256-B rows, no output stream): L2 -> Infinity Cache -> HBM plateaus
for n_rows in (8192, 65536, 262144, 524288, 1_000_000, 2_000_000, 8_000_000):
    run("A uniform rows=%d (%.0f MB) d=64" % (n_rows, n_rows * 256 / 2**20), 64, n_rows, uniform_idx(n_rows, N), 18, 0)
# B: the synthetic graphs' popularity (rank^-0.8, ids spread), 1 M rows = 256 MB, with the output stream of an 18-nnz row
zi = zipf_idx(1_000_000, N)
run("B zipf0.8 1M rows d=64 no-write", 64, 1_000_000, zi, 18, 0)
run("B zipf0.8 1M rows d=64 write", 64, 1_000_000, zi, 18, 1)
run("B zipf0.8 1M rows d=64 write nt-store", 64, 1_000_000, zi, 18, 1, ynt=1)
run("B zipf0.8 1M rows d=64 write nt-load", 64, 1_000_000, zi, 18, 1, xpol=1)
run("B zipf0.8 1M rows d=64 write nt-load nt-store", 64, 1_000_000, zi, 18, 1, xpol=1, ynt=1)
run("B zipf0.8 1M rows d=64 write unroll4", 64, 1_000_000, zi, 18, 1, unroll=4)
run("B zipf0.8 1M rows d=64 write unroll16", 64, 1_000_000, zi, 18, 1, unroll=16)
run("B zipf0.8 1M rows d=64 write unroll16 nt-store", 64, 1_000_000, zi, 18, 1, unroll=16, ynt=1)
for blocks in (256, 512, 1024, 2048, 4096):
    run("B zipf0.8 1M rows d=64 write persistent blocks=%d" % blocks, 64, 1_000_000, zi, 18, 1, blocks=blocks)
run("B zipf0.8 1M rows d=64 write ids=rank (hot rows adjacent)", 64, 1_000_000, zipf_idx(1_000_000, N, spread=False), 18, 1)
# C: d-slices of the same table (the column block that fits the Infinity Cache / more rows per L2 byte)
run("C zipf0.8 1M rows d=32 slice (128 MB) write", 32, 1_000_000, zi, 18, 1)
run("C zipf0.8 1M rows d=32 slice (128 MB) write unroll16", 32, 1_000_000, zi, 18, 1, unroll=16)
run("C zipf0.8 1M rows d=16 slice (64 MB) write", 16, 1_000_000, zi, 18, 1)
run("C zipf0.8 1M rows d=16 slice (64 MB) write unroll16", 16, 1_000_000, zi, 18, 1, unroll=16)
run("C zipf0.8 1M rows d=128 (512 MB) write", 128, 1_000_000, zi, 18, 1)
# D: the iu direction of cfg 4: gathers from a 2 M / 10 M row table by a flat-ish (degree-proportional) distribution
run("D uniform 2M rows d=64 deg36 write", 64, 2_000_000, uniform_idx(2_000_000, N), 36, 1)
run("D uniform 10M rows d=64 deg36 write", 64, 10_000_000, uniform_idx(10_000_000, N), 36, 1)
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=1)
# E: rows per lane group matter? the same 36 M gathers cut into rows of deg nnz (each row also writes its 256-B output)
for deg in (1, 2, 4, 8, 18, 64):
    n = N // deg * deg
    run("E zipf0.8 1M rows d=64 write deg=%d" % deg, 64, 1_000_000, zi[:n], deg, 1)
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=1)
