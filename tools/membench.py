import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "membench.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "membench.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda")
M, K = 17366 * 5, 1536
X = torch.randn(M * K + 4096, device=dev)
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
byts = 4.0 * (M // 128) * 128 * K
blocks = M // 128
ms = timeit(lambda: lib.run_stream(p(X), ctypes.c_int64(M * K // 4), 2048, p(out), st))
print("coalesced stream            : %.3f ms  %.0f GB/s" % (ms, 4.0 * M * K / ms / 1e6))
ms = timeit(lambda: lib.run_frag(p(X), ctypes.c_int64(K), ctypes.c_int64(128 * K), K, blocks, p(out), st))
print("fragment, row-major X (6 KB row stride): %.3f ms  %.0f GB/s" % (ms, byts / ms / 1e6))
# blocked layout: tile = [128 rows][64 k] contiguous (row stride 64 floats); a block walks its K/64 tiles
for kt in (64,):
    tiles_per_block = K // kt
    # emulate: row stride = kt floats, consecutive k-tiles of a row block are 128*kt floats apart -> use kfloats=kt per launch "tile", blocks = all tiles
    ms = timeit(lambda: lib.run_frag(p(X), ctypes.c_int64(kt), ctypes.c_int64(128 * kt), kt, blocks * tiles_per_block, p(out), st))
    print("fragment, blocked tiles [128][%d] (one block per tile): %.3f ms  %.0f GB/s" % (kt, ms, byts / ms / 1e6))

for contig in (0, 1):
    for barrier in (0, 1):
        ms = timeit(lambda: lib.run_gemmx(p(X), ctypes.c_int64(K), K, blocks, contig, barrier, p(out), st))
        print("projection X pattern (32 k per step, 3 stages) contiguous-64B=%d barrier=%d: %.3f ms  %.0f GB/s" % (contig, barrier, ms, byts / ms / 1e6))

# the weight-gradient kernel's X stream alone (one wave per SIMD, 240 blocks): how the rate depends on the tiles in flight
lib.run_wgradx.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
MC = 2176                                            # 5 x 17366 rows in 40 slabs -> 10 groups x 24 k slabs = 240 blocks
for stages in (2, 3, 4, 6):
    ms = timeit(lambda: lib.run_wgradx(p(X), K, M, K, MC, stages, p(out), st))
    print("wgrad X stream alone, %d tiles (of 8 KB per wave) rotating, one wave per SIMD: %.3f ms  %.0f GB/s" % (stages, ms, 4.0 * M * K / ms / 1e6))
