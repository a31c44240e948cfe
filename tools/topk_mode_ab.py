"""A/B of llmrec_score_topk_mode_f32's two modes (exact-fp32 sweep vs bf16 prefilter + exact rescoring): time per call (HIP events) and
bit-identity, on random tables and on trained-shape tables (softmax-layer outputs + normalised terms: scores close together).
    python tools/topk_mode_ab.py [--json out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops


def tables(kind, U, I, d, g):
    if kind == "random":
        return torch.randn(U, d, generator=g, device="cuda") * 0.1, torch.randn(I, d, generator=g, device="cuda") * 0.1
    sm = lambda n: torch.softmax(torch.randn(n, d, generator=g, device="cuda") * 0.05, dim=1)
    nz = lambda n: torch.nn.functional.normalize(torch.randn(n, d, generator=g, device="cuda"), dim=1)
    # the fused embeddings' shape (Models.py:185-197): mean of (table, layer, softmax layer) + rates x normalised side terms
    mk = lambda n: (torch.randn(n, d, generator=g, device="cuda") * 0.1 + sm(n) + sm(n)) / 3 + 0.26 * (nz(n) + nz(n)) + 0.55 * nz(n) + 0.012 * nz(n)
    return mk(U), mk(I)


def main():
    out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    res = []
    for kind in ("random", "trained_shape"):
        shapes = ((13187, 17366, 64, 50, 20), (65536, 1_000_000, 64, 50, 2), (16384, 1_000_000, 128, 50, 2))
        if "--parts" in sys.argv:                              # round 6: the item-part plan at 10^6 items, part-size sweep (llmrec_topk_set_part_items)
            shapes = ((65536, 1_000_000, 64, 50, 2), (16384, 1_000_000, 128, 50, 2)) if kind == "random" else ()
        if "--crossover" in sys.argv:                          # where does the bf16 mode stop paying? (item table: I x d x 4 bytes)
            shapes = tuple((16384, I, 64, 50, 5) for I in (10_322, 32_768, 65_536, 131_072, 262_144, 524_288, 1_000_000)) if kind == "random" else ()
        for U, I, d, K, iters in shapes:
            Eu, Ei = tables(kind, U, I, d, g)
            q = torch.arange(U, device="cuda")
            rec = {"tables": kind, "U": U, "I": I, "d": d, "K": K}
            outs = {}
            variants = ("exact", "prefilter")
            if "--parts" in sys.argv:
                variants = ("exact", "prefilter@-1", "prefilter@8192", "prefilter@16384", "prefilter@32768", "prefilter@65536")
            for variant in variants:
                mode = variant.split("@")[0]
                ops.topk_set_part_items(int(variant.split("@")[1]) if "@" in variant else 0)
                st = {}
                out_v = ops.score_topk(Eu, Ei, q, None, K, mode=mode, stats=st)
                torch.cuda.synchronize()
                if mode == "prefilter":
                    rec["fallback_tiles"], rec["tiles"] = st["fallback_tiles"], st["tiles"]
                    if "prefilter" in outs:
                        assert torch.equal(outs["prefilter"][0], out_v[0]) and torch.equal(outs["prefilter"][1].view(torch.int32), out_v[1].view(torch.int32)), variant
                outs[mode] = out_v
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    ops.score_topk(Eu, Ei, q, None, K, mode=mode)
                e.record(); torch.cuda.synchronize()
                ms = s.elapsed_time(e) / iters
                rec[variant + "_ms"] = ms
                rec[variant + "_tflops_fp32_equivalent"] = 2.0 * U * I * d / ms / 1e9
            ops.topk_set_part_items(0)
            if "--parts" in sys.argv:
                rec["prefilter_ms"] = min(v for k, v in rec.items() if k.startswith("prefilter@") and k.endswith("_ms"))
            rec["bit_identical"] = bool(torch.equal(outs["exact"][0], outs["prefilter"][0]) and
                                        torch.equal(outs["exact"][1].view(torch.int32), outs["prefilter"][1].view(torch.int32)))
            rec["speedup"] = rec["exact_ms"] / rec["prefilter_ms"]
            res.append(rec)
            print(json.dumps(rec), flush=True)
            del Eu, Ei, outs
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
