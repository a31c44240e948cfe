"""Round-5 experiment: cache-policy separation of hot and cold columns in the HBM-bound SpMM.

    python tools/spmm_nt.py [launches] [--json out.json]      # timing + bit-identity (no profiler)
    bash tools/spmm_nt.sh                                      # the same program under rocprofv3 --pmc passes -> profiles/r05_pmc_spmm_nt.json

Graph: the bench's 2 M users x 1 M items x 40 M edges (exactly), d = 64, with BOTH sides relabelled by descending degree (hot rows of
the gathered operand first; the permutation never has to leave a sharded ID step). For H in {0 (default policy everywhere), 4 K, 16 K, 64 K,
256 K}: the gathered rows with index >= H are loaded non-temporally (llmrec_spmm_epilogue_t.x_nt_from_row), the rows below H with the
default policy. Program order of the spmm_kernel dispatches: for H in Hs: `launches` x ui (rows = users, gathers item rows), then
`launches` x iu. Results must be bit-identical across H (only the load policy changes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmrec_amd import ops, synth

HS = [0, 4096, 16384, 65536, 262144]
U, I, E, d = 2_000_000, 1_000_000, 40_000_000, 64


def relabel(ids, n):
    deg = torch.bincount(ids, minlength=n)
    order = torch.argsort(deg, descending=True, stable=True)
    new_of_old = torch.empty_like(order)
    new_of_old[order] = torch.arange(n, device=ids.device)
    return new_of_old[ids], deg[order]


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    dev = torch.device("cuda")
    rows, cols = synth.bipartite_edges_device(U, I, E, 0, dev)
    rows, deg_u = relabel(rows, U)
    cols, deg_i = relabel(cols, I)
    g = ops.BipartiteGraph.from_edges(rows, cols, U, I)
    del rows, cols
    share = lambda deg, h: float(deg[:h].sum()) / float(deg.sum()) if h else 0.0
    Xi, Xu = torch.randn(I, d, device=dev), torch.randn(U, d, device=dev)
    res = {"graph": {"n_users": U, "n_items": I, "nnz": int(g.ui.fwd.nnz), "d": d, "order": "both sides relabelled by descending degree"}, "launches": launches,
           "variants": []}
    ref = {}
    for H in HS:
        rec = {"H": H}
        for name, a, X, deg in (("ui", g.ui.fwd, Xi, deg_i), ("iu", g.iu.fwd, Xu, deg_u)):
            Y = torch.empty(a.n_rows, d, device=dev)
            epi = ops.spmm_epilogue(ops.EPI_NONE, x_nt_from_row=H) if H > 0 else None
            ops.spmm_raw(a, X, out=Y, epilogue=epi)            # (one untimed launch; it is a dispatch of the PMC passes too)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(launches - 1):
                ops.spmm_raw(a, X, out=Y, epilogue=epi)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / max(launches - 1, 1)
            if H == 0:
                ref[name] = Y.clone()
            alg = 4.0 * a.nnz + 4.0 * (a.n_rows + 1) + 4.0 * a.n_rows + 4.0 * d * a.n_cols + 4.0 * d * a.n_rows
            rec[name] = {"ms": ms, "frac_hbm_algorithmic": alg / ms / 1e6 / 8000.0, "edges_per_s": a.nnz / ms * 1e3,
                         "bit_identical_to_default_policy": bool(torch.equal(Y, ref[name])),
                         "edge_share_of_rows_below_H": share(deg, H), "hot_set_mb": H * d * 4 / 1e6}
        res["variants"].append(rec)
        print(json.dumps(rec), flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
