"""bench.py - the driver's measurement contract for the LLMRec Stage-2 hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one mini-batch: on-device BPR sampling + LLM-augmented
triples, the full-graph forward (projections, 20 SpMM, softmax, fusion), the 8 fused BPR+prune
losses, backward and AdamW - the loop body of the reference's Trainer.train (main.py:210-283).

N = 1 workload (BASELINE.json configs[1]): Netflix-SHAPED synthetic data (the real files are not
distributable): U=13187, I=17366, 55146 train edges, d=64, 2 propagation layers, image/text/LLM
side features (512/768/1536-d, 5 attribute keys), batch 1024 + 10 % augmented triples, prune 0.71.
metric value = BPR train edges/s = steps * batch_size / time (the reference's own timer
definition, main.py:200,297); the full-rank eval rate (users/s, main.py:297-303) is reported in
"eval". Inputs are resident in HBM before the timed region.

N > 1 (one process per GPU, RCCL): the SAME workload with the global batch N x 1024 sharded over
batch-sharded replicas (llmrec_amd/dp.py): graph and tables replicated (they are < 1 GB), prune
threshold and regulariser norms over the GLOBAL batch (one 36 KB all-gather), one all-reduce of
the flat gradient bucket (8.9 MB) per step - weak scaling in the batch. Evaluation shards the
users. The user-ROW-sharded path for graphs that need it (cfg 4/5, SURVEY.md 8(e): per-layer
all-reduce of the item messages, llmrec_amd/dist.py) is --workload synth.
One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", help="auto | nf (cfg 2) | ml (cfg 3) | synth (user-sharded ID path, cfg 4 shape)")
    ap.add_argument("--synth-users", type=int, default=1_250_000, help="users PER GPU for --workload synth")
    ap.add_argument("--synth-items", type=int, default=1_000_000)
    ap.add_argument("--synth-edges", type=int, default=25_000_000, help="edges PER GPU for --workload synth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def import_dropin_models(dataset: str, extra=()):
    """Models.py parses sys.argv at import (as the reference does); give it the workload's flags."""
    old = sys.argv
    sys.argv = ["main.py", "--dataset", dataset, "--debug"] + list(extra)
    try:
        for name in ("Models", "utility.parser"):
            sys.modules.pop(name, None)
        import Models
        return Models
    finally:
        sys.argv = old


def event_time_ms(fn, iters: int, warmup: int = 3):
    """Average duration of fn() in ms, HIP events on torch's current stream (the stream the C ABI
    launches on)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters


class NetflixShaped:
    """cfg 2 (and cfg 3 with shape='ml'): the full multi-modal step on one GPU."""

    def __init__(self, shape: str, seed: int, device, rank: int = 0, world: int = 1, comm=None):
        import numpy as np
        self.rank, self.world = rank, world
        import torch
        from llmrec_amd import ops, engine, synth
        self.shape_name = shape
        sh = synth.NF_SHAPE if shape == "nf" else synth.ML_SHAPE
        dataset = "netflix_valid_item" if shape == "nf" else "preprocessed_raw_MovieLens"
        extra = [] if shape == "nf" else ["--weight_size", "[64,64,64]"]
        self.Models = import_dropin_models(dataset, extra)
        self.args = self.Models.args
        self.keys = synth.DATASET_KEYS[dataset]
        self.sh = sh
        rows, cols = synth.bipartite_edges(sh.n_users, sh.n_items, sh.n_train, seed=seed)
        self.rows, self.cols = rows, cols
        g = torch.Generator(device=device); g.manual_seed(seed + 1)
        rn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=torch.float32)
        self.feats = {"image": rn(sh.n_items, sh.image_dim), "text": rn(sh.n_items, sh.text_dim),
                      "user": rn(sh.n_users, sh.llm_dim)}
        for k in self.keys:
            self.feats["attr/" + k] = rn(sh.n_items, sh.llm_dim)
        torch.manual_seed(seed)
        self.graph = ops.BipartiteGraph.from_edges(torch.from_numpy(rows).to(device), torch.from_numpy(cols).to(device),
                                                   sh.n_users, sh.n_items)
        weight_size = eval(self.args.weight_size)
        self.model = self.Models.MM_Model(sh.n_users, sh.n_items, self.args.embed_size, weight_size, [0.1] * len(weight_size),
                                          self.feats["image"], self.feats["text"], self.feats["user"],
                                          {k: self.feats["attr/" + k] for k in self.keys}).to(device)
        self.opt = ops.FusedAdamW(self.model.parameters(), lr=self.args.lr)
        self.hp = engine.Hyper.from_args(self.args)
        rng = np.random.default_rng(seed + 2)
        hi = int(sh.n_items * 1.05) + 1                      # ~5 % of the LLM pairs point past n_items and are filtered
        self.aug_pos = torch.from_numpy(rng.integers(0, hi, size=sh.n_users)).to(device)
        self.aug_neg = torch.from_numpy(rng.integers(0, hi, size=sh.n_users)).to(device)
        exist = torch.unique(torch.from_numpy(rows)).to(device)
        self.batcher = engine.DeviceBatcher(self.graph.by_user, exist, sh.n_items, self.hp.batch_size,
                                            self.aug_pos, self.aug_neg, self.hp.aug_sample_rate, seed, rank=rank, world=world)
        self.engine, self.ops, self.device = engine, ops, device
        self.step_id = 0
        self.units_per_step = self.hp.batch_size
        # the fused step (hand-written backward, 7-stream SpMM operands, multi-BPR) replayed from a HIP graph
        from llmrec_amd.fused import FusedStep
        a = self.args
        rates, cap = (a.model_cat_rate, a.user_cat_rate, a.item_cat_rate), self.hp.batch_size + self.batcher.n_aug
        if world > 1 or os.environ.get("LLMREC_FORCE_DP", "0") == "1":   # batch-sharded replicas: global prune + gradient all-reduce
            from llmrec_amd.dp import DataParallelStep
            self.fused = DataParallelStep(self.model, self.graph, self.hp, rates, self.opt, cap, comm=comm)
        else:
            self.fused = FusedStep(self.model, self.graph, self.hp, rates, self.opt, cap)
        self.use_graph = os.environ.get("LLMREC_GRAPH", "1") == "1"

    def step(self):
        """One training step. With the HIP graph: sampler + forward + losses + backward + AdamW are ONE graph
        replay (three between the two exchanges on batch-sharded replicas); nothing else is enqueued."""
        self.step_id += 1
        if self.use_graph:
            if self.fused.graph_exec is None:
                self.fused.capture(batcher=self.batcher)       # the capture's warm-up is a real step
                return self.fused.scal[1:4]
            return self.fused.step()
        u, p, n, nv = self.batcher.next()
        return self.fused.step_eager(u, p, n, nv)

    def step_modular(self):
        """The same step through torch.autograd over the per-op Functions (reference-shaped path)."""
        u, p, n, nv = self.batcher.next()
        self.step_id += 1
        return self.engine.train_step(self.model, self.opt, self.graph.ui, self.graph.iu, u, p, n, self.hp, n_valid=nv)

    def eval_once(self):
        """Full-rank evaluation of this rank's user block (users shard, items are replicated: SURVEY.md 8(e))."""
        import torch
        per = (self.sh.n_users + self.world - 1) // self.world
        q = torch.arange(min(self.rank * per, self.sh.n_users), min((self.rank + 1) * per, self.sh.n_users), dtype=torch.int64,
                         device=self.device)
        if not hasattr(self, "_eval_q") or self._eval_q.numel() != q.numel():
            self._eval_q = q                                    # fixed query set: the evaluation graph is captured once
        with torch.no_grad():                                   # eval-mode forward (no dropout in this config) + scoring + top-50
            return self.fused.eval_topk(self._eval_q, self.graph.by_user, 50, use_graph=self.use_graph)

    def config(self):
        return {"workload": "netflix_shaped_cfg2" if self.shape_name == "nf" else "movielens_shaped_cfg3",
                "n_users": self.sh.n_users, "n_items": self.sh.n_items, "n_train_edges": int(self.rows.size),
                "embed_size": self.args.embed_size, "prop_layers": len(eval(self.args.weight_size)),
                "batch_size": self.hp.batch_size, "aug_sample_rate": self.hp.aug_sample_rate,
                "prune_loss_drop_rate": self.hp.prune_loss_drop_rate, "side_features": "image512+text768+llm1536x(1+5)",
                "sampler": "device (llmrec_sample_bpr)", "global_batch": self.hp.batch_size * self.world,
                "parallelism": "single GPU" if not hasattr(self.fused, "gsz") else
                ("dp%d: batch-sharded replicas (llmrec_amd/dp.py), graph + tables replicated, prune over the global batch "
                 "(1 all-gather of %d B) + 1 all-reduce of the %d B gradient bucket per step; eval shards the users"
                 % (self.world, 4 * self.fused.gsz, 4 * self.fused.bucket.numel())),
                "step": ("fused (llmrec_amd/fused.py)" if not hasattr(self.fused, "gsz") else "fused, 3 segments between the 2 exchanges (llmrec_amd/dp.py)")
                        + (" + HIP graph replay" if self.use_graph else "")}

    # ---- per-kernel roofline (dominant kernels of this workload, timed in isolation) -------------
    def kernel_rooflines(self):
        import torch
        ops, sh, d = self.ops, self.sh, self.args.embed_size
        out = []
        W = self.model.item_trans.weight.detach(); b = self.model.item_trans.bias.detach()
        X = self.feats["attr/" + self.keys[0]]
        feats = [self.feats["image"], self.feats["text"], self.feats["user"]] + [self.feats["attr/" + k] for k in self.keys]
        flop_all = sum(2.0 * x.shape[0] * x.shape[1] * d for x in feats)
        byts_all = sum(4.0 * (x.shape[0] * x.shape[1] + d * x.shape[1] + x.shape[0] * d) for x in feats)
        ms = event_time_ms(self.fused._project_all, 20)
        bf = self.fused.gemm == "bf16x3"
        out.append({"kernel": ("linear_fwd_grouped_bf16x3_kernel (all 8 projections, one launch; 3-term bf16 split, 6 bf16 MFMAs: "
                               "HBM-bound on the X stream - tflops/frac_mfma_f32 are fp32-EQUIVALENT figures)") if bf else
                              "linear_fwd_grouped_kernel (all 8 projections of one forward, one launch, exact fp32 MFMA)",
                    "pmc": [("linear_fwd_grouped_bf16x3_kernel" if bf else "linear_fwd_grouped_kernel", 1)],
                    "bound": "hbm" if bf else "mfma", "calls_per_step": 1, "ms": ms,
                    "tflops": flop_all / ms / 1e9, "frac_mfma_f32": flop_all / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                    "gbs": byts_all / ms / 1e6, "frac_hbm": byts_all / ms / 1e6 / HBM_PEAK_GBS,
                    "algorithmic_flop_per_launch": flop_all, "algorithmic_bytes_per_launch": byts_all})
        flop = 2.0 * sh.n_items * sh.llm_dim * d
        byts = 4.0 * (sh.n_items * sh.llm_dim + d * sh.llm_dim + sh.n_items * d)
        # the step's four weight-gradient launches: item_trans (5 attribute streams grouped), user, text, image
        dYi = torch.randn(sh.n_items, 7 * d, device=self.device); dYu = torch.randn(sh.n_users, d, device=self.device)
        m_ = self.model
        ws = self.fused.ws_wgrad

        def wgrad_all():
            pr = self.fused.gemm
            ops.linear_wgrad_grouped([(dYi[:, (2 + k) * d:(3 + k) * d], m_.item_feats[key]) for k, key in enumerate(self.keys)],
                                     m_.item_trans.weight.grad, m_.item_trans.bias.grad, False, ws, precision=pr)
            ops.linear_wgrad_grouped([(dYu, m_.user_feats)], m_.user_trans.weight.grad, m_.user_trans.bias.grad, False, ws, precision=pr)
            ops.linear_wgrad_grouped([(dYi[:, d:2 * d], m_.text_feats)], m_.text_trans.weight.grad, m_.text_trans.bias.grad, False, ws, precision=pr)
            ops.linear_wgrad_grouped([(dYi[:, 0:d], m_.image_feats)], m_.image_trans.weight.grad, m_.image_trans.bias.grad, False, ws, precision=pr)
        ms = event_time_ms(wgrad_all, 20)
        out.append({"kernel": ("linear_wgrad_bf16x3_kernel" if bf else "linear_wgrad_kernel<true>") +
                              " + reduce_chunks_kernel (the step's 4 launches: item_trans x5 grouped, user, text, image" +
                              ("; 3-term bf16 split: HBM-bound on the X stream, tflops are fp32-EQUIVALENT)" if bf else ")"),
                    "pmc": [("linear_wgrad_bf16x3_kernel" if bf else "linear_wgrad_kernel", 4), ("reduce_chunks_kernel", 8)],
                    "launches": 4, "avg_launch_ms": ms / 4,
                    "bound": "hbm" if bf else "mfma", "calls_per_step": 1, "ms": ms, "tflops": flop_all / ms / 1e9, "frac_mfma_f32": flop_all / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                    "gbs": byts_all / ms / 1e6, "frac_hbm": byts_all / ms / 1e6 / HBM_PEAK_GBS,
                    "algorithmic_flop_per_launch": flop_all, "algorithmic_bytes_per_launch": byts_all})
        Xi = torch.randn(sh.n_items, d, device=self.device)
        a = self.graph.ui.fwd
        ms = event_time_ms(lambda: ops.spmm_raw(a, Xi), 50)
        byts = 4.0 * a.nnz + 4.0 * (a.n_rows + 1) + 4.0 * a.n_rows + 4.0 * d * a.n_cols + 4.0 * d * a.n_rows
        out.append({"kernel": "spmm_rows_segments_kernel<16,1,4> + spmm_finalize_kernel (ui, d = 64, NF scale: L2-resident, launch-bound)", "calls_per_step": 20,
                    "ms": ms, "gbs": byts / ms / 1e6, "frac_hbm": byts / ms / 1e6 / HBM_PEAK_GBS, "edges_per_s": a.nnz / ms * 1e3})
        return out


def pmc_traffic_bytes(parts):
    """HBM-side bytes of one "launch" as the roofline defines it, from the committed PMC pass
    (profiles/r01_pmc_bench_step.json: separate rocprofv3 --pmc runs of this bench with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled per its gfx950 note, WRITE_SIZE taken as reported;
    both in KB). parts: [(kernel-name substring, launches of it per roofline launch)]. None when a kernel
    is missing from the pass."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_bench_step.json")
    if not os.path.exists(path):
        return None
    try:
        data = json.load(open(path))
    except Exception:
        return None
    total = 0.0
    for key, launches in parts:
        hit = [c for name, c in data.items() if key in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
        if not hit:
            return None
        total += launches * (2.0 * hit[0]["FETCH_SIZE"]["mean"] + hit[0]["WRITE_SIZE"]["mean"]) * 1024.0
    return total


def spmm_roofline_large(device, seed, n_users=2_000_000, n_items=1_000_000, n_edges=40_000_000, d=64):
    """The HBM-bound regime of the SpMM (north_star's roofline target): cfg-4-shaped synthetic
    graph at single-GPU size. Algorithmic bytes per SURVEY.md 8(d):
    4 nnz + 4 (rows + 1) + 4 rows + 4 d cols + 4 d rows."""
    import torch
    from llmrec_amd import ops, synth
    rows, cols = synth.bipartite_edges_device(n_users, n_items, n_edges, seed, device)
    g = ops.BipartiteGraph.from_edges(rows, cols, n_users, n_items)
    nnz = g.ui.fwd.nnz
    del rows, cols
    res = {}
    Xi = torch.randn(n_items, d, device=device); Xu = torch.randn(n_users, d, device=device)
    for name, a, X in (("ui", g.ui.fwd, Xi), ("iu", g.iu.fwd, Xu)):
        Y = torch.empty(a.n_rows, d, device=device)
        ms = event_time_ms(lambda: ops.spmm_raw(a, X, out=Y), 10, warmup=2)
        alg = 4.0 * nnz + 4.0 * (a.n_rows + 1) + 4.0 * a.n_rows + 4.0 * d * a.n_cols + 4.0 * d * a.n_rows
        gather = nnz * (4.0 + 4.0 * d) + 4.0 * d * a.n_rows
        res[name] = {"ms": ms, "edges_per_s": nnz / ms * 1e3, "algorithmic_gbs": alg / ms / 1e6,
                     "frac_hbm_algorithmic": alg / ms / 1e6 / HBM_PEAK_GBS, "no_reuse_gather_gbs": gather / ms / 1e6,
                     "n_long_rows": a.plan.n_long}
    return {"graph": {"n_users": n_users, "n_items": n_items, "nnz": int(nnz), "d": d}, **res}


def cpu_baseline_nf(w: NetflixShaped, budget_s: float = 20.0):
    """The oracle (CPU restatement of the reference, oracle/oracle.py) timed on this host on a
    bounded number of steps of the SAME workload. kind = "port"."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    from oracle import oracle as O
    sh = w.sh
    cfg = O.Config.from_args(vars(w.args), w.keys)
    R = sp.csr_matrix((np.ones(w.rows.size, dtype=np.float32), (w.rows, w.cols)), shape=(sh.n_users, sh.n_items))
    a_ui, a_iu = O.normalized_graphs(R)
    feats = {k: v.cpu() for k, v in w.feats.items()}
    names = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_trans.weight",
             "user_trans.bias", "item_trans.weight", "item_trans.bias", "user_id_embedding.weight", "item_id_embedding.weight"]
    sd = w.model.state_dict()
    params = {k: sd[k].detach().cpu().clone().requires_grad_(True) for k in names}
    opt = O.AdamW(params, lr=cfg.lr)
    rng = np.random.default_rng(0)
    B = cfg.batch_size + int(cfg.batch_size * cfg.aug_sample_rate)
    steps, t0 = 0, time.perf_counter()
    while True:
        users = rng.integers(0, sh.n_users, size=B); pos = rng.integers(0, sh.n_items, size=B); neg = rng.integers(0, sh.n_items, size=B)
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
        loss, _ = O.step_loss(fw, users, pos, neg, sh.n_items, cfg)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        opt.step(grads)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 64:
            break
    dt = time.perf_counter() - t0
    # eval sample: 256 users, full ranking (the reference's per-user python ranking, batch_test.py:83-109)
    with torch.no_grad():
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
    train_items = {u: w.cols[w.rows == u].tolist() for u in range(256)}
    test_set = {u: [int(rng.integers(0, sh.n_items))] for u in range(256)}
    t1 = time.perf_counter()
    O.evaluate(fw["E_u"].numpy(), fw["E_i"].numpy(), list(range(256)), train_items, test_set, cfg.Ks, batch_size=cfg.batch_size)
    de = time.perf_counter() - t1
    return {"value": steps * cfg.batch_size / dt, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full training steps of the same workload (oracle/oracle.py, torch-CPU fp32); eval on 256 users" % steps,
            "ms_per_step": dt / steps * 1e3, "eval_users_per_s": 256 / de}


def main():
    a = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("LLMREC_BENCH_SINGLE_DEVICE", "0") == "1":   # test hook: all ranks on cuda:0 (1-GPU box, with LLMREC_DIST_BACKEND=gloo)
        local_dev = 0
    else:
        local_dev = local
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if local == 0:                                           # one builder per node; the .so normally travels prebuilt
        from llmrec_amd import build as _build
        _build.build(force=False, verbose=False)
    use_pg = world > 1 or ("RANK" in os.environ and os.environ.get("LLMREC_DP_FORCE_COLLECTIVES", "0") == "1")
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("LLMREC_DIST_BACKEND", "nccl")             # "nccl" IS RCCL on ROCm; gloo only for the 1-GPU smoke of the N > 1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        dist.barrier()                                       # the library exists before any rank loads it
    workload = a.workload
    if workload == "auto":
        workload = "nf"

    if workload in ("nf", "ml"):
        from llmrec_amd import dist as ldist
        w = NetflixShaped(workload, a.seed, device, rank, world, ldist.Comm() if use_pg else None)
        step, units = w.step, w.units_per_step * world      # global batch = world x batch_size
    else:
        from llmrec_amd import dist as ldist
        w = ldist.ShardedBench(a.synth_users, a.synth_items, a.synth_edges, a.seed, device, rank, world)
        step, units = w.step, w.units_per_step              # global batch per step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    finish = getattr(getattr(w, "fused", None), "flush", lambda: None)   # batch-sharded replicas defer the last AdamW
    for _ in range(a.warmup):
        step()
    finish()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    finish()                                                 # inside the timed region: every step's update is applied
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    line = {"metric": "bpr_train_edges_per_sec", "value": a.steps * units / dt, "unit": "edges/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": w.config()}

    if workload in ("nf", "ml"):
        w.eval_once(); torch.cuda.synchronize(); barrier()
        t1 = time.perf_counter(); w.eval_once(); torch.cuda.synchronize(); barrier()
        te = time.perf_counter() - t1                        # every rank ranks its user block; the barrier makes it the slowest rank's time
    if rank == 0 and workload in ("nf", "ml"):
        line["eval"] = {"metric": "full_rank_eval_users_per_sec", "value": w.sh.n_users / te, "ms": te * 1e3,
                        "n_users": w.sh.n_users, "users_per_rank": (w.sh.n_users + world - 1) // world,
                        "includes": "no-grad full-graph forward + fp32 MFMA scoring + masked top-50"}
        if not a.no_kernel_roofline:
            ks = w.kernel_rooflines()
            line["kernels"] = ks
            # dominant kernel = the longest single launch of the step (the grouped projection: one 0.2 ms launch; the four
            # weight-gradient launches of different shapes are listed in "kernels" with their sum)
            dom = max(ks[:2], key=lambda k: k["ms"] / k.get("launches", 1))
            hbm = dom.get("bound") == "hbm"
            line["roofline"] = {"kernel": dom["kernel"], "bound": dom.get("bound", "mfma"),
                                "achieved": dom["gbs"] if hbm else dom["tflops"], "peak": HBM_PEAK_GBS if hbm else MFMA_F32_PEAK_TFLOPS,
                                "unit": "GB/s" if hbm else "TFLOP/s", "frac": dom["frac_hbm"] if hbm else dom["frac_mfma_f32"],
                                "traffic": pmc_traffic_bytes(dom["pmc"]),
                                "algorithmic_flop_per_launch": dom["algorithmic_flop_per_launch"],
                                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                                "ms_per_launch": dom["ms"], "hbm_gbs": dom["gbs"], "frac_hbm": dom["frac_hbm"]}
            line["spmm_roofline"] = spmm_roofline_large(device, a.seed)
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline_nf(w)
    elif rank == 0:
        line.update(w.extras())
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_pg:
        import torch.distributed as dist
        dist.barrier()                                       # rank 0's per-kernel measurements are done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
