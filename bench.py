"""bench.py - the driver's measurement contract for the LLMRec Stage-2 hot path on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...                         # no launcher in the environment: bench.py re-executes itself as N ranks, or refuses
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one mini-batch: on-device BPR sampling + LLM-augmented
triples, the full-graph forward (projections, 20 SpMM, softmax, fusion), the 8 fused BPR+prune
losses, backward and AdamW - the loop body of the reference's Trainer.train (main.py:210-283).

N = 1 workload (BASELINE.json configs[1]): Netflix-SHAPED synthetic data (the real files are not
distributable): U=13187, I=17366, 55146 train edges, d=64, 2 propagation layers, image/text/LLM
side features (512/768/1536-d, 5 attribute keys), batch 1024 + 10 % augmented triples, prune 0.71.
metric value = BPR train edges/s = steps * batch_size / time (the reference's own timer
definition, main.py:200,297); the full-rank eval rate (users/s, main.py:297-303) is reported in
"eval", `python main.py` itself timed by its own epoch timers in "end_to_end". Inputs are resident in HBM
before the timed region. Beside `value`: the oracle parity gate (run before the timed region), `roofline` of
the dominant kernel with PMC traffic, `cpu_baseline` (the oracle port live + the unmodified reference's record),
`spmm_roofline`, the cfg-4-shaped `row_sharded` lines.

N > 1 (one process per GPU, RCCL): BASELINE.json configs[3] - the user-ROW-sharded ID path (llmrec_amd/dist_fused.py:
per layer and direction one I x d exchange) on the 10 M x 1 M x 200 M-edge graph, STRONG scaling, with rank 0's
single-GPU run of the same workload, the row-restricted-forward variant and the Netflix workload as batch-sharded
replicas (llmrec_amd/dp.py) in the same line. --workload nf|ml|cfg4|cfg5 selects one explicitly.
One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before torch / HIP initialise: see llmrec_amd/__init__.py

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 200 (nf / ml), 20 (cfg4), 5 (cfg5)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 20 (nf / ml), 3 (cfg4), 1 (cfg5)")
    ap.add_argument("--workload", default="auto", help="auto | nf (cfg 2) | ml (cfg 3) | synth = cfg4 (row-sharded ID path, 10 M x 1 M x 200 M, d = 64) | "
                                                      "cfg5 (50 M x 5 M x 1.05 B, d = 128, + 5 % augmented triples)")
    ap.add_argument("--synth-scaling", default="weak", help="weak: 2 of the config's 16 user blocks per GPU (cfg 4 / 5 exactly at 8 GPUs); "
                                                           "strong: all 16 blocks = the whole config, split over the ranks")
    ap.add_argument("--synth-blocks", type=int, default=0, help="override the number of user blocks per GPU (weak) / in total (strong)")
    ap.add_argument("--synth-chunks", type=int, default=0, help="item-row chunks per all-reduced message (0 = automatic, >= 32 MB each)")
    ap.add_argument("--synth-exchange", default="all_reduce", help="all_reduce | rs_ag: the per-chunk exchange of the row-sharded step (llmrec_amd/dist_fused.py)")
    ap.add_argument("--synth-dense-backward", action="store_true", help="row-sharded step: run the last layer's backward as dense products (A/B of the operand-sparsity path)")
    ap.add_argument("--synth-dense-forward", action="store_true", help="(the default since round 4: every forward product for every row)")
    ap.add_argument("--synth-restricted-forward", action="store_true", help="row-sharded step: the labelled variant - the last layer's two forward products only in the rows the step reads (forward(needed=...))")
    ap.add_argument("--no-single-gpu-reference", action="store_true", help="N > 1, row-sharded strong scaling: skip rank 0's run of the same workload on one GPU")
    ap.add_argument("--no-row-sharded", action="store_true", help="nf / ml workloads: skip the cfg-4-shaped row-sharded measurements added to the line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity gate that runs before the timed region")
    ap.add_argument("--no-end-to-end", action="store_true", help="nf workload: skip the `python main.py` run timed by the drop-in's own epoch timers (tools/e2e_main.py)")
    ap.add_argument("--no-extra-configs", action="store_true", help="default nf line: skip the MovieLens-shaped (cfg 3) and cfg-5 step times measured in fresh processes")
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


def event_time_deferred(fn, iters: int, warmup: int = 3, stream=None):
    """event_time_ms without the host synchronisation: the launches are enqueued now, the returned callable reads the average (ms) later.
    stream: launch on that side stream (joined to the current one on both sides, on the device)."""
    import torch
    cur = torch.cuda.current_stream()
    st = cur if stream is None else stream
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if stream is not None:
        st.wait_stream(cur)
    with torch.cuda.stream(st):
        for _ in range(warmup):
            fn()
        start.record()
        for _ in range(iters):
            fn()
        stop.record()
    if stream is not None:
        cur.wait_stream(st)

    def read():
        stop.synchronize()
        return start.elapsed_time(stop) / iters
    return read


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
        import llmrec_amd
        self.use_graph = os.environ.get("LLMREC_GRAPH", "1") == "1" and llmrec_amd.graph_replay_safe()

    def step(self):
        """One training step. With the HIP graph: sampler + forward + losses + backward + AdamW are ONE graph
        replay (three between the two exchanges on batch-sharded replicas); nothing else is enqueued."""
        self.step_id += 1
        if self.use_graph:
            if self.fused.graph_exec is None:
                self._capture()                                # the capture's warm-up is a real step
                return self.fused.scal[1:4]
            return self.fused.step()
        u, p, n, nv = self.batcher.next()
        return self.fused.step_eager(u, p, n, nv)

    UNROLL = int(os.environ.get("LLMREC_BENCH_UNROLL", "4"))      # steps per replayed graph in run_steps()

    def _capture(self):
        f = self.fused
        if hasattr(f, "run_steps") and not hasattr(f, "gsz"):
            f.capture(batcher=self.batcher, unroll=self.UNROLL)
        else:                                                  # batch-sharded replicas: graphs between the exchanges
            f.capture(batcher=self.batcher)

    def run_steps(self, n: int):
        """n training steps; single-GPU fused path: graphs of UNROLL steps (FusedStep.run_steps), else n times step()."""
        f = self.fused
        if self.use_graph and hasattr(f, "run_steps") and not hasattr(f, "gsz"):
            self.step_id += n
            if f.graph_exec is None:
                self._capture(); n -= 1                        # (the capture's warm-up is one of the n steps)
            return f.run_steps(n)
        for _ in range(n):
            self.step()

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
            return self.fused.eval_topk(self._eval_q, self.graph.by_user, 50, use_graph=self.use_graph and os.environ.get("LLMREC_EVAL_GRAPH", "1") == "1")

    def config(self):
        return {"workload": "netflix_shaped_cfg2" if self.shape_name == "nf" else "movielens_shaped_cfg3",
                "n_users": self.sh.n_users, "n_items": self.sh.n_items, "n_train_edges": int(self.rows.size),
                "embed_size": self.args.embed_size, "prop_layers": len(eval(self.args.weight_size)),
                "batch_size": self.hp.batch_size, "aug_sample_rate": self.hp.aug_sample_rate,
                "prune_loss_drop_rate": self.hp.prune_loss_drop_rate, "side_features": "image512+text768+llm1536x(1+5)",
                "item_side_operands": ("pre-propagated once at set-up: the step projects A_ui F_k [U x K] (llmrec_amd/fused.py; the features and the graph are "
                                       "constants of a run, A (F W^T + 1 b^T) = (A F) W^T + (A 1) b^T), every W-dependent product runs every step"
                                       if getattr(self.fused, "preprop", False) else "projected, then propagated (the reference's order of operations); "
                                       "llmrec_amd/fused.py pre-propagates iff U <= I"),
                "sampler": "device, inside the step graph (llmrec_sample_batch: BPR triples + LLM-augmented triples, device step counter)", "global_batch": self.hp.batch_size * self.world,
                "parallelism": "single GPU" if not hasattr(self.fused, "gsz") else
                ("dp%d: batch-sharded replicas (llmrec_amd/dp.py), graph + tables replicated, prune over the global batch "
                 "(1 all-gather of %d B) + 1 all-reduce of the %d B gradient bucket per step; eval shards the users"
                 % (self.world, 4 * self.fused.gsz, 4 * self.fused.bucket.numel())),
                "step": ("fused (llmrec_amd/fused.py)" if not hasattr(self.fused, "gsz") else "fused, 3 segments between the 2 exchanges (llmrec_amd/dp.py)")
                        + ((" + HIP graph replay (graphs of %d steps: sampler, forward, losses, backward, AdamW x %d per hipGraphLaunch)" % (self.UNROLL, self.UNROLL)
                            if not hasattr(self.fused, "gsz") and self.UNROLL > 1 else " + HIP graph replay") if self.use_graph else "")}


    def _wgrad_launch_deferred(self, dY_cat, dYu, iters: int = 20):
        """Average duration of the step's weight-gradient launch (the GEMM over all four Linears + its slab-reduction launch), HIP events
        on the stream it is launched on, `iters` launches back to back between the two events, the launch built exactly as the step builds
        it (llmrec_amd/fused.py wgrad_targets; the AdamW update that rides in the step's reduction launch is left out: it would move the
        parameters `iters` times). Enqueued now, read later (event_time_deferred)."""
        import torch
        ops, f = self.ops, self.fused
        targets = f.wgrad_targets(dY_cat, dYu)
        bb = getattr(f, "wgrad_blocks", 0)
        ws = torch.empty(max(ops.linear_wgrad_multi_workspace(targets, bb), 16), dtype=torch.uint8, device=dY_cat.device)
        self._wgrad_probe_keep = (targets, ws)
        return event_time_deferred(lambda: ops.linear_wgrad_multi(targets, ws, block_budget=bb), iters, 3, stream=torch.cuda.Stream())

    def kernel_timings_enqueue(self):
        """The isolated launches of kernel_rooflines() enqueued WITHOUT a host synchronisation (the bench runs them, and the evaluation leg,
        right ahead of the warm-up steps: the timed region then does not begin inside the power-management transient of the first ~12 ms
        of load, profiles/experiments/r05_step_chain.md). Needs the gradient buffers a training step left."""
        import torch
        ops, sh, d, f = self.ops, self.sh, self.args.embed_size, self.fused
        pre = {"proj": event_time_deferred(f._project_all, 20)}
        if f.gemm == "bf16x3" and d == 64:
            pre["wgrad"] = self._wgrad_launch_deferred(f.dU_cat if f.preprop else f.dP_cat, f.dP_usr)
            pre["act_n"] = f.act_n.clone() if getattr(f, "wgrad_rows", False) else None   # the row list's length of the step the launch reads
        Xi = torch.randn(sh.n_items, d, device=self.device)
        a = self.graph.ui.fwd
        pre["spmm"] = event_time_deferred(lambda: ops.spmm_raw(a, Xi), 50)
        return pre

    def in_graph_durations(self, iters: int = 20):
        """Durations INSIDE the replayed step graph, live: the step is re-captured with llmrec_timestamp launches (a single lane storing the
        device's constant-rate counter) around the whole step, the projection launch and the weight-gradient + reduction launches
        (FusedStep.stamps), replayed `iters` times, the slot differences averaged. HIP events cannot do this (ROCm refuses event-record
        nodes for external events in a captured graph); the committed rocprofv3 summary of this command is the cross-check
        (in_step_us_rocprof). Each bracket includes the one or two dispatch gaps between the stamp launches and the bracketed kernel.
        Returns {"projection_us", "wgrad_us", "span_us", "rate_hz"} or None (path without a single captured graph)."""
        import torch
        from llmrec_amd import _lib
        f = self.fused
        if not self.use_graph or not hasattr(f, "run_steps") or hasattr(f, "gsz") or getattr(f, "batcher", None) is None:
            return None
        rate = _lib.query("llmrec_timestamp_rate_hz")
        if rate <= 0:
            return None
        stamps = f.stamps = torch.zeros(f.STAMP_SLOTS, dtype=torch.int64, device=self.device)
        acc, n = [0.0, 0.0, 0.0], 0
        try:
            f.capture(batcher=self.batcher, unroll=1)
            for _ in range(3):
                f.graph_exec.replay()
            torch.cuda.synchronize()
            for _ in range(iters):
                for _ in range(6):                              # back to back, as in the timed region: the slots keep the LAST replay's stamps
                    f.graph_exec.replay()
                torch.cuda.synchronize()
                t = stamps.tolist()
                if not (t[0] <= t[1] <= t[2] and t[5] >= t[0]):
                    continue
                acc[0] += t[2] - t[1]; acc[1] += (t[4] - t[3]) if t[4] >= t[3] > 0 else 0.0; acc[2] += t[5] - t[0]
                n += 1
        finally:
            #  the instrumented graph's llmrec_timestamp nodes write to `stamps`: back to the uninstrumented graphs on EVERY exit
            # path, with the tensor still alive, before anything can replay the step again
            f.stamps = None
            torch.cuda.synchronize()
            f.capture(batcher=self.batcher, unroll=self.UNROLL)
            del stamps
        if n == 0:
            return None
        us = lambda ticks: ticks / n / rate * 1e6
        return {"projection_us": us(acc[0]), "wgrad_us": us(acc[1]) if acc[1] > 0 else None, "rate_hz": rate, "replays": n,
                "instrumented_span_us": us(acc[2]),           # (of the INSTRUMENTED single-step graph: not the step's span - that is in profiles/r*_step_timeline.txt)
                "how": "llmrec_timestamp launches inside the re-captured step graph (FusedStep.stamps), averaged over the replays"}

    # ---- per-kernel roofline (dominant kernels of this workload, timed in isolation) -------------
    def kernel_rooflines(self, pre=None):
        """pre: kernel_timings_enqueue()'s deferred timings (else measured here)."""
        import torch
        ops, sh, d = self.ops, self.sh, self.args.embed_size
        if pre is None:
            pre = self.kernel_timings_enqueue()
        out = []
        W = self.model.item_trans.weight.detach(); b = self.model.item_trans.bias.detach()
        X = self.feats["attr/" + self.keys[0]]
        # the operands the step's GEMM launches actually stream (pre-propagated item side: A_ui F_k with U rows instead of F_k with I rows)
        feats = [j[0] for j in self.fused.projection_jobs()]
        flop_all = sum(2.0 * x.shape[0] * x.shape[1] * d for x in feats)
        byts_all = sum(4.0 * (x.shape[0] * x.shape[1] + d * x.shape[1] + x.shape[0] * d) for x in feats)
        ms = pre["proj"]()
        t_how = ("HIP events on the launch's stream around 20 launches back to back (the replayed step graph cannot carry timing events: ROCm refuses "
                 "external event-record nodes in a captured graph); in_step_us_rocprof = the same kernel's average duration inside the replayed "
                 "graph in the committed rocprofv3 --kernel-trace --stats summary of this command")
        bf = self.fused.gemm == "bf16x3"
        out.append({"kernel": ("linear_fwd_grouped_bf16x3_kernel (all 8 projections, one launch; 3-term bf16 split, 6 bf16 MFMAs: "
                               "HBM-bound on the X stream - tflops/frac_mfma_f32 are fp32-EQUIVALENT figures)") if bf else
                              "linear_fwd_grouped_kernel (all 8 projections of one forward, one launch, exact fp32 MFMA)",
                    "pmc": [("linear_fwd_grouped_bf16x3_kernel" if bf else "linear_fwd_grouped_kernel", 1)],
                    "bound": "hbm" if bf else "mfma", "calls_per_step": 1, "ms": ms, "timing": t_how,
                    "tflops": flop_all / ms / 1e9, "frac_mfma_f32": flop_all / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                    "gbs": byts_all / ms / 1e6, "frac_hbm": byts_all / ms / 1e6 / HBM_PEAK_GBS,
                    "algorithmic_flop_per_launch": flop_all, "algorithmic_bytes_per_launch": byts_all,
                    "algorithmic_flop_per_step": flop_all, "algorithmic_bytes_per_step": byts_all, "launches": 1})
        # the step's weight gradients: ONE multi-target launch (item_trans x5, user_trans, text_trans, image_trans) + its slab reduction
        bf_ok = bf and d == 64
        # operands = the gradient buffers the last training step left (the chip is power-bound in this kernel and its clock depends
        # on the data: N(0, 1) stand-ins for the ~1e-5-sized gradients make the same launch ~30 % slower than it is in the step)
        dY_cat = self.fused.dU_cat if self.fused.preprop else self.fused.dP_cat
        dYu = self.fused.dP_usr
        if bf_ok:
            ms = pre["wgrad"]()
            # ALGORITHMIC bytes / flop of the rows this launch streams: a row-listed pair (the five attribute streams, llmrec_amd/fused.py) reads
            # the listed rows of dY and X only - the list length of the last training step, read back here
            listed_rows = None
            byts_w, flop_w = 0.0, 0.0
            for pairs, dW, db, acc in self.fused.wgrad_targets(dY_cat, dYu):
                for pr in pairs:
                    Mp, Kp = pr[1].shape
                    if len(pr) > 3 and pr[3] is not None:
                        listed_rows = int((pre.get("act_n") if pre.get("act_n") is not None else pr[3][1]).item())
                        Mp = listed_rows
                    byts_w += 4.0 * (Mp * Kp + d * Kp + Mp * d); flop_w += 2.0 * Mp * Kp * d
            out.append({"kernel": "linear_wgrad_bf16x3_v2_multi_kernel + reduce_chunks_multi_kernel: the weight gradients of all four Linears "
                                  "(item_trans x5, user_trans, text_trans, image_trans) in one launch; 3-term bf16 split: tflops are fp32-EQUIVALENT",
                        "pmc": [("linear_wgrad_bf16x3_v2_multi_kernel", 1), ("reduce_chunks_multi_kernel", 1)],
                        "launches": 1, "avg_launch_ms": ms, "timing": t_how,
                        "bound": "hbm", "calls_per_step": 1, "ms": ms, "tflops": flop_w / ms / 1e9, "frac_mfma_f32": flop_w / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                        "gbs": byts_w / ms / 1e6, "frac_hbm": byts_w / ms / 1e6 / HBM_PEAK_GBS,
                        "rows": ({"listed_rows_of_the_attribute_pairs": listed_rows, "of": int(dY_cat.shape[0]),
                                  "what": "the attribute streams' gradient is exactly zero outside the rows the batch reaches (its users + the users adjacent to its "
                                          "items): the launch streams those rows only (llmrec_wgrad_problem_t.row_list); bytes / flop count the streamed rows"}
                                 if listed_rows is not None else "every row (dense launch)"),
                        "dense_equivalent_bytes": byts_all,
                        "note": "power-bound, not bandwidth-bound: the shader clock averages 1.37 GHz in this kernel (2.33 GHz for its load stream alone, "
                                "1.88 GHz for its MFMAs alone; profiles/experiments/r03_wgrad.md)",
                        "algorithmic_flop_per_launch": flop_w, "algorithmic_bytes_per_launch": byts_w,
                        "algorithmic_flop_per_step": flop_w, "algorithmic_bytes_per_step": byts_w})
        a = self.graph.ui.fwd
        ms = pre["spmm"]()
        byts = 4.0 * a.nnz + 4.0 * (a.n_rows + 1) + 4.0 * a.n_rows + 4.0 * d * a.n_cols + 4.0 * d * a.n_rows
        out.append({"kernel": "spmm_kernel<16,1,4> (ui, d = 64, NF scale: L2-resident, launch-bound; one launch, no finalize pass)", "calls_per_step": 12,
                    "ms": ms, "gbs": byts / ms / 1e6, "frac_hbm": byts / ms / 1e6 / HBM_PEAK_GBS, "edges_per_s": a.nnz / ms * 1e3})
        return out


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] / [4]: the ID-embedding path on a synthetic graph, users row-sharded over the ranks
# ---------------------------------------------------------------------------------------------------------------
SYNTH_CONFIGS = {
    # 16 user blocks per config; one block = 1/16 of the users and of the edges
    "cfg4": {"n_users": 10_000_000, "n_items": 1_000_000, "n_edges": 200_000_000, "d": 64, "layers": 2, "aug_rate": 0.0},
    "cfg5": {"n_users": 50_000_000, "n_items": 5_000_000, "n_edges": 1_050_000_000, "d": 128, "layers": 2, "aug_rate": 0.05},
}
SYNTH_BLOCKS = 16


def synth_blocks_of_rank(scaling: str, blocks: int, rank: int, world: int):
    """(total user blocks of the job, the blocks of `rank`): strong scaling splits all SYNTH_BLOCKS over the ranks, weak gives every rank
    `blocks` (default 2) of them."""
    if scaling == "strong":
        total = blocks or SYNTH_BLOCKS
        if total % world:
            raise SystemExit("strong scaling: %d blocks do not divide over %d ranks" % (total, world))
        return total, list(range(rank * (total // world), (rank + 1) * (total // world)))
    per = blocks or 2
    return per * world, list(range(rank * per, (rank + 1) * per))


def row_sharded_plan(name: str, scaling: str, world: int, exchange: str = "all_reduce", n_chunks: int = 0, blocks: int = 0, batch_local: int = 1024,
                     sparse_forward: bool = False):
    """The shape of an N-rank row-sharded line WITHOUT launching anything (tests/test_bench_line_cpu.py): per-rank
    users / edges, the user partition (llmrec_amd.dist.user_block over the job's users), the chunks of the exchanged I x d message and the
    bytes a step exchanges (llmrec_amd.dist_fused.plan_chunks / message_plan - the functions ShardedFusedID itself uses)."""
    from llmrec_amd import dist as ldist
    from llmrec_amd.dist_fused import plan_chunks, message_plan
    cfg = SYNTH_CONFIGS[name]
    bu, be_ = cfg["n_users"] // SYNTH_BLOCKS, cfg["n_edges"] // SYNTH_BLOCKS
    ranks = []
    for r in range(world):
        total, mine = synth_blocks_of_rank(scaling, blocks, r, world)
        u0, u1 = ldist.user_block(bu * total, r, world)
        assert (u0, u1) == (mine[0] * bu, (mine[-1] + 1) * bu), "the block split and the contiguous user partition disagree"
        ranks.append({"rank": r, "blocks": mine, "users": [u0, u1], "edges_nominal": be_ * len(mine)})
    n_aug = int(batch_local * cfg["aug_rate"])
    chunks = plan_chunks(cfg["n_items"], cfg["d"], world, n_chunks or None, exchange)
    return {"config": {"workload": "synthetic_%s_shape_row_sharded_id_path" % name, "scaling": scaling, "n_users_global": bu * total,
                       "n_items": cfg["n_items"], "users_per_gpu": bu * len(ranks[0]["blocks"]), "edges_per_gpu_nominal": be_ * len(ranks[0]["blocks"]),
                       "embed_size": cfg["d"], "prop_layers": cfg["layers"], "batch_per_gpu": batch_local, "augmented_triples_per_gpu": n_aug,
                       "global_batch": (batch_local + n_aug) * world, "user_blocks_total": total},
            "ranks": ranks, "chunks": chunks,
            "messages": message_plan(cfg["n_items"], cfg["d"], cfg["layers"], batch_local + n_aug, len(chunks), exchange, sparse_forward)}


class RowSharded:
    """llmrec_amd/dist_fused.ShardedFusedID on cfg-4 / cfg-5-shaped synthetic graphs. The graph is generated in 16
    user blocks with per-block seeds, so the GLOBAL graph does not depend on the number of ranks: weak scaling gives
    every rank `blocks` of them (2 per GPU = the whole config at 8 GPUs), strong scaling splits all 16 over the ranks."""

    def __init__(self, name, scaling, blocks, seed, device, rank, world, n_chunks=0, batch_local=1024, exchange="all_reduce", single=False,
                 sparse_backward=True, sparse_forward=True):
        import torch
        from llmrec_amd import dist as ldist, synth
        from llmrec_amd.dist_fused import ShardedFusedID
        cfg = SYNTH_CONFIGS[name]
        self.name, self.cfg, self.scaling, self.device, self.rank, self.world = name, cfg, scaling, device, rank, world
        bu, be_ = cfg["n_users"] // SYNTH_BLOCKS, cfg["n_edges"] // SYNTH_BLOCKS
        total, mine = synth_blocks_of_rank(scaling, blocks, rank, world)
        self.blocks_total, self.blocks_mine = total, len(mine)
        self.comm, self.backend = ldist.Comm(single=single), ldist.HipBackend()
        t0 = time.perf_counter()
        rows, cols = [], []
        for k, b in enumerate(mine):
            r, c = synth.bipartite_edges_device(bu, cfg["n_items"], be_, seed * 131 + b, device)
            rows.append(r + k * bu); cols.append(c)
        rows, cols = torch.cat(rows), torch.cat(cols)
        self.nnz_local = int(rows.numel())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        n_local = bu * len(mine)
        self.graph = ldist.ShardedGraph.build(rows, cols, n_local, cfg["n_items"], rank * n_local, self.comm, self.backend)
        del rows, cols
        torch.cuda.synchronize(); t2 = time.perf_counter()
        self.n_aug = int(batch_local * cfg["aug_rate"])
        self.B = batch_local + self.n_aug
        self.batch_local = batch_local
        self.step_obj = ShardedFusedID(self.graph, self.comm, self.backend, cfg["d"], cfg["layers"], bu * total, seed, 1e-4, self.B, 0.71, 1e-5,
                                       n_chunks=n_chunks or None, batch_size_flag=float(batch_local * world),      # the FLAG, not B + aug (main.py:340)
                                       exchange=exchange, sparse_backward=sparse_backward, sparse_forward=sparse_forward)
        if self.n_aug:                                        # the LLM-augmented triples of main.py:216-224: a per-user (pos, neg) table
            g = torch.Generator(device=device); g.manual_seed(seed + 17 + rank)
            self.aug_pos = torch.randint(0, cfg["n_items"], (n_local,), generator=g, device=device)
            self.aug_neg = torch.randint(0, cfg["n_items"], (n_local,), generator=g, device=device)
        torch.cuda.synchronize()
        self.ingest = {"generate_s": t1 - t0, "csr_build_and_plans_s": t2 - t1, "tables_s": time.perf_counter() - t2,
                       "hbm_allocated_gb": torch.cuda.memory_allocated(device) / 1e9, "hbm_peak_gb": torch.cuda.max_memory_allocated(device) / 1e9}
        self.units_per_step = batch_local * world              # the reference's timer counts batch_size per step (main.py:203)
        self.last = None

    def _triples(self):
        import torch
        st = self.step_obj
        u, p, n = st.be.sample(st.seed, st.step_id, st.exist, st.I, st.g.by_user, self.batch_local)
        if self.n_aug:                                        # extra triples for a sample of the batch's users
            ua = u[: self.n_aug]
            u, p, n = torch.cat([u, ua]), torch.cat([p, self.aug_pos[ua]]), torch.cat([n, self.aug_neg[ua]])
        return u, p, n

    def step(self):
        self.last = self.step_obj.step(self._triples())
        return self.last

    def config(self):
        c, st = self.cfg, self.step_obj
        return {"workload": "synthetic_%s_shape_row_sharded_id_path" % self.name, "scaling": self.scaling,
                "n_users_global": c["n_users"] // SYNTH_BLOCKS * self.blocks_total, "n_items": c["n_items"],
                "users_per_gpu": st.U, "edges_per_gpu": self.nnz_local, "embed_size": c["d"], "prop_layers": c["layers"],
                "batch_per_gpu": self.batch_local, "augmented_triples_per_gpu": self.n_aug, "global_batch": self.B * self.world,
                "prune_loss_drop_rate": 0.71, "user_blocks_total": self.blocks_total,
                "last_layer_forward": ("restricted to the rows the step reads (labelled variant, --synth-restricted-forward)" if st.sparse_forward
                                       else "dense: every forward product for every row"),
                "last_layer_backward": "products with exactly-zero operand rows skip them" if st.sparse_backward else "dense",
                "parallelism": ("user-row-sharded x%d (llmrec_amd/dist_fused.py): item tables replicated, per layer and direction one I x d "
                                "all-reduce in %d asynchronous chunks behind the next chunk's SpMM, BPR gradient rows by all-gather"
                                % (self.world, len(st.chunks)))}

    def sampled_row_parity(self, n_rows=192):
        """Every SpMM of one forward at FULL size, checked on sampled rows (plus the longest ones): row r of each layer's
        output is recomputed on the CPU in fp64 from the previous layer's GPU tensor and the row's neighbour list."""
        import numpy as np
        import torch
        st, g = self.step_obj, self.graph
        if self.world != 1:
            return None
        st.forward()
        torch.cuda.synchronize()
        rng = np.random.default_rng(5)
        worst = 0.0

        def check(csr, X, Y, softmax):
            nonlocal worst
            deg = (csr.rowptr[1:] - csr.rowptr[:-1])
            top = torch.topk(deg, 4).indices.cpu().numpy()
            rows = np.unique(np.concatenate([rng.integers(0, csr.n_rows, size=n_rows), top]))
            rp = csr.rowptr
            for r in rows:
                s, e = int(rp[r]), int(rp[r + 1])
                cols = csr.colidx[s:e].long()
                acc = X[cols].double().sum(0) * float(csr.row_scale[r]) if e > s else torch.zeros(X.shape[1], dtype=torch.float64, device=X.device)
                if softmax:
                    acc = torch.softmax(acc, dim=-1)
                want = acc
                got = Y[r].double()
                worst = max(worst, float((got - want).abs().max() / max(float(want.abs().max()), 1e-30)))
        prev = st.item_tab.detach()
        for l in range(st.L):
            last = l == st.L - 1
            check(g.ui_fwd, prev, st.Ul[l], last)
            check(g.iu_fwd, st.Ul[l], st.Il[l], last)
            prev = st.Il[l]
        return {"rows_per_spmm": n_rows + 4, "spmm_checked": 2 * st.L, "max_rel_err_vs_fp64": worst, "ok": bool(worst < 1e-4)}

    def extras(self, ms_per_step):
        import torch
        st, c = self.step_obj, self.cfg
        nnz = torch.tensor([self.nnz_local], dtype=torch.float64, device=self.device)
        self.comm.all_reduce_(nnz)
        nnz = float(nnz.item())
        L, d = c["layers"], c["d"]
        # algorithmic bytes of the step's 4 L SpMMs (4 nnz + 8 rows + 4 d (rows + cols) each; here per rank)
        U, I = st.U, st.I
        per_pair = 2 * (4.0 * self.nnz_local + 8.0 * (U + I) / 2 + 4.0 * d * (U + I))
        out = {"propagated_edges_per_step": nnz * 4 * L, "propagated_edges_per_sec": nnz * 4 * L / ms_per_step * 1e3,
               "spmm_algorithmic_bytes_per_step_per_gpu": per_pair * 2 * L, "ingest": self.ingest, "messages": st.message_bytes_per_step()}
        if self.last is not None:
            out["loss"] = float(self.last[0]); out["mf_emb"] = [float(x) for x in self.last[1]]
        return out

    def eval_sample(self, n_users=65536, K=50):
        """Full-rank scoring + masked top-50 of a sample of this rank's users against all items (R9 at the config's size)."""
        import torch
        st = self.step_obj
        n = min(n_users, st.U)
        q = torch.arange(0, st.U, max(1, st.U // n), device=self.device, dtype=torch.int64)[:n]
        st.forward()
        st.eval_topk(q, K, forward=False)                        # warm-up: the workspace (lists + the item table in fragment order) is allocated here
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            idx, _ = st.eval_topk(q, K, forward=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        flop = 2.0 * q.numel() * st.I * self.cfg["d"]
        return {"users": int(q.numel()), "items": st.I, "K": K, "ms": dt * 1e3, "users_per_s": q.numel() / dt, "tflops": flop / dt / 1e12,
                "frac_mfma_f32": flop / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, "lists_full": bool((idx[:, K - 1] >= 0).all())}

    def spmm_times_ms(self, iters=3):
        """In-situ size, each direction alone: the numbers the SpMM roofline is computed from."""
        import torch
        st, g = self.step_obj, self.graph
        d = self.cfg["d"]
        res = {}
        Xi, Xu = st.item_tab.detach(), st.Ul[0]
        for name, csr, X, Y in (("ui_fwd", g.ui_fwd, Xi, st.hU), ("iu_fwd", g.iu_fwd, Xu, st.tmpI), ("iu_bwd_pattern", st.R_user, Xi, st.hU), ("ui_bwd_weighted_not_on_the_path", g.ui_bwd, Xu, st.tmpI)):
            ms = event_time_ms(lambda: self.backend.spmm(csr, X, out=Y), iters, warmup=1)
            alg = 4.0 * csr.nnz + 8.0 * csr.n_rows + 4.0 * d * (csr.n_cols + csr.n_rows)
            res[name] = {"ms": ms, "edges_per_s": csr.nnz / ms * 1e3, "algorithmic_gbs": alg / ms / 1e6, "frac_hbm_algorithmic": alg / ms / 1e6 / HBM_PEAK_GBS,
                         "gather_gbs": csr.nnz * 4.0 * d / ms / 1e6}
        return res


def kernel_time_shares():
    """Per-kernel summed GPU time per step from the newest committed rocprofv3 --kernel-trace --stats summary of THIS bench
    (profiles/r*_bench_nf_kernel_stats*.csv, written by tools/rocpd_stats.py): the line then shows the same ranking as the
    summary - e.g. the SpMM class, whose launches overlap the GEMMs on side streams, next to the roofline's dominant
    kernel on the critical path. None when no summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_nf_kernel_stats*.csv")))   # by name: the newest round's last
    if not files:
        return None
    path = files[-1]
    rows = []
    try:
        with open(path) as f:
            for rec in csv.reader(line for line in f if not line.startswith("#")):
                if rec and rec[0] != "name":
                    rows.append((rec[0], int(rec[1]), float(rec[5])))
    except Exception:
        return None
    classes = {}
    for name, calls, per_step in rows:
        short = name.split("(")[0].replace("void ", "").replace("llmrec::", "")
        cls = short.split("<")[0]
        c = classes.setdefault(cls, {"per_step_us": 0.0, "kernels": []})
        c["per_step_us"] += per_step
        c["kernels"].append(short)
    total = sum(c["per_step_us"] for c in classes.values())
    top = sorted(classes.items(), key=lambda kv: -kv[1]["per_step_us"])[:8]
    return {"file": os.path.basename(path), "sum_of_kernel_time_per_step_us": total,
            "note": "summed launch durations per step; launches on the step's five streams overlap, so the sum exceeds the step time",
            "classes": {k: {"per_step_us": round(v["per_step_us"], 2), "share_of_kernel_time": round(v["per_step_us"] / total, 4)} for k, v in top}}


def rocprof_avg_us(parts):
    """Sum of the average durations (us) of the named kernels in the newest committed rocprofv3 --kernel-trace --stats summary of this
    bench (profiles/r*_bench_nf_kernel_stats*.csv): the in-step cross-check of the roofline's own event timing. None if absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_nf_kernel_stats*.csv")))
    if not files:
        return None
    total = 0.0
    try:
        with open(files[-1]) as f:
            rows = [rec for rec in csv.reader(line for line in f if not line.startswith("#")) if rec and rec[0] != "name"]
    except Exception:
        return None
    for key, launches in parts:
        hit = [float(rec[3]) for rec in rows if key in rec[0]]
        if not hit:
            return None
        total += launches * hit[0]
    return {"file": os.path.basename(files[-1]), "avg_us": round(total, 2)}


def pmc_traffic_bytes(parts):
    """HBM-side bytes of one "launch" as the roofline defines it, from the newest committed PMC pass over THIS bench
    (profiles/r*_pmc_bench_step.json: separate rocprofv3 --pmc runs with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled per its gfx950 note, WRITE_SIZE taken as reported;
    both in KB). parts: [(kernel-name substring, launches of it per roofline launch)]. Returns
    (bytes or None, {"file", "kernels"}): None when a kernel is missing from the pass."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_step.json")))
    if not files:
        return None, None
    path = files[-1]
    try:
        data = json.load(open(path))
    except Exception:
        return None, None
    total, names = 0.0, []
    for key, launches in parts:
        hit = [(name, c) for name, c in data.items() if key in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
        if not hit:
            return None, {"file": os.path.basename(path), "missing": key}
        names.append(hit[0][0])
        total += launches * (2.0 * hit[0][1]["FETCH_SIZE"]["mean"] + hit[0][1]["WRITE_SIZE"]["mean"]) * 1024.0
    return total, {"file": os.path.basename(path), "kernels": names}


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
        # the same product through the operand's own CSR in row order (round 6: the plan's length-class ordered, permuted CSR is the default)
        _, key = ops.spmm_shape(d, a.nnz)
        keep = a.plans.get(key)
        ms_plain = None
        if keep is not None and keep.slot_row is not None:
            a.plans[key] = ops.SpmmPlan.build(a.rowptr, *key)
            ms_plain = event_time_ms(lambda: ops.spmm_raw(a, X, out=Y), 10, warmup=2)
            a.plans[key] = keep
        alg = 4.0 * nnz + 4.0 * (a.n_rows + 1) + 4.0 * a.n_rows + 4.0 * d * a.n_cols + 4.0 * d * a.n_rows
        gather = nnz * (4.0 + 4.0 * d) + 4.0 * d * a.n_rows
        res[name] = {"ms": ms, "edges_per_s": nnz / ms * 1e3, "algorithmic_gbs": alg / ms / 1e6,
                     "frac_hbm_algorithmic": alg / ms / 1e6 / HBM_PEAK_GBS, "no_reuse_gather_gbs": gather / ms / 1e6,
                     "frac_gather_model": gather / ms / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "no_reuse_gather_bytes": gather,
                     "n_long_rows": a.plan.n_long, "ms_rows_in_id_order": ms_plain}
    out = {"graph": {"n_users": n_users, "n_items": n_items, "nnz": int(nnz), "d": d},
           "fractions": "frac_hbm_algorithmic = SURVEY 8(d)'s bytes (every X row read ONCE: 4 nnz + 8 rows + 4 d (rows + cols)) / time / 8 TB/s - north_star's >= 0.40 "
                        "target, UNMET on a structureless graph; frac_gather_model = the no-reuse gather model (nnz (4 + 4 d) + 4 d rows: every edge fetches "
                        "its 256-B row) / time / 8 TB/s - above 1 means the L2 served part of the gathers (DESIGN.md section 4)", **res}
    pmc = spmm_pmc_traffic()
    if pmc is not None:
        for name in ("ui", "iu"):
            if name in pmc.get("directions", {}):
                t = pmc["directions"][name]
                out[name]["traffic"] = t["hbm_bytes_per_launch"]
                out[name]["traffic_over_algorithmic"] = t["hbm_bytes_per_launch"] / out[name]["algorithmic_bytes"]
                out[name]["l2_hit_rate"] = t.get("l2_hit_rate")
        out["traffic_source"] = pmc.get("file")
    cp = spmm_cache_policy_record()
    if cp is not None:
        out["cache_policy"] = cp
    xm = spmm_xcd_map_record()
    if xm is not None:
        out["xcd_map"] = xm
    return out


def spmm_cache_policy_record():
    """The round-5 cache-policy experiment (non-temporal loads for cold columns: profiles/r*_pmc_spmm_nt.json, tools/spmm_nt.sh) as six numbers:
    default policy vs the best-traffic threshold. None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spmm_nt.json")))
    if not files:
        return None
    try:
        v = {x["H"]: x for x in json.load(open(files[-1]))["variants"]}
        base, best = v[0], min((x for h, x in v.items() if h > 0), key=lambda x: x["ui"].get("hbm_bytes_per_launch", 1e30))
        return {"experiment": "nt_loads_for_cold_columns", "outcome": "negative" if best["ui"]["ms"] >= base["ui"]["ms"] else "positive", "H": best["H"],
                "ui_ms": [_r(base["ui"]["ms"], 4), _r(best["ui"]["ms"], 4)], "ui_l2_hit": [_r(base["ui"].get("l2_hit_rate"), 3), _r(best["ui"].get("l2_hit_rate"), 3)],
                "ui_traffic_gb": [_r(base["ui"].get("hbm_bytes_per_launch", 0) / 1e9, 3), _r(best["ui"].get("hbm_bytes_per_launch", 0) / 1e9, 3)],
                "file": os.path.basename(files[-1])}
    except Exception:
        return None


def spmm_xcd_map_record():
    """The round-6 block -> row map experiment (profiles/r*_pmc_spmm_xcd.json, tools/spmm_xcd.sh): linear vs XCD-contiguous on the standard
    generator and on the community-ordered graph, rows = users. None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spmm_xcd.json")))
    if not files:
        return None
    try:
        rows = json.load(open(files[-1]))["rows"]
        pick = lambda g, x: next(r for r in rows if r["graph"] == g and r["dir"] == "ui" and r["xcd_contiguous"] == x)
        o = {"experiment": "xcd_contiguous_block_to_row_map", "file": os.path.basename(files[-1])}
        for g in ("std", "comm"):
            a_, b_ = pick(g, 0), pick(g, 1)
            o[g] = {"ui_ms": [_r(a_["ms"], 4), _r(b_["ms"], 4)], "ui_traffic_gb": [_r(a_.get("traffic_gb"), 4), _r(b_.get("traffic_gb"), 4)],
                    "ui_l2_hit": [_r(a_.get("l2_hit_rate"), 3), _r(b_.get("l2_hit_rate"), 3)]}
        o["outcome"] = "positive on community-ordered ids, neutral on the standard generator; off by default (not combinable with the length-class order)"
        return o
    except Exception:
        return None


def spmm_pmc_traffic():
    """HBM-side bytes per launch of the product SpMM at this function's graph (2 M x 1 M x 40 M, d = 64), both directions, from the newest
    committed PMC pass (profiles/r*_pmc_spmm_40M.json, tools/pmc_spmm.sh: rocprofv3 --pmc in separate passes with --kernel-trace only;
    2 x FETCH_SIZE + WRITE_SIZE in KB as MI355X_MICROARCH.md prescribes for gfx950). None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spmm_40M.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    d["file"] = os.path.basename(files[-1])
    return d


ORACLE_PARAMS = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_trans.weight",
                 "user_trans.bias", "item_trans.weight", "item_trans.bias", "user_id_embedding.weight", "item_id_embedding.weight"]


def _oracle_state(w: "NetflixShaped", init_params=None):
    """CPU copies of everything the oracle needs for this workload (graph, features, the parameters - the model's current ones, or the
    snapshot `init_params` {name: tensor})."""
    import numpy as np
    import scipy.sparse as sp
    from oracle import oracle as O
    sh = w.sh
    cfg = O.Config.from_args(vars(w.args), w.keys)
    R = sp.csr_matrix((np.ones(w.rows.size, dtype=np.float32), (w.rows, w.cols)), shape=(sh.n_users, sh.n_items))
    a_ui, a_iu = O.normalized_graphs(R)
    feats = {k: v.cpu() for k, v in w.feats.items()}
    sd = w.model.state_dict() if init_params is None else init_params
    params = {k: sd[k].detach().cpu().clone().requires_grad_(True) for k in ORACLE_PARAMS}
    return O, cfg, R, a_ui, a_iu, feats, params


class ParityGate:
    """parity_check in two phases: capture() runs the GPU side BEFORE the timed region - the first `steps` steps of the
    timed path and one evaluation, every compared tensor cloned on the device - and compare() runs the CPU oracle against those snapshots
    AFTER it, so the timed region does not begin behind ~10 s of host-only work."""

    def __init__(self, w: "NetflixShaped", n_eval_users: int = 256, steps: int = 2, tol: float = 1e-4):
        self.w, self.n_eval_users, self.steps, self.tol = w, n_eval_users, steps, tol
        self.snap, self.init, self.eval_snap = [], None, None

    def capture(self):
        import torch
        w, f = self.w, self.w.fused
        assert w.world == 1 and w.step_id == 0, "the parity gate runs on a fresh single-GPU workload"
        self.t0 = time.perf_counter()
        gp = dict(w.model.named_parameters())
        cl = lambda t: t.detach().clone()
        self.init = {nm: cl(gp[nm]) for nm in ORACLE_PARAMS}
        for s in range(self.steps):
            sn = {"pre": {nm: cl(gp[nm]) for nm in ORACLE_PARAMS},
                  "pre_mv": {nm: tuple(cl(t) for t in w.opt.state[gp[nm]]) if gp[nm] in w.opt.state else None for nm in ORACLE_PARAMS}}
            if w.use_graph:
                w.step()                                         # step 1 = the capture's eager warm-up, then graph replays
                torch.cuda.synchronize()
                st = f.static
                nv = int(st["n_valid"])
                sn["batch"] = tuple(st[k][:nv].cpu().numpy() for k in ("users", "pos", "neg"))
            else:
                ud, pd_, nd, nvd = w.batcher.next()
                w.step_id += 1
                f.step_eager(ud, pd_, nd, nvd)
                torch.cuda.synchronize()
                nv = int(nvd)
                sn["batch"] = (ud[:nv].cpu().numpy(), pd_[:nv].cpu().numpy(), nd[:nv].cpu().numpy())
            out = f.outputs()
            sn["named"] = {k: cl(v) for k, v in dict(E_u=out[0], E_i=out[1], img_i=out[2], txt_i=out[3], img_u=out[4], txt_u=out[5], P_usr=out[6],
                                                     prof_u=out[8], prof_i=out[9]).items()}
            sn["att_u"] = {k: cl(out[10][k]) for k in w.keys}
            sn["att_i"] = {k: cl(out[11][k]) for k in w.keys}
            sn["bpr_out"], sn["loss"] = cl(f.out[: f.n_prob]), float(f.scal[1])
            sn["grad"] = {nm: cl(gp[nm].grad) for nm in ORACLE_PARAMS}
            sn["post"] = {nm: cl(gp[nm]) for nm in ORACLE_PARAMS}
            sn["post_mv"] = {nm: tuple(cl(t) for t in w.opt.state[gp[nm]]) if gp[nm] in w.opt.state else None for nm in ORACLE_PARAMS}
            self.snap.append(sn)
        idx, _ = w.eval_once()                                   # evaluation: forward with the post-step parameters + scoring + masked top-50
        torch.cuda.synchronize()
        self.eval_snap = {"idx": cl(idx), "E_u": cl(f.E_u), "E_i": cl(f.E_i)}
        self.capture_s = time.perf_counter() - self.t0
        return self

    def compare(self):
        return _parity_compare(self)


def _parity_compare(gate: "ParityGate"):
    """The parity gate printed next to the timings: the first `steps` training steps of THIS
    workload on the path that is timed (fused step, HIP-graph replay from the second step on, in-graph device
    sampler) and one evaluation, against the CPU oracle (oracle/oracle.py = the reference's arithmetic,
    Models.py:127-199, main.py:228-278, utility/batch_test.py:21-36) fed with the identical samples read back
    from the device. Per step: the forward outputs of the reference's 14-tuple, the 8 (mf, emb) BPR pairs,
    the loss, the 10 gradients and the post-AdamW parameters, as max |a - b| / max |b| per tensor (gate: 1e-4 on every stage
    given the previous one - forward, losses, gradients; 1e-5 on the AdamW kernel given the GPU's own gradients; 1e-4 end to end
    on E_u / E_i after the steps; the per-entry end-to-end parameter figure is reported, see param_note). Evaluation:
    E_u / E_i after the steps, and the ranked top-50 lists of `n_eval_users` users, which must EQUAL the
    reference ranking rule (score desc, item id asc) applied to the kernel's bit-exact fp32 fma-chain scores;
    the lists from the oracle's own embeddings are compared too (near-ties may swap there: reported, not gated)."""
    import numpy as np
    import torch
    w, steps, tol, n_eval_users = gate.w, gate.steps, gate.tol, gate.n_eval_users
    t0 = time.perf_counter()
    O, cfg, R, a_ui, a_iu, feats, params = _oracle_state(w, gate.init)
    sh = w.sh
    opt = O.AdamW(params, lr=cfg.lr)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))
    worst = {"forward": 0.0, "bpr": 0.0, "loss": 0.0, "grad": 0.0, "adamw": 0.0, "param": 0.0, "moment": 0.0, "param_l2": 0.0}
    worst_name = {}

    def upd(kind, name, e):
        if e > worst[kind]:
            worst[kind], worst_name[kind] = e, name
    f = w.fused
    host = lambda t: t.detach().cpu()
    for s in range(steps):
        sn = gate.snap[s]
        # the GPU's own optimiser inputs before this step (parameters, moments), for the AdamW-in-isolation check below
        pre = {nm: host(sn["pre"][nm]) for nm in ORACLE_PARAMS}
        pre_mv = {nm: tuple(host(t) for t in sn["pre_mv"][nm]) if sn["pre_mv"][nm] is not None else None for nm in ORACLE_PARAMS}
        u, p, n = sn["batch"]
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
        loss, parts = O.step_loss(fw, u, p, n, sh.n_items, cfg)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        for nm, t in sn["named"].items():
            upd("forward", "step%d/%s" % (s, nm), rel(host(t), fw[nm].detach()))
        for k in w.keys:
            upd("forward", "step%d/att_u/%s" % (s, k), rel(host(sn["att_u"][k]), fw["att_u"][k].detach()))
            upd("forward", "step%d/att_i/%s" % (s, k), rel(host(sn["att_i"][k]), fw["att_i"][k].detach()))
        got = host(sn["bpr_out"]).double()
        want = torch.tensor([[float(a.detach()), float(b.detach())] for a, b in parts["bpr"]], dtype=torch.float64)
        upd("bpr", "step%d" % s, float((got - want).abs().max() / want.abs().max()))
        upd("loss", "step%d" % s, abs(sn["loss"] - float(loss.detach())) / abs(float(loss.detach())))
        g_gpu = {nm: host(sn["grad"][nm]) for nm in ORACLE_PARAMS}
        p_gpu = {nm: host(sn["post"][nm]) for nm in ORACLE_PARAMS}
        for nm in ORACLE_PARAMS:
            upd("grad", "step%d/%s" % (s, nm), rel(g_gpu[nm], grads[nm]))
        # the optimiser kernel in isolation: the oracle's AdamW fed with the GPU's OWN gradients must land on the GPU's
        # parameters (the end-to-end comparison below also carries Adam's amplification of gradient noise: in the first
        # steps the update is lr * g / (|g| + 1e-8), ill-conditioned where |g| is of the order of the gradient's
        # absolute error)
        iso = {nm: pre[nm].clone() for nm in ORACLE_PARAMS}
        opt_iso = O.AdamW(iso, lr=cfg.lr)
        opt_iso.t = s
        for nm in ORACLE_PARAMS:
            if pre_mv[nm] is not None:
                opt_iso.m[nm], opt_iso.v[nm] = pre_mv[nm][0].clone(), pre_mv[nm][1].clone()
        opt_iso.step({nm: g_gpu[nm] for nm in ORACLE_PARAMS})
        for nm in ORACLE_PARAMS:
            upd("adamw", "step%d/%s" % (s, nm), rel(p_gpu[nm], iso[nm]))
        opt.step(grads)
        for nm in ORACLE_PARAMS:
            upd("param", "step%d/%s" % (s, nm), rel(p_gpu[nm], params[nm].detach()))
            # end-to-end gates that Adam's first steps do not ill-condition: the two moments are linear / quadratic in the
            # gradients, and the L2 distance of a whole tensor does not notice the few entries whose near-zero gradient flips sign
            if sn["post_mv"][nm] is not None:
                mg, vg = (host(t) for t in sn["post_mv"][nm])
                upd("moment", "step%d/m/%s" % (s, nm), rel(mg, opt.m[nm]))
                upd("moment", "step%d/v/%s" % (s, nm), rel(vg, opt.v[nm]))
            a64, b64 = p_gpu[nm].double(), params[nm].detach().double()
            upd("param_l2", "step%d/%s" % (s, nm), float((a64 - b64).norm() / b64.norm()))
        gate.snap[s] = None                                      # (release the step's device clones)
    # evaluation (captured right after the steps): forward with the post-step parameters + scoring + masked top-50
    idx = gate.eval_snap["idx"]
    with torch.no_grad():
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
    e_u, e_i = host(gate.eval_snap["E_u"]), host(gate.eval_snap["E_i"])
    # E_u / E_i of the evaluation = the model after `steps` optimiser steps on each side: the end-to-end check that is not
    # ill-conditioned by Adam's first updates (a parameter entry whose gradient is of the order of the gradient's absolute
    # error moves by lr in either direction; the embeddings the loss and the ranking read do not notice)
    eval_E = max(rel(e_u, fw["E_u"]), rel(e_i, fw["E_i"]))
    upd("forward", "eval/E_u", rel(e_u, fw["E_u"])); upd("forward", "eval/E_i", rel(e_i, fw["E_i"]))
    rng = np.random.default_rng(123)
    users = np.sort(rng.choice(sh.n_users, size=min(n_eval_users, sh.n_users), replace=False))
    K = idx.shape[1]
    S = O.scores_fma_chain(e_u.numpy()[users], e_i.numpy(), order="mfma16x16x4")     # the kernel's scores, bit for bit
    S_or = (fw["E_u"][torch.as_tensor(users)] @ fw["E_i"].t()).numpy()
    Rc = R.tocsr()
    idx_np = idx.cpu().numpy()
    equal = equal_oracle = 0
    test_items = rng.integers(0, sh.n_items, size=sh.n_users)                          # one synthetic held-out item per user
    m_want = np.zeros((4, len(cfg.Ks)))
    # lists from the ORACLE's embeddings (end to end: oracle steps -> oracle forward -> torch matmul) vs the GPU's lists: where they
    # differ, the two items at the first differing rank must be a near-tie under BOTH score sets - their gap, in units in the last place
    # of the score, is bounded by twice the largest difference between the two score matrices on these users (a swap needs
    # s_a >= s_b on one side and s_a <= s_b on the other)
    # one unit in the last place at the scale of a user's scores (its largest |score|): a difference between two scores of one user in
    # these units says how many fp32 roundings of a score-sized number apart they are (a score near zero has a tiny ulp of its own)
    row_ulp = np.spacing(np.maximum(np.abs(S).max(axis=1), np.abs(S_or).max(axis=1)).astype(np.float32)).astype(np.float64)
    ulp = lambda r_: float(row_ulp[r_])
    score_diff_ulps = float((np.abs(S.astype(np.float64) - S_or.astype(np.float64)).max(axis=1) / row_ulp).max())
    gap_ulps, n_mismatch_positions, not_neighbour_swaps = 0.0, 0, 0
    for r, uu in enumerate(users):
        tr = Rc.indices[Rc.indptr[uu]:Rc.indptr[uu + 1]]
        want = O.rank_topk_np(S[r], tr, K)
        got = idx_np[uu][idx_np[uu] >= 0]
        equal += int(got.tolist() == want.tolist())
        want_or = O.rank_topk_np(S_or[r], tr, K)
        same = got.tolist() == want_or.tolist()
        equal_oracle += int(same)
        if not same:
            for pos_ in np.flatnonzero(got[:len(want_or)] != want_or[:len(got)]):
                a_, b_ = int(got[pos_]), int(want_or[pos_])
                n_mismatch_positions += 1
                gap_ulps = max(gap_ulps, abs(float(S[r][a_]) - float(S[r][b_])) / ulp(r), abs(float(S_or[r][a_]) - float(S_or[r][b_])) / ulp(r))
                nb = [int(x) for x in want_or[max(0, pos_ - 2):pos_ + 3]]           # (three near-tied items rotate by up to two ranks)
                not_neighbour_swaps += int(a_ not in nb and pos_ < K - 2)      # (at the last rank the partner may sit just outside the list)
        mm = O.metrics_from_hits([1 if int(i) == int(test_items[uu]) else 0 for i in want], 1, cfg.Ks)
        for j, kname in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            m_want[j] += mm[kname] / len(users)
    # R10 on the device for the same users (llmrec_topk_hits + llmrec_topk_metrics)
    dev = w.device
    q = torch.as_tensor(users, dtype=torch.int64, device=dev)
    t_rp = torch.arange(sh.n_users + 1, dtype=torch.int32, device=dev)
    t_ci = torch.as_tensor(test_items, dtype=torch.int32, device=dev)
    idx_q = idx[q].contiguous()
    hits = w.ops.topk_hits(idx_q, q, t_rp, t_ci)
    m_got = w.ops.topk_metrics(idx_q, hits, q, t_rp, cfg.Ks).mean(dim=0).cpu().numpy()
    metrics_abs = float(np.abs(m_got - m_want).max())
    rep = {"against": "oracle/oracle.py (CPU restatement of the reference, pinned to tests/golden) on the identical device-sampled batches",
           "path": "fused step%s, gemm=%s" % (" + HIP graph replay (step 1 = the capture's eager warm-up)" if w.use_graph else " (eager)", f.gemm),
           "steps_checked": steps, "tolerance_rel": tol,
           "forward_max_rel": worst["forward"], "bpr_max_rel": worst["bpr"], "loss_rel": worst["loss"],
           "grad_max_rel": worst["grad"], "adamw_given_gpu_grads_max_rel": worst["adamw"], "embeddings_after_steps_max_rel": eval_E,
           "param_max_rel": worst["param"], "param_l2_rel_max": worst["param_l2"], "adam_moments_max_rel": worst["moment"],
           "param_note": "gated end to end: embeddings_after_steps_max_rel (E_u / E_i of the evaluation after the optimiser steps on both sides) "
                         "adamw_given_gpu_grads (the timed path's own update - the AdamW inside the weight-gradient reduction launch for the four Linears, "
                         "llmrec_adamw_multi for the tables - against the oracle's AdamW fed with the GPU's own gradients and moments), "
                         "adam_moments_max_rel (exp_avg / exp_avg_sq after the steps, oracle gradients vs GPU gradients: < 1e-4) and param_l2_rel_max "
                         "(||p_gpu - p_oracle|| / ||p_oracle|| per tensor: < 1e-5); param_max_rel "
                         "(oracle gradients -> oracle AdamW vs GPU gradients -> GPU AdamW, per entry) is reported, not gated: in Adam's first steps "
                         "the update lr * g / (|g| + 1e-8) turns an absolute gradient error on a near-zero entry into a full lr step", "worst_tensor": worst_name,
           "topk_lists_checked": int(len(users)), "topk_lists_equal": int(equal),
           "topk_lists_equal_oracle_embeddings": int(equal_oracle), "topk_mismatch_positions": int(n_mismatch_positions),
           "topk_mismatch_max_gap_ulps": gap_ulps, "topk_mismatch_not_neighbour_swaps": int(not_neighbour_swaps),
           "scores_gpu_vs_oracle_max_diff_ulps": score_diff_ulps,
           "topk_note": "topk_lists_equal: the GPU's lists vs the reference ranking rule (score desc, item id asc) applied to the kernel's own bit-exact fp32 "
                        "scores (gated: all equal). topk_lists_equal_oracle_embeddings: vs the lists ranked from the ORACLE's end-to-end embeddings; every "
                        "differing position is measured: the two items' score gap under both score sets, in ulps at the scale of that user's largest |score| "
                        "(topk_mismatch_max_gap_ulps), gated <= max(8, 2 x scores_gpu_vs_oracle_max_diff_ulps) - a swap of two scores that "
                        "sit closer together than the two embedding sets differ - and every such position must be a swap of list neighbours",
           "metrics_max_abs": metrics_abs,
           "when": "GPU side (the steps, the evaluation, device clones of every compared tensor) before the timed region; the oracle's side after it",
           "seconds": time.perf_counter() - t0 + gate.capture_s, "capture_seconds": gate.capture_s}
    stagewise = max(worst[k] for k in ("forward", "bpr", "loss", "grad"))
    e2e_ok = worst["moment"] < tol and worst["param_l2"] < 1e-5
    swaps_ok = gap_ulps <= max(8.0, 2.0 * score_diff_ulps) and not_neighbour_swaps == 0
    rep["ok"] = bool(stagewise < tol and worst["adamw"] < 1e-5 and eval_E < tol and equal == len(users) and metrics_abs < 1e-12 and swaps_ok and e2e_ok)
    gate.eval_snap = gate.init = None
    return rep


def parity_check(w: "NetflixShaped", n_eval_users: int = 256, steps: int = 2, tol: float = 1e-4):
    """Both phases back to back (tests/test_gpu_bench_shapes.py)."""
    return ParityGate(w, n_eval_users, steps, tol).capture().compare()


def cpu_baseline_nf(w: "NetflixShaped", budget_s: float = 20.0):
    """The oracle (CPU restatement of the reference, oracle/oracle.py) timed on this host on a bounded number of
    steps of the SAME workload, batches from the reference's host sampler restated in the oracle
    (load_data.py:157-195 + main.py:216-224). kind = "port" (the reference tree does not travel to the GPU box).
    Thread count: the best of {8, 16, 32, 64} (capped at the host's cores) on one calibration step each (the reference runs with
    torch's default = all cores; BASELINE.md section 2 measured it on 8)."""
    import random as _random
    import numpy as np
    import torch
    O, cfg, R, a_ui, a_iu, feats, params = _oracle_state(w)
    sh = w.sh
    opt = O.AdamW(params, lr=cfg.lr)
    Rc = R.tocsr()
    train_items = {u: Rc.indices[Rc.indptr[u]:Rc.indptr[u + 1]].tolist() for u in range(sh.n_users) if Rc.indptr[u + 1] > Rc.indptr[u]}
    exist = sorted(train_items)
    aug_p, aug_n = w.aug_pos.cpu().numpy(), w.aug_neg.cpu().numpy()
    aug = {u: (int(aug_p[u]), int(aug_n[u])) for u in range(sh.n_users)}
    rd, nprng = _random.Random(2022), np.random.RandomState(2022)

    def one_step():
        users, pos, neg = O.sample_batch(exist, train_items, sh.n_items, sh.n_users, cfg.batch_size, rd=rd, nprng=nprng)
        users, pos, neg = O.augment_batch(users, pos, neg, aug, sh.n_items, cfg.aug_sample_rate, rd=rd)
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
        loss, _ = O.step_loss(fw, users, pos, neg, sh.n_items, cfg)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        opt.step(grads)
    all_cores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cal = {}
    one_step()                                               # warm the allocator / MKL
    for nt in sorted({min(c, all_cores) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        t = time.perf_counter(); one_step(); cal[nt] = time.perf_counter() - t
    best = min(cal, key=cal.get)
    torch.set_num_threads(best)
    steps, t0 = 0, time.perf_counter()
    while True:
        one_step()
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 64:
            break
    dt = time.perf_counter() - t0
    # eval sample: 256 users, full ranking (the reference's per-user python ranking, batch_test.py:83-109)
    with torch.no_grad():
        fw = O.forward(params, feats, a_ui, a_iu, cfg)
    rng = np.random.default_rng(0)
    test_set = {u: [int(rng.integers(0, sh.n_items))] for u in range(256)}
    t1 = time.perf_counter()
    O.evaluate(fw["E_u"].numpy(), fw["E_i"].numpy(), list(range(256)), train_items, test_set, cfg.Ks, batch_size=cfg.batch_size)
    de = time.perf_counter() - t1
    torch.set_num_threads(default_threads)
    return {"value": steps * cfg.batch_size / dt, "unit": "edges/s", "cores": best, "host_cores": all_cores, "kind": "port",
            "sample": "%d full training steps of the same workload (oracle/oracle.py, torch-CPU fp32, host sampler stream); eval on 256 users" % steps,
            "threads_calibration_s_per_step": {str(k): round(v, 3) for k, v in cal.items()},
            "ms_per_step": dt / steps * 1e3, "eval_users_per_s": 256 / de}


def reference_unmodified_record():
    """The UNMODIFIED reference timed on CPU in the build container (oracle/time_reference.py -> profiles/r*_reference_cpu.json): the
    reference tree does not travel to the GPU box, its measurement does. None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    keep = {k: d[k] for k in ("train_s", "eval_s", "edges_per_s", "users_per_s", "n_batch", "batch_size", "n_test_users", "host", "dataset") if k in d}
    keep["file"] = os.path.basename(files[-1])
    keep["how"] = ("oracle/time_reference.py: the reference's own Trainer.train() and its own `Epoch %d [%.1fs + %.1fs]` timers (main.py:200,297,303) on the "
                   "dataset tools/e2e_main.py writes; recorded where /root/reference exists")
    return keep


def end_to_end_main(epochs: int = 6):
    """BASELINE.json's metric as the reference defines it: `python main.py` on the full Netflix-shaped dataset, timed by the drop-in's own
    epoch timers t2 - t1 / t3 - t2 (reference main.py:200,297,303) in its two modes - tools/e2e_main.py, run as a subprocess beside
    this process (it writes the dataset under /tmp first)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="llmrec_e2e_"), "e2e.json")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_main.py"), "--epochs", str(epochs), "--out", out,
                        "--modes", "default,graph_device_sampler"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
    if r.returncode != 0 or not os.path.exists(out):
        return {"error": (r.stderr or r.stdout)[-400:]}
    d = json.load(open(out))
    res = {"command": d["command"], "timers": d["timers"], "dataset": d["dataset"]["shape"], "wall_s": round(time.perf_counter() - t0, 1)}
    for mode in ("default", "graph_device_sampler"):
        m = d.get(mode, {})
        res[mode] = {k: m.get(k) for k in ("train_s", "eval_s", "edges_per_s", "users_per_s", "sample_s", "sample_share_of_train", "epoch0_train_s",
                                           "epoch0_eval_s", "init_s", "n_batch", "n_test_users", "epochs_timed", "final_loss", "final_recall20")}
        res[mode]["python_main_py"] = m.get("python_main_py")
        if "vs_reference" in m:                               # the headline shape against the reference ITSELF (profiles/r05_reference_cpu.json)
            res[mode]["vs_reference"] = m["vs_reference"]
    res["modes"] = {"default": "the reference's host sample stream (utility/load_data.py, same seed -> the reference's batches) + one H2D copy + the fused step "
                               "and the evaluation replayed from HIP graphs",
                    "graph_device_sampler": "LLMREC_DEVICE_SAMPLER=1: the HIP sampler inside the step graph (what `value` above times)"}
    ref = reference_unmodified_record()
    if ref is not None:
        res["reference_unmodified_cpu"] = {k: ref.get(k) for k in ("train_s", "eval_s", "edges_per_s", "users_per_s", "host", "file")}
    return res


def row_sharded_measure(name, scaling, seed, device, rank, world, steps, barrier, exchange="all_reduce", single=False, sparse_forward=True):
    """Time `steps` steps of the row-sharded ID path (all ranks call this; max over ranks; result on every rank).
    single: this process alone runs the WHOLE workload (a communicator of one rank) - the 1-GPU reference of a strong-scaling line."""
    import gc
    import torch
    w = RowSharded(name, scaling, 0, seed, device, rank, world, exchange=exchange, single=single, sparse_forward=sparse_forward)
    for _ in range(2):
        w.step()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / steps * 1e3
    out = {"metric": "bpr_train_edges_per_sec", "value": steps * w.units_per_step / dt, "unit": "edges/s", "ms_per_step": ms, "steps": steps,
           "n_gpus": world, "scaling": scaling, "config": w.config()}
    out.update(w.extras(ms))
    del w
    gc.collect(); torch.cuda.empty_cache()
    return out


def variant_step_time(w: "NetflixShaped", steps: int, env: dict, what: dict):
    """The same workload with one switch of llmrec_amd/fused.py flipped, timed in a FRESH process (python bench.py --workload ... with the switch in
    the environment): a second and third FusedStep inside this process share the HIP runtime's hardware queues with the timed one's graph
    executables, and with GPU_MAX_HW_QUEUES=8 (llmrec_amd/__init__.py) the third one's step measured 2 x slower than the same step alone."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", w.shape_name, "--steps", str(steps), "--warmup", "20", "--seed", "0", "--no-parity",
           "--no-kernel-roofline", "--no-cpu-baseline", "--no-row-sharded", "--no-end-to-end"]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run(cmd, capture_output=True, text=True, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return dict(what, error=(r.stderr or r.stdout)[-300:])
    d = json.loads(lines[-1])
    return dict(what, ms_per_step=d["ms_per_step"], ms_per_step_hip_events=d["ms_per_step_hip_events"], value=d["value"], unit="edges/s", steps=d["steps"],
                how="a fresh process: python bench.py --workload %s --steps %d with %s" % (w.shape_name, steps, " ".join("%s=%s" % kv for kv in env.items())))


def other_workload_step_time(workload: str, steps: int, warmup: int, extra=()):
    """ms_per_step / value of another BASELINE.json configuration on this GPU, from a fresh `python bench.py --workload ...` process (its own
    compact line, the side legs switched off)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup), "--seed", "0", "--no-parity",
           "--no-kernel-roofline", "--no-cpu-baseline", "--no-row-sharded", "--no-end-to-end", "--no-extra-configs"] + list(extra)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-300:]}
    d = json.loads(lines[-1])
    out = {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": "edges/s", "steps": d["steps"], "warmup": d["warmup"],
           "workload": d.get("config", {}).get("workload"), "wall_s": round(time.perf_counter() - t0, 1),
           "how": "a fresh process: " + " ".join(cmd[1:])}
    if isinstance(d.get("eval"), dict):
        out["eval_users_per_s"] = d["eval"].get("value")
    if isinstance(d.get("ingest"), dict):
        out["hbm_peak_gb"] = d["ingest"].get("hbm_peak_gb")
    return out


def exact_f32_step_time(w: "NetflixShaped", steps: int):
    """The same step with the exact fp32 MFMA chain in the 12 GEMM launches (LLMREC_GEMM=f32) instead of the default 3-term bf16 split."""
    return variant_step_time(w, steps, {"LLMREC_GEMM": "f32"}, {"gemm": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32) in projections and weight-gradients"})


def other_order_step_time(w: "NetflixShaped", steps: int):
    """The same step with the OTHER order of the two constant products on the item side (llmrec_amd/fused.py chooses by shape:
    pre-propagated operands (A_ui F_k) W^T iff U <= I, else projection then propagation A_ui (F_k W^T) as the reference writes it,
    Models.py:145-157)."""
    to_preprop = not getattr(w.fused, "preprop", False)
    return variant_step_time(w, steps, {"LLMREC_PREPROPAGATE": "1" if to_preprop else "0"},
                             {"order": ("pre-propagated operands: the step projects A_ui F_k [U x K] (formed once at set-up)" if to_preprop else
                                        "project F_k [I x K], then propagate through A_ui and A_iu every step (Models.py:145-157 as written)"),
                              "chosen_by_shape": "no: the timed step runs the other order (U %s I)" % ("<=" if not to_preprop else ">")})


COMPACT_LIMIT = 6144          # bytes: the driver keeps a bounded tail of stdout (BENCH_r04.json: a 20.5 KB line came back unparsed)


def _r(x, sig=6):
    """Round a float to `sig` significant digits (ints, None, bools, strings pass through; NaN / inf become None: strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    if x == 0.0:
        return 0.0
    return float("%.*g" % (sig, x))


def _pick(d, keys, sig=6):
    return {k: _r(d[k], sig) for k in keys if isinstance(d, dict) and k in d}


def compact_line(line: dict) -> dict:
    """The ONE stdout line: numbers only, <= COMPACT_LIMIT bytes. Everything else - notes, definitions, kernel lists,
    messages, ingest times, log lines - is the detail record (bench_detail.json + stderr). A pure function of the detail dict
    (tests/test_bench_line_cpu.py builds it from a canned one)."""
    out = _pick(line, ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "n_devices_seen", "steps", "warmup", "ms_per_step", "ms_per_step_hip_events",
                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data"), 7)
    cfg = line.get("config", {})
    out["config"] = {k: (_r(v) if not isinstance(v, str) else v) for k, v in cfg.items()
                     if k == "workload" or (not isinstance(v, (str, dict, list)) and v is not None)}
    out["config"] = dict(list(out["config"].items())[:14])

    def roof(r):
        if not isinstance(r, dict):
            return None
        o = {"kernel": str(r.get("kernel", "")).split(" ")[0].split("(")[0].rstrip(",:;")[:48]}
        o.update(_pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_source", "frac_isolated", "frac_rocprof_committed", "traffic",
                           "algorithmic_bytes_per_launch", "in_step_us", "in_step_us_rocprof", "isolated_us", "gather_gbs"), 5))
        if o.get("in_step_us") is None:
            o.pop("in_step_us", None)
        if isinstance(r.get("in_step_us_rocprof"), dict):
            o["in_step_us_rocprof"] = _r(r["in_step_us_rocprof"].get("avg_us"), 5)
        return o
    if "roofline" in line:
        out["roofline"] = roof(line["roofline"])
        if isinstance(line["roofline"].get("second"), dict):
            out["roofline"]["second"] = roof(line["roofline"]["second"])
    sp = line.get("spmm_roofline")
    if isinstance(sp, dict):
        out["spmm_roofline"] = {k: _pick(sp[k], ("ms", "ms_rows_in_id_order", "frac_hbm_algorithmic", "frac_gather_model", "traffic_over_algorithmic", "l2_hit_rate"), 4)
                                for k in ("ui", "iu") if k in sp}
        if "cache_policy" in sp:
            out["spmm_roofline"]["cache_policy"] = {k: v for k, v in sp["cache_policy"].items() if k in ("experiment", "outcome", "file")}
        if isinstance(sp.get("xcd_map"), dict):
            out["spmm_roofline"]["xcd_map"] = {k: v for k, v in sp["xcd_map"].items() if k in ("experiment", "file", "comm")}
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ms_per_step", "eval_users_per_s"), 5)
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:80]
        ru = cb.get("reference_unmodified")
        if isinstance(ru, dict):
            o = _pick(ru, ("edges_per_s", "users_per_s", "train_s", "eval_s"), 5)
            o["cores"] = (ru.get("host") or {}).get("cores")
            out["cpu_baseline"]["reference_unmodified"] = o
    par = line.get("parity")
    if isinstance(par, dict):
        keys = ("ok", "forward_max_rel", "bpr_max_rel", "loss_rel", "grad_max_rel", "adamw_given_gpu_grads_max_rel", "embeddings_after_steps_max_rel",
                "param_l2_rel_max", "adam_moments_max_rel", "topk_lists_checked", "topk_lists_equal", "topk_lists_equal_oracle_embeddings",
                "topk_mismatch_max_gap_ulps", "metrics_max_abs",
                "rows_per_spmm", "spmm_checked", "max_rel_err_vs_fp64", "tolerance_rel")           # (the row-sharded workloads' sampled-row gate)
        out["parity"] = _pick(par, keys, 3)
    if isinstance(line.get("eval"), dict):
        out["eval"] = _pick(line["eval"], ("value", "ms", "n_users", "topk_mode", "topk_tiles", "topk_tiles_redone_exact", "topk_train_rows_as_bitmaps"), 6)
        out["eval"]["unit"] = "users/s"
    for k in ("exact_f32", "reference_order", "pre_propagated_order"):
        if isinstance(line.get(k), dict):
            out[k] = _pick(line[k], ("ms_per_step", "value", "error"), 6)
    for k in ("ml", "cfg5"):
        if isinstance(line.get(k), dict):
            out[k] = _pick(line[k], ("ms_per_step", "value", "eval_users_per_s", "hbm_peak_gb", "steps"), 6)
            if "error" in line[k]:
                out[k]["error"] = str(line[k]["error"])[-120:]
    pe = line.get("propagated_edges_per_sec")
    if isinstance(pe, dict):
        out["propagated_edges_per_sec"] = _pick(pe, ("executed", "reference_equivalent", "executed_per_step", "reference_equivalent_per_step"), 6)
    ee = line.get("end_to_end")
    if isinstance(ee, dict):
        o = {}
        for mode in ("default", "graph_device_sampler"):
            if isinstance(ee.get(mode), dict):
                o[mode] = _pick(ee[mode], ("edges_per_s", "users_per_s", "train_s", "eval_s"), 5)
                if isinstance(ee[mode].get("vs_reference"), dict):
                    o[mode]["vs_reference"] = _pick(ee[mode]["vs_reference"], ("ok", "epochs", "loss_rel", "mf_rel", "emb_rel", "metric_max_abs"), 4)
        if "error" in ee:
            o["error"] = str(ee["error"])[-160:]
        out["end_to_end"] = o
    rs = line.get("row_sharded")
    if isinstance(rs, dict):
        o = {}
        for k in ("weak", "strong"):
            if isinstance(rs.get(k), dict):
                o[k] = _pick(rs[k], ("ms_per_step", "value", "propagated_edges_per_sec"), 6)
                if "error" in rs[k]:
                    o[k]["error"] = str(rs[k]["error"])[:120]
                if isinstance(rs[k].get("row_restricted_forward"), dict):
                    o[k]["row_restricted_forward_ms"] = _r(rs[k]["row_restricted_forward"].get("ms_per_step"))
                if "vs_prev" in rs[k]:
                    o[k]["vs_prev"] = str(rs[k]["vs_prev"])[:100]
        out["row_sharded"] = o
    # row-sharded workloads (--workload cfg4 / cfg5, and N > 1)
    if isinstance(line.get("propagated_edges_per_sec"), (int, float)):
        out["propagated_edges_per_sec"] = _r(line["propagated_edges_per_sec"])
    if isinstance(line.get("ingest"), dict):
        out["ingest"] = _pick(line["ingest"], ("generate_s", "csr_build_and_plans_s", "hbm_peak_gb"), 4)
    if isinstance(line.get("eval_sample"), dict):
        out["eval_sample"] = _pick(line["eval_sample"], ("users", "items", "ms", "users_per_s", "tflops", "frac_mfma_f32"), 5)
    for k in ("propagated_edges_per_step", "spmm_algorithmic_bytes_per_step_per_gpu", "speedup_vs_single_gpu", "peak_memory_gb", "loss"):
        if k in line:
            out[k] = _r(line[k])
    if isinstance(line.get("messages"), dict):
        out["messages"] = _pick(line["messages"], ("exchange", "allreduce_I_x_d_bytes", "exchanged_bytes_last_step", "allreduce_messages"), 5)
    for k in ("single_gpu_reference", "netflix_replicas", "row_restricted_forward"):
        if isinstance(line.get(k), dict):
            out[k] = _pick(line[k], ("ms_per_step", "value", "error"), 6)
    if isinstance(line.get("step_in_graph"), dict):
        out["step_in_graph"] = _pick(line["step_in_graph"], ("entry_point_calls", "projection_us"), 5)
    out["detail"] = line.get("detail_file", "bench_detail.json")
    text = json.dumps(out, allow_nan=False)
    # a last guard: drop optional blocks, least important first, until the line fits
    for k in ("pre_propagated_order", "reference_order", "messages", "step_in_graph", "cfg5", "ml", "row_sharded", "spmm_roofline", "end_to_end"):
        if len(text) <= COMPACT_LIMIT:
            break
        out.pop(k, None)
        text = json.dumps(out, allow_nan=False)
    return out


def write_detail(line: dict):
    """The full record next to the compact line: bench_detail.json at the repo root and under gpurun_out/ (the directory gpurun merges back)."""
    paths = []
    for p in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as f:
                json.dump(line, f, indent=1, default=str)
            paths.append(p)
        except OSError:
            pass
    return paths


def launch_decision(gpus: int, env, n_devices: int):
    """What `python bench.py --gpus N` does before any work, as a pure function of (flag, environment, visible devices):
      ("run", world)      - run in this process as one rank of `world` (world = WORLD_SIZE when a launcher set it, else 1);
      ("spawn", N)        - no launcher in the environment and N > 1: re-exec under torch.distributed.run with N ranks on this node;
      ("refuse", reason)  - the request cannot be honoured (fewer devices than ranks, or --gpus contradicting WORLD_SIZE): exit code 2,
                            never a mislabelled 1-GPU line.
    LLMREC_BENCH_SINGLE_DEVICE=1 (test hook, with LLMREC_DIST_BACKEND=gloo) puts every rank on cuda:0 and waives the device count."""
    single = env.get("LLMREC_BENCH_SINGLE_DEVICE", "0") == "1"
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if gpus not in (1, world):                           # (--gpus left at its default under a launcher: the launcher's world counts)
            return "refuse", "--gpus %d contradicts WORLD_SIZE=%d set by the launcher" % (gpus, world)
        if world > n_devices and not single:
            return "refuse", "WORLD_SIZE=%d ranks but this node shows %d GPU(s)" % (world, n_devices)
        return "run", world
    if gpus <= 1:
        return "run", 1
    if gpus > n_devices and not single:
        return "refuse", "--gpus %d but this node shows %d GPU(s)" % (gpus, n_devices)
    return "spawn", gpus


def spawn_ranks(n: int, argv):
    """Re-exec this script as n ranks on this node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port <free> bench.py <argv>` (what the driver's own N > 1 command line does); returns the launcher's exit code. Rank 0's
    JSON line goes to this process' stdout unchanged."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] --gpus %d without a launcher in the environment: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    import torch
    kind, what = launch_decision(a.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if kind == "refuse":
        print("[bench] refused: %s" % what, file=sys.stderr, flush=True)
        raise SystemExit(2)
    if kind == "spawn":
        raise SystemExit(spawn_ranks(what, sys.argv[1:]))
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
    n_ranks_seen = 1
    if use_pg:                                               # what the communicator itself reports, and how many distinct devices answered
        import torch.distributed as dist
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, "%s/%d" % (os.uname().nodename, torch.cuda.current_device()))
        n_ranks_seen = int(ones.item())
        n_devices_seen = len(set(ids))
    else:
        n_devices_seen = 1
    if n_ranks_seen != world or (n_devices_seen != world and os.environ.get("LLMREC_BENCH_SINGLE_DEVICE", "0") != "1"):
        # the line's n_gpus must be what actually ran: n_gpus == n_ranks_seen == n_devices_seen, or no line at all
        print("[bench] refused: world %d, but the communicator counts %d rank(s) on %d distinct device(s)" % (world, n_ranks_seen, n_devices_seen),
              file=sys.stderr, flush=True)
        raise SystemExit(3)
    workload = a.workload
    auto = workload == "auto"
    if auto:
        # N = 1: BASELINE.json configs[1] (Netflix shape). N > 1: north_star's multi-GPU split - configs[3], the user-row-sharded ID
        # path with one I x d exchange per layer and direction, STRONG scaling (the same 10 M x 1 M x 200 M graph over the ranks);
        # the batch-sharded Netflix replicas ride along as `netflix_replicas`
        workload = "nf" if world == 1 else "cfg4"
        if world > 1:
            a.synth_scaling = "strong"
    dflt = {"nf": (200, 20), "ml": (200, 20), "cfg5": (5, 1)}.get(workload, (20, 3))
    a.steps = dflt[0] if a.steps is None else a.steps
    a.warmup = dflt[1] if a.warmup is None else a.warmup

    if workload in ("nf", "ml"):
        from llmrec_amd import dist as ldist
        w = NetflixShaped(workload, a.seed, device, rank, world, ldist.Comm() if use_pg else None)
        step, units = w.step, w.units_per_step * world      # global batch = world x batch_size
    else:
        name = "cfg4" if workload in ("synth", "cfg4") else workload
        if name not in SYNTH_CONFIGS:
            raise SystemExit("unknown --workload %s" % workload)
        w = RowSharded(name, a.synth_scaling, a.synth_blocks, a.seed, device, rank, world, n_chunks=a.synth_chunks, exchange=a.synth_exchange,
                       sparse_backward=not a.synth_dense_backward, sparse_forward=a.synth_restricted_forward)
        step, units = w.step, w.units_per_step              # global batch per step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # Order of the single-GPU Netflix / MovieLens line: [parity gate, GPU side: 2 steps + 1 evaluation, device clones]
    # -> [evaluation leg + isolated-kernel legs, enqueued without a host synchronisation] -> W warm-up steps -> barrier + synchronize -> K
    # timed steps -> everything else, the CPU phases (the oracle's side of the parity gate, cpu_baseline) LAST. Round 5 ran the gate's ~10 s
    # of host-only oracle work right before the warm-up: with W = 5 the timed region began inside the power-management transient of the
    # first ~12 ms of load and read 6 % above the steady state. Nothing here adds warm-up steps: the legs ahead of the warm-up are
    # measurements the line reports (`eval`, `roofline.frac_isolated`).
    gate, parity, eval_read, pre_kernels = None, None, None, None
    single_nf = workload in ("nf", "ml") and world == 1 and os.environ.get("LLMREC_FORCE_DP", "0") != "1"
    if single_nf and not a.no_parity:
        gate = ParityGate(w).capture()
    if single_nf:
        if w.step_id == 0:
            w.step()                                         # (no gate: the legs below read the gradient buffers of a training step)
        w.eval_once()                                        # captures the evaluation graph (synchronises)
        eval_read = event_time_deferred(w.eval_once, 5, 0)
        if not a.no_kernel_roofline:
            pre_kernels = w.kernel_timings_enqueue()

    finish = getattr(getattr(w, "fused", None), "flush", lambda: None)   # batch-sharded replicas defer the last AdamW
    if hasattr(w, "run_steps"):
        w.run_steps(a.warmup)                                # (the multi-step graph is replayed here first, not inside the timed region)
    else:
        for _ in range(a.warmup):
            step()
    finish()
    barrier(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                             # HIP events on the stream the step graphs are launched on
    if hasattr(w, "run_steps"):
        w.run_steps(a.steps)                                 # exactly a.steps steps (graphs of several steps + single-step graphs)
    else:
        for _ in range(a.steps):
            step()
    finish()                                                 # inside the timed region: every step's update is applied
    ev1.record()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    dt_events = ev0.elapsed_time(ev1) / 1e3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    line = {"metric": "bpr_train_edges_per_sec", "value": a.steps * units / dt, "unit": "edges/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "ms_per_step_hip_events": dt_events / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": w.config(),
            "n_ranks_seen": n_ranks_seen, "n_devices_seen": n_devices_seen}
    if workload in ("nf", "ml"):
        gemm = getattr(w.fused, "gemm", "f32")
        line["arithmetic"] = ("fp32 storage and accumulation everywhere; the 8 projections and 4 weight-gradients multiply exact 3-term bf16 "
                              "splits of both fp32 operands on the bf16 MFMA (24 significant bits, 2e-6 vs fp64: fp32-class); "
                              "LLMREC_GEMM=f32 selects the exact fp32 MFMA chain, timed below as exact_f32") if gemm == "bf16x3" else \
                             "exact fp32 (fp32 MFMA fma chains in the projections / weight-gradients)"

    if workload in ("nf", "ml"):
        if eval_read is not None:
            te = eval_read() / 1e3                           # HIP events around 5 evaluations, enqueued ahead of the warm-up steps
        else:
            w.eval_once(); torch.cuda.synchronize(); barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                w.eval_once()
            torch.cuda.synchronize(); barrier()
            te = (time.perf_counter() - t1) / 5              # every rank ranks its user block; the barrier makes it the slowest rank's time
    if rank == 0 and workload in ("nf", "ml"):
        tk = {}
        try:                                                  # the top-K launch's mode and how many user tiles its verification sent to the exact sweep
            from llmrec_amd import _lib as _l, ops as _o
            evs = list(getattr(w.fused, "_eval_graphs", {}).values())
            mode = _o.topk_mode(None, w.sh.n_items, w.args.embed_size, 50)
            tk = {"topk_mode": "bf16_sweep_exact_verify" if mode == 1 else "exact_fp32_sweep"}
            if mode == 1 and evs and evs[0][5] is not None:
                off = _l.query("llmrec_score_topk_stats_offset", evs[0][3].numel(), w.sh.n_items)
                tk["topk_tiles"] = (evs[0][3].numel() + 15) // 16
                tk["topk_tiles_redone_exact"] = int(evs[0][5][off + 4:off + 8].view(torch.int32).item())
                tk["topk_train_rows_as_bitmaps"] = int(evs[0][5][off + 8:off + 12].view(torch.int32).item())
        except Exception as e:                                # pragma: no cover
            tk = {"topk_stats_error": str(e)[:100]}
        line["eval"] = {"metric": "full_rank_eval_users_per_sec", "value": w.sh.n_users / te, "ms": te * 1e3, **tk,
                        "n_users": w.sh.n_users, "users_per_rank": (w.sh.n_users + world - 1) // world,
                        "includes": "no-grad full-graph forward + fp32 MFMA scoring + masked top-50"}
        if world == 1 and not a.no_kernel_roofline and getattr(w.fused, "gemm", "f32") == "bf16x3" and os.environ.get("LLMREC_FORCE_DP", "0") != "1":
            line["exact_f32"] = exact_f32_step_time(w, min(a.steps, 100))
            line["reference_order" if getattr(w.fused, "preprop", False) else "pre_propagated_order"] = other_order_step_time(w, min(a.steps, 100))
        if not a.no_kernel_roofline:
            ks = w.kernel_rooflines(pre_kernels)
            line["kernels"] = ks
            try:                                              # (a measurement aid must never cost the line: on any failure the fractions
                ig = w.in_graph_durations() if world == 1 else None   #  fall back to the committed summary / the isolated launches)
            except Exception as e:                            # pragma: no cover
                print("[bench] in-graph timestamps unavailable: %s: %s" % (type(e).__name__, str(e)[:200]), file=sys.stderr, flush=True)
                ig = None
            if ig is not None:
                line["step_in_graph"] = dict(ig, entry_point_calls=getattr(w.fused, "entry_point_calls_per_step", None))
            # dominant kernel = the largest share of the step's GPU time: the weight-gradient launches (rocprofv3 round 1:
            # 25 % of the step) ahead of the single grouped-projection launch; both are reported, the dominant one first
            gemms = [k for k in ks if "algorithmic_bytes_per_launch" in k]

            def live_us(k):
                """(duration, source) of one launch as THIS run measured it (the printed fraction never comes from a committed file).
                Projection: the device-timestamp bracket inside the replayed step graph - an upper bound (it includes the dispatch gaps of the two
                stamp launches), tight at the head of the graph (136.3 us against 132.1 us in the rocprofv3 summary of round 5). Weight gradient:
                HIP events around 20 isolated launches on the launch's stream - its in-graph bracket sits behind two cross-queue joins and reads
                35 % long (dropped from the line; kept in the detail record as in_step_bracket_us)."""
                n_ = k.get("launches", 1)
                us_ = None if ig is None else ig.get("projection_us" if "linear_fwd" in k["kernel"] else None)
                if us_:
                    return us_, "in_graph_timestamps"
                return k["ms"] / n_ * 1e3, "isolated_events"
            dom = max(gemms, key=lambda k: live_us(k)[0])
            other = min(gemms, key=lambda k: live_us(k)[0])

            def roof(k):
                hbm = k.get("bound") == "hbm"
                traffic, tsrc = pmc_traffic_bytes(k["pmc"]) if workload == "nf" else (None, None)
                n = k.get("launches", 1)
                iso = k["frac_hbm"] if hbm else k["frac_mfma_f32"]
                bracket = None if ig is None else ig.get("projection_us" if "linear_fwd" in k["kernel"] else "wgrad_us")
                use, src = live_us(k)
                per_us = (lambda us_: (k["algorithmic_bytes_per_launch"] / us_ / 1e3) if hbm else (k["algorithmic_flop_per_launch"] / us_ / 1e6))
                ach = per_us(use)
                peak = HBM_PEAK_GBS if hbm else MFMA_F32_PEAK_TFLOPS
                rp = rocprof_avg_us(k["pmc"]) if workload == "nf" else None     # (the committed summary is of the Netflix-shaped command)
                return {"kernel": k["kernel"], "bound": k.get("bound", "mfma"),
                        "achieved": ach, "peak": peak,
                        "unit": "GB/s" if hbm else "TFLOP/s", "frac": ach / peak, "frac_isolated": iso,
                        "in_step_us": use if src == "in_graph_timestamps" else None, "isolated_us": k["ms"] / n * 1e3,
                        "frac_source": src, "in_step_us_used": use, "in_step_bracket_us": bracket,
                        # the cross-check, never the source of `frac`: the same kernel's average duration inside the replayed graph in the
                        # rocprofv3 --kernel-trace --stats summary committed under profiles/ (of an earlier run of this command)
                        "frac_rocprof_committed": (per_us(rp["avg_us"]) / peak) if rp else None, "in_step_us_rocprof": rp,
                        "traffic": None if traffic is None else traffic / n, "traffic_source": tsrc,
                        "launches_per_step": n, "ms_per_launch": k["ms"] / n, "ms_per_step": k["ms"],
                        "algorithmic_flop_per_launch": k["algorithmic_flop_per_launch"],
                        "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"],
                        "timing": k.get("timing", "HIP events around the launch on its stream, in isolation"),
                        "hbm_gbs": k["gbs"], "frac_hbm": k["frac_hbm"]}
            line["roofline"] = roof(dom)
            if other is not dom:
                line["roofline"]["second"] = roof(other)
            line["roofline"]["step_kernel_time_by_class"] = kernel_time_shares()
            line["spmm_roofline"] = spmm_roofline_large(device, a.seed)
        # the edge traversals behind `value` - the reference's forward runs 20 SpMMs and its backward 20 transposed ones per step
        # (Models.py:153-180); the fused step forms the same products as fewer, wider launches (7 d operands, pre-propagated A_ui F_k)
        # `executed` = the traversals the step's SpMM launches actually perform (sum over the launches the graph was built
        # from of nnz x d / 64: FusedStep.spmm_edge_units - the pre-propagated A_ui F_k products are formed once at set-up and are NOT counted),
        # `reference_equivalent` = the reference's 40 d = 64 SpMMs per step
        nnz = int(w.rows.size)
        ex_units = float(getattr(w.fused, "spmm_edge_units", 0.0)) or None
        line["propagated_edges_per_sec"] = {"executed": None if ex_units is None else ex_units * a.steps * world / dt, "executed_per_step": ex_units,
                                            "reference_equivalent": 40.0 * nnz * a.steps * world / dt, "reference_equivalent_per_step": 40 * nnz,
                                            "definition": "executed: steps x sum over the step's SpMM launches of nnz x d / 64, / time; reference_equivalent: "
                                                          "steps x (20 forward + 20 transposed SpMMs of the reference's step) x nnz / time"}
        if workload == "nf" and world == 1 and not a.no_end_to_end:
            line["end_to_end"] = end_to_end_main()
    if workload in ("nf", "ml") and not a.no_row_sharded:
        # north_star's split (configs[3]): the row-sharded ID path on the cfg-4-shaped graph, next to the replica line above -
        # weak (2 of the 16 user blocks per GPU: cfg 4 exactly at 8 GPUs) and strong (all 16 blocks = cfg 4, split over the ranks)
        rs = {}
        for scaling, steps_rs in (("weak", 20), ("strong", 8)):
            try:
                # `value` = the step that forms every forward product for every row (VERDICT r03 next #4b); the last layer restricted to
                # the rows the step reads is the labelled variant beside it (same losses, gradients and parameters: tests/test_dist_cpu.py)
                rs[scaling] = row_sharded_measure("cfg4", scaling, a.seed, device, rank, world, steps_rs, barrier, sparse_forward=False)
                rs[scaling]["forward"] = "dense: every product of Models.py:169-186 for every row (this is `value`)"
                if scaling == "strong":
                    rs[scaling]["vs_prev"] = "r03 51.0 ms was the row-restricted forward; `value` is the dense forward since r04"
                if scaling == "strong":
                    rr = row_sharded_measure("cfg4", scaling, a.seed, device, rank, world, steps_rs, barrier, sparse_forward=True)
                    rs[scaling]["row_restricted_forward"] = {"ms_per_step": rr["ms_per_step"], "value": rr["value"], "messages": rr.get("messages"),
                                                             "what": "the last layer's two forward products formed only in the rows the step reads "
                                                                     "(llmrec_amd/dist_fused.py forward(needed=...)): a labelled variant, not the headline"}
            except torch.OutOfMemoryError as e:               # pragma: no cover
                rs[scaling] = {"error": "out of memory: %s" % str(e)[:120]}
            except Exception as e:                            # pragma: no cover - the replica line above must still be printed
                rs[scaling] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if rank == 0:
            line["row_sharded"] = rs
    if rank == 0 and workload == "nf" and world == 1 and auto and not a.no_extra_configs:
        # the other single-GPU configurations of BASELINE.json in the same line, each in a fresh process
        line["ml"] = other_workload_step_time("ml", 100, 20)
        line["cfg5"] = other_workload_step_time("cfg5", 3, 1, ["--synth-scaling", "strong"])
    if rank == 0 and single_nf:
        # the CPU phases, LAST: the oracle's side of the parity gate (its GPU side ran before the timed region) and the CPU baseline
        if gate is not None:
            parity = line["parity"] = gate.compare()
            print("[bench] parity gate: %s" % json.dumps({k: v for k, v in parity.items() if k != "worst_tensor"}), file=sys.stderr, flush=True)
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_nf(w)
            ref = reference_unmodified_record()
            if ref is not None:                               # the real thing, measured where it can run (build container, 8 cores)
                line["cpu_baseline"]["reference_unmodified"] = ref
    if workload not in ("nf", "ml"):
        line["scaling"] = w.scaling
        ex = w.extras(dt / a.steps * 1e3)                    # (collective inside: every rank calls it)
        if rank == 0:
            line.update(ex)
            if world == 1 and not a.no_parity:
                line["parity"] = w.sampled_row_parity()
            if not a.no_kernel_roofline:
                sp = w.spmm_times_ms()                       # rank 0's local products (no collective inside)
                line["spmm_in_situ"] = sp
                path = {k: v for k, v in sp.items() if "not_on_the_path" not in k}
                worst = min(path, key=lambda k: path[k]["frac_hbm_algorithmic"])
                line["roofline"] = {"kernel": "spmm_kernel, the step's dominant kernel (4 L products per step): the slowest direction in situ (%s) on rank 0's shard" % worst,
                                    "bound": "hbm", "achieved": path[worst]["algorithmic_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": path[worst]["frac_hbm_algorithmic"], "traffic": None,
                                    "gather_gbs": path[worst]["gather_gbs"],
                                    "note": "fraction of the ALGORITHMIC-bytes roofline (every X row read once); the kernel moves gather_gbs of row gathers, "
                                            "the measured ceiling of that access pattern (7.3 - 7.8 TB/s, profiles/r02_gatherbench_v2.txt; DESIGN.md 4)"}
                if world == 1:
                    line["eval_sample"] = w.eval_sample()
    if workload not in ("nf", "ml") and world > 1 and auto:
        import gc
        del w, step
        gc.collect(); torch.cuda.empty_cache()
        # (0) the labelled variant of the same N-rank step: the last layer's forward restricted to the rows the step reads
        try:
            rr = row_sharded_measure("cfg4", "strong", a.seed, device, rank, world, 8, barrier, exchange=a.synth_exchange, sparse_forward=True)
            restricted = {"ms_per_step": rr["ms_per_step"], "value": rr["value"], "messages": rr.get("messages"),
                          "what": "forward(needed=...): a labelled variant, not `value` (which forms every forward product for every row)"}
        except Exception as e:                                # pragma: no cover
            restricted = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # (a) the same workload on ONE GPU, run by rank 0 alone inside this job (the others wait): the reference the strong-scaling
        #     speed-up is quoted against, measured on the same box in the same run
        ref = None
        if not a.no_single_gpu_reference:
            if rank == 0:
                try:
                    ref = row_sharded_measure("cfg4", "strong", a.seed, device, 0, 1, 8, lambda: None, exchange=a.synth_exchange, single=True, sparse_forward=False)
                except Exception as e:                        # pragma: no cover
                    ref = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            barrier()
        # (b) BASELINE.json's Netflix workload as batch-sharded replicas (llmrec_amd/dp.py): the global batch N x 1024 over the ranks
        rep = None
        try:
            from llmrec_amd import dist as ldist
            wn = NetflixShaped("nf", a.seed, device, rank, world, ldist.Comm())
            for _ in range(20):
                wn.step()
            wn.fused.flush(); barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                wn.step()
            wn.fused.flush(); torch.cuda.synchronize(); barrier()
            dtn = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            import torch.distributed as dist
            dist.all_reduce(dtn, op=dist.ReduceOp.MAX)
            rep = {"metric": "bpr_train_edges_per_sec", "value": 200 * wn.units_per_step * world / float(dtn.item()), "unit": "edges/s",
                   "ms_per_step": float(dtn.item()) / 200 * 1e3, "steps": 200, "scaling": "weak", "config": wn.config()}
            del wn
        except Exception as e:                                # pragma: no cover - the main line must still be printed
            rep = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if rank == 0:
            line["scaling_note"] = ("N > 1 measures BASELINE.json configs[3] (north_star's multi-GPU split: user-row-sharded ID path, one I x d exchange per layer "
                                    "and direction, the same 10 M x 1 M x 200 M graph over the ranks = STRONG scaling); the N = 1 line is configs[1] (Netflix shape). "
                                    "Scaling is value / single_gpu_reference.value (the same workload on one GPU, measured by rank 0 inside this run), "
                                    "not value / the N = 1 line's value; the Netflix workload as batch-sharded replicas is netflix_replicas")
            line["netflix_replicas"] = rep
            line["row_restricted_forward"] = restricted
            if ref is not None:
                line["single_gpu_reference"] = ref
                if "value" in ref:
                    line["speedup_vs_single_gpu"] = line["value"] / ref["value"]
    if rank == 0:
        write_detail(line)
        print("[bench] detail: %s" % json.dumps(line, default=str), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(line), allow_nan=False), flush=True)   # the ONE stdout line (<= COMPACT_LIMIT bytes)
    if use_pg:
        import torch.distributed as dist
        dist.barrier()                                       # rank 0's per-kernel measurements are done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
