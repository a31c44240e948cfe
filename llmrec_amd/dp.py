"""Batch-sharded replicas of the fused step: the multi-GPU form of cfg 2 / cfg 3.

At Netflix / MovieLens scale the whole problem (100 k edges, 30 k x 64 embeddings, < 0.5 GB of side
features) is a rounding error of one MI355X's 288 GB, and a step is 1.2 ms. Row-sharding the users
(llmrec_amd.dist, the layout for cfg 4 / 5) would all-reduce every per-layer item message - two
I x 7d operands and ~7 I x d ones per step, ~93 MB - to save SpMM time that is not the bottleneck.
Here every rank keeps the full graph and all tables, takes 1/world of the GLOBAL batch, and the
step exchanges exactly two things:

  1. one all-gather of LLMREC_BPR_GATHER_FLOATS(8, B) floats per rank (36 KB): the log-sigmoids of
     the 8 BPR problems, the local squared-norm sums and the valid-sample count - so that the prune
     threshold of reference main.py:158-165 is taken over the GLOBAL batch and the reciprocal
     regulariser sees the global norms (identical values on every rank);
  2. one all-reduce (sum) of the flat gradient bucket (every .grad is a view of it; 9.4 MB at
     Netflix shape) whose tail carries the loss scalars for logging.

The result equals the single-GPU step of the reference on the concatenated batch (batch size
world x B, same --batch_size flag in the regulariser): tests/test_gpu_step.py runs two replicas
through a loop-back exchange on one GPU against FusedStep on the concatenated batch.
Replicas stay bit-identical: both collectives deliver identical values to every rank, and every
rank applies the same AdamW update.

Between the exchanges the step is HIP graphs (sampler + forward + scores | selection + backward | AdamW); the
collectives run between them on the same stream (RCCL, or nothing when world == 1). The AdamW graph of a step is
merged into the first graph of the next one (two graph boundaries per step instead of three; flush() applies the
last one).
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .fused import FusedStep, _capture_without_gc, _call, _p


def gather_floats(n_prob: int, b_max: int) -> int:
    """LLMREC_BPR_GATHER_FLOATS(P, B) of include/llmrec_hip.h."""
    return n_prob * b_max + 4 * n_prob + 1


def bucket_gradients(model, extra: int = 16):
    """Make every trainable parameter's .grad a view of one flat fp32 buffer (+ `extra` floats of
    tail for the loss scalars). Returns (bucket, n_grad_floats)."""
    params = [p for p in model.parameters() if p.requires_grad and p is not model.batch_norm.weight and p is not model.batch_norm.bias]
    pad = lambda k: (k + 63) // 64 * 64                        # 256-byte aligned views (vector loads in AdamW / weight-grad)
    n = sum(pad(p.numel()) for p in params)
    bucket = torch.zeros(n + extra, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        p.grad = bucket[off:off + p.numel()].view_as(p)
        off += pad(p.numel())
    return bucket, n


class DataParallelStep(FusedStep):
    """FusedStep over a batch that is sharded across `world` replicas (this rank holds `b_max`
    slots of it). comm: llmrec_amd.dist.Comm (.rank, .world, .dist = torch.distributed); None = single
    replica or a loop-back harness that moves the two exchange buffers itself (rank=, world=)."""

    def __init__(self, model, graph, hp, rates, optimizer, b_max: int, comm=None, rank: int = None, world: int = None):
        self.bucket, self.n_grad = bucket_gradients(model)
        super().__init__(model, graph, hp, rates, optimizer, b_max)
        self.comm = comm
        self.rank = comm.rank if comm is not None else (rank or 0)
        self.world = comm.world if comm is not None else (world or 1)
        if self.world * b_max > 4 * _lib.CONST["LLMREC_BPR_MAX_B"]:
            raise RuntimeError("DataParallelStep: global batch capacity %d exceeds %d" % (self.world * b_max, 4 * _lib.CONST["LLMREC_BPR_MAX_B"]))
        dev = self.E_u.device
        self.gsz = gather_floats(self.n_prob, b_max)
        self.g_local = torch.zeros(self.gsz, dtype=torch.float32, device=dev)
        self.g_all = torch.zeros(self.world * self.gsz, dtype=torch.float32, device=dev)
        self.tail = self.bucket[self.n_grad:]
        self.graphs = None
        import os
        # LLMREC_DP_FORCE_COLLECTIVES=1: issue the RCCL calls even in a world of one rank (plumbing check on a 1-GPU box)
        self.force = os.environ.get("LLMREC_DP_FORCE_COLLECTIVES", "0") == "1" and comm is not None and comm.dist is not None
        self.lazy_update = os.environ.get("LLMREC_DP_LAZY", "1") == "1"   # defer AdamW into the next step's first graph (see step())
        self._pending_update = False

    # -- the three compute phases -----------------------------------------------------------------
    WGRAD_ROWS = False          # (the replicas keep the dense weight-gradient launch: their gradients are summed over ranks afterwards)
    INLINE_ADAMW = False        # the gradients are all-reduced over the replicas first (exchange_grads), then phase_c updates
    FOLD = False                # (the folded launches of the single-GPU step assume the in-step AdamW)

    def _bpr_phase(self, phase, users, pos, neg, n_valid):
        hp = self.hp
        _call("llmrec_bpr_multi_fwd_sharded_f32", self.n_prob, self._problems(), self.d, _p(users), _p(pos), _p(neg), users.numel(),
              _p(n_valid), float(1 - hp.prune_loss_drop_rate), float(hp.decay), float(hp.batch_size), phase, _p(self.g_local),
              _p(self.g_all), self.world, self.gsz, self.rank, _p(self.out), _p(self.saved))

    def phase_a(self, users, pos, neg, n_valid=None, sampler=None):
        """[sampler +] forward + BPR scores + this rank's gather block."""
        if users.numel() != self.b_max:
            raise RuntimeError("DataParallelStep: every rank passes exactly b_max = %d slots (n_valid marks the used ones)" % self.b_max)
        if sampler is not None:
            sampler()
        self.build_scatter_plan(users, pos, neg, n_valid)                # the deterministic scatter of phase_b's backward (fused.py)
        self._train_forward()                                            # (also starts the feature regulariser's value on s3)
        self._bpr_phase(1, users, pos, neg, n_valid)
        self._join(self.s3)

    def phase_b(self, users, pos, neg, n_valid=None):
        """selection against the gathered global batch + backward into the gradient bucket."""
        self._bpr_phase(2, users, pos, neg, n_valid)
        P = self.n_prob
        self._fork(self.s3)
        with self._on(self.s3):                                          # loss scalars for the bucket's tail, off the critical path
            # tail[:P] = this rank's shares of the mf values; tail[P] = (emb + feat_reg) / world (identical on all ranks)
            self._assemble_loss(1, self.tail, 1.0 / self.world)
        self._backward(self._problems(), users, pos, neg, n_valid, replicated_scale=1.0 / self.world)
        self._join(self.s3)

    def phase_c(self):
        """loss scalars from the reduced tail + AdamW."""
        self._assemble_loss(2, self.tail)
        self.opt.step(advanced=True)                             # the counter was advanced during the forward

    # -- the two exchanges ------------------------------------------------------------------------
    def exchange_scores(self):
        if self.comm is not None and (self.world > 1 or self.force):
            self.comm.all_gather_into(self.g_all, self.g_local, force=self.force)
        else:                                                  # single replica, or a loop-back harness that fills the other blocks
            self.g_all[self.rank * self.gsz:(self.rank + 1) * self.gsz].copy_(self.g_local)

    def exchange_grads(self):
        if self.comm is not None and (self.world > 1 or self.force):
            self.comm.all_reduce_(self.bucket, force=self.force)

    def loss_backward(self, users, pos, neg, n_valid=None):
        raise RuntimeError("DataParallelStep: use step_eager()/step() (the loss needs the score exchange)")

    def step_eager(self, users, pos, neg, n_valid=None):
        self.phase_a(users, pos, neg, n_valid)
        self.exchange_scores()
        self.phase_b(users, pos, neg, n_valid)
        self.exchange_grads()
        self.phase_c()
        return self.scal[1], self.scal[2], self.scal[3]

    # -- HIP graphs -------------------------------------------------------------------------------
    def capture(self, warm_users=None, warm_pos=None, warm_neg=None, warm_n_valid=None, batcher=None):
        """Three graphs between the two exchanges; with `batcher` the sampler opens the first one."""
        st = self._make_static()
        self.batcher = batcher
        if batcher is not None:
            if batcher.capacity != self.b_max:
                raise RuntimeError("DataParallelStep.capture: batcher capacity %d != b_max %d" % (batcher.capacity, self.b_max))
        else:
            self._load(warm_users, warm_pos, warm_neg, warm_n_valid)
        args = (st["users"], st["pos"], st["neg"], st["n_valid"])

        def first():
            self.phase_a(*args, sampler=(lambda: batcher.fill(*args)) if batcher is not None else None)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                             # warm-up: one full eager step (also warms the collectives)
            first(); self.exchange_scores(); self.phase_b(*args); self.exchange_grads(); self.phase_c()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        def first_with_update():                               # the previous step's AdamW opens the next step's first graph
            self.phase_c()
            first()
        graphs = []
        for fn in (first, lambda: self.phase_b(*args), self.phase_c, first_with_update):
            g = torch.cuda.CUDAGraph()                         # capturing records, it does not execute: one step ran (the warm-up)
            with _capture_without_gc(g):                       # (no Python garbage collection inside a stream capture: llmrec_amd/fused.py)
                fn()
            graphs.append(g)
        self.graphs = graphs
        self.graph_exec = graphs[0]
        self._pending_update = False

    def step(self, users=None, pos=None, neg=None, n_valid=None):
        """One step from the captured graphs. The AdamW update of a step is deferred into the first graph of the NEXT
        step (two graph boundaries per step instead of three; a boundary costs ~50 us of launch latency): parameters
        and the loss scalars therefore lag by one step until flush() - evaluation and any read of the parameters go
        through flush() first (eval_topk does)."""
        if self.graphs is None:
            return self.step_eager(users, pos, neg, n_valid)
        if users is not None:
            if getattr(self, "batcher", None) is not None:
                raise RuntimeError("DataParallelStep.step: this graph samples its own batch; call step() without arguments")
            self._load(users, pos, neg, n_valid)
        elif getattr(self, "batcher", None) is None:
            raise RuntimeError("DataParallelStep.step: a batch is needed (the graphs were captured without a sampler)")
        ga, gb, gc, ga_upd = self.graphs
        if self.lazy_update:
            (ga_upd if self._pending_update else ga).replay()
            self.exchange_scores()
            gb.replay()
            self.exchange_grads()
            self._pending_update = True
        else:
            ga.replay()
            self.exchange_scores()
            gb.replay()
            self.exchange_grads()
            gc.replay()
        return self.scal[1], self.scal[2], self.scal[3]

    def run_steps(self, n: int):
        """n steps (the exchanges sit between this step's graphs: no multi-step graph here)."""
        out = self.scal[1], self.scal[2], self.scal[3]
        for _ in range(n):
            out = self.step()
        return out

    def flush(self):
        """Apply the deferred AdamW update of the last step (and its loss scalars)."""
        if self.graphs is not None and getattr(self, "_pending_update", False):
            self.graphs[2].replay()
            self._pending_update = False

    def eval_topk(self, query_users, train, K, use_graph=False):
        self.flush()
        return super().eval_topk(query_users, train, K, use_graph=use_graph)
