"""User-sharded propagation over several GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). Users are partitioned
into contiguous blocks; rank r owns
  * the edges of its users (two pattern-only CSRs: by local user, by item),
  * its slice of the user-side tensors (user_id_embedding rows, their AdamW state),
  * a REPLICA of every item-side tensor (I x d: 256 MB at cfg 4, 2.56 GB at cfg 5).
Per propagation layer there is exactly one exchange, an all-reduce (sum, fp32) of an I x d array:
  forward   I^{l+1} = sum_r A_iu[:, blk_r] U^{l+1}[blk_r]        (ShardedIU.forward)
  backward  dI^{l}  = sum_r A_ui[blk_r, :]^T dU^{l+1}[blk_r]     (ShardedUI.backward)
U^{l+1}[blk] = A_ui[blk, :] I^l and its mirror in backward are local. The BPR batch is sharded by
user owner; the prune selection ranks against the all-gathered log-sigmoids of the whole batch
(B floats), so the result equals the single-GPU step.

The local kernels come from a ``backend`` object (default: the HIP library through
llmrec_amd.ops). tests/test_dist_cpu.py injects a torch-CPU backend to check the partitioning and
the placement of the collectives with gloo, world_size 2, against the single-process oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn


class Comm:
    """Thin view of torch.distributed (identity when the world has one rank)."""

    def __init__(self, single: bool = False):
        """single: a communicator of this rank alone even when a process group exists (a 1-GPU reference run inside an N-GPU job)."""
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized() and not single) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        import os
        # LLMREC_FORCE_COLLECTIVES=1: issue every collective even in a world of one rank (plumbing check of the RCCL calls
        # on a 1-GPU box: `torch.distributed.run --nproc-per-node 1 bench.py ...`)
        self.force = self.dist is not None and os.environ.get("LLMREC_FORCE_COLLECTIVES", "0") == "1"

    def _host_staged(self) -> bool:
        """gloo (the CPU-test backend) moves device tensors through host copies and lacks some fused collectives."""
        return self.dist is not None and self.dist.get_backend() == "gloo"

    def all_reduce_(self, t: torch.Tensor, force: bool = False) -> torch.Tensor:
        if self.dist and (self.world > 1 or force or self.force):
            if self._host_staged() and t.is_cuda:
                h = t.cpu()
                self.dist.all_reduce(h)
                t.copy_(h)
            else:
                self.dist.all_reduce(t)
        return t

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor, force: bool = False) -> torch.Tensor:
        """out[world * n] <- the ranks' t[n] in rank order (preallocated, graph-step friendly)."""
        if self.dist and (self.world > 1 or force or self.force):
            if self._host_staged() and t.is_cuda:
                parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
                self.dist.all_gather(parts, t.cpu())
                out.copy_(torch.cat(parts).reshape(out.shape))
            else:
                self.dist.all_gather_into_tensor(out, t)
        else:
            out.copy_(t)
        return out

    def all_gather_cat(self, t: torch.Tensor) -> torch.Tensor:
        if not (self.dist and (self.world > 1 or self.force)):
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous())
        return out


def user_block(n_users: int, rank: int, world: int):
    """Contiguous user partition: [u0, u1) of rank."""
    per = (n_users + world - 1) // world
    return min(rank * per, n_users), min((rank + 1) * per, n_users)


class HipBackend:
    """Local kernels = the C-ABI HIP library."""

    def __init__(self):
        from . import ops
        self.ops = ops
        self._saved = None

    def bpr_local_m(self, saved, B):
        """The local log-sigmoids pass 1 left in the scratch part of `saved`."""
        return saved[B + 4: 2 * B + 4]

    def pattern_csr(self, rows, cols, n_rows, n_cols):
        rp, ci, _ = self.ops.csr_from_coo(rows, cols, None, n_rows, n_cols)
        return self.ops.Csr(n_rows, n_cols, rp, ci, None, None, None, {})

    def with_scales(self, csr, row_scale, col_scale):
        o = self.ops
        return o.Csr(csr.n_rows, csr.n_cols, csr.rowptr, csr.colidx, None, row_scale, col_scale, csr.plans)

    def degrees(self, csr) -> torch.Tensor:
        return (csr.rowptr[1:] - csr.rowptr[:-1]).to(torch.float32)

    def spmm(self, csr, X, out=None, epilogue=None):
        """Y = epilogue(A X); epilogue: None or {"op": "none" | "softmax" | "softmax_bwd", "alpha", "Z", "S", "post_scale",
        "x_row_mask", "x_mask_active", "y_row_flag", "z_row_flag", "y_row_gate", "y_row_needed"} (llmrec_spmm_epilogue_t's row sparsity)."""
        o = self.ops
        epi = None
        if epilogue is not None:
            op = {"none": o.EPI_NONE, "softmax": o.EPI_SOFTMAX, "softmax_bwd": o.EPI_SOFTMAX_BWD}[epilogue.get("op", "none")]
            epi = o.spmm_epilogue(op, epilogue.get("alpha", 0.0), epilogue.get("Z"), epilogue.get("S"), epilogue.get("post_scale"),
                                  epilogue.get("x_row_mask"), epilogue.get("x_mask_active", 0), epilogue.get("y_row_flag"), epilogue.get("z_row_flag"),
                                  epilogue.get("y_row_gate"), epilogue.get("y_row_needed"))
        return o.spmm_raw(csr, X, out=out, epilogue=epi)

    def spmm_listed(self, csr, X, rows, out):
        """out[r] = (A X)[r] for the listed distinct rows only (ops.spmm_listed); other rows of out keep their contents."""
        return self.ops.spmm_listed(csr, X, rows, out)

    # -- "these rows of A X" with the list and its length on the device (round 5: the restricted forward without host read-backs) --
    def sort_unique_ids(self, ids, out_list, out_n):
        self.ops.sort_unique_ids(ids, out_list, out_n)

    def spmm_rows_compact(self, csr, X, row_list, n_list, out):
        """out[j] = (A X)[row_list[j]] for j < n_list[0], zero rows behind (one workspace per output block, kept)."""
        key = ("_rows_compact_ws", tuple(out.shape))
        ws = csr.plans.get(key) if hasattr(csr, "plans") and isinstance(csr.plans, dict) else None
        ws = self.ops.spmm_rows_compact(csr, X, row_list, n_list, out, ws)
        if hasattr(csr, "plans") and isinstance(csr.plans, dict):
            csr.plans[key] = ws

    def scatter_set_rows(self, row_list, n_list, src, dst):
        self.ops.scatter_set_rows(row_list, n_list, src, dst)

    def mark_rows(self, ids, value: int, flags):
        """flags[ids] = value (ids < 0 skipped): llmrec_mark_rows_u8."""
        o = self.ops
        o._lib.call("llmrec_mark_rows_u8", ids.numel(), o._p(ids), int(value), o._p(flags), o._stream())

    def softmax_bwd_listed_into(self, ids, alpha, Y, dY, post_scale, out):
        """out[ids] = post_scale[ids] . softmax_bwd(Y[ids], alpha dY[ids]); other rows of out untouched (llmrec_softmax_rows_bwd_listed_f32)."""
        o = self.ops
        o._lib.call("llmrec_softmax_rows_bwd_listed_f32", ids.numel(), o._p(ids), Y.shape[1], float(alpha), o._p(Y), o._ld(Y), o._p(dY), o._ld(dY),
                    o._p(post_scale), o._p(out), o._ld(out), o._stream())

    def mark_neighbours(self, ids, csr, value: int, flags):
        """flags[c] = value for the columns c of csr's rows ids (llmrec_mark_neighbours_u8)."""
        o = self.ops
        o._lib.call("llmrec_mark_neighbours_u8", ids.numel(), o._p(ids), o._p(csr.rowptr), o._p(csr.colidx), int(value), o._p(flags), o._stream())

    def row_chunk(self, csr, r0, r1):
        """The operand restricted to rows [r0, r1) (views; its own row plan)."""
        o = self.ops
        sl = lambda t: None if t is None else t[r0:r1]
        return o.Csr(r1 - r0, csr.n_cols, csr.rowptr[r0:r1 + 1], csr.colidx, csr.val, sl(csr.row_scale), csr.col_scale, {})

    def softmax_rows_into(self, Z, out):
        o = self.ops
        o._lib.call("llmrec_softmax_rows_fwd_f32", Z.shape[0], Z.shape[1], o._p(Z), o._ld(Z), o._p(out), o._ld(out), o._stream())

    def softmax_bwd_into(self, Y, dY, out):
        o = self.ops
        o._lib.call("llmrec_softmax_rows_bwd_f32", Y.shape[0], Y.shape[1], o._p(Y), o._ld(Y), o._p(dY), o._ld(dY), o._p(out), o._ld(out), o._stream())

    def scale_rows_into(self, s, X, out):
        o = self.ops
        o._lib.call("llmrec_scale_rows_f32", X.shape[0], X.shape[1], o._p(s), o._p(X), o._ld(X), o._p(out), o._ld(out), o._stream())

    def axpy_into(self, alpha, X, out):
        o = self.ops
        o._lib.call("llmrec_axpy_f32", X.shape[0], X.shape[1], float(alpha), None, o._p(X), o._ld(X), o._p(out), o._ld(out), 0, o._stream())

    def layer_mean_into(self, terms, out):
        o = self.ops
        mp, ml = o._ptr_table(list(terms))
        npt, nl = o._ptr_table([])
        r = (o._c.c_float * 1)(0.0)
        o._lib.call("llmrec_fuse_fwd_f32", out.shape[0], out.shape[1], 1.0 / len(terms), len(terms), mp, ml, 0, npt, nl, r,
                    o._p(out), o._ld(out), o._stream())

    def gather_mean_into(self, terms, idx, out):
        """out[b] = mean_t terms[t][idx[b]] (llmrec_gather_mean_f32)."""
        o = self.ops
        tp, tl = o._ptr_table(list(terms))
        o._lib.call("llmrec_gather_mean_f32", idx.numel(), o._p(idx), out.shape[1], 1.0 / len(terms), len(terms), tp, tl, o._p(out), o._ld(out), o._stream())

    def optimizer_step(self, opt, grad_scales):
        """AdamW over opt's parameters; grad_scales {param: s}: that parameter's gradient is s * param.grad."""
        opt.grad_scale = dict(grad_scales)
        opt.step()

    def optimizer_advance(self, opt):
        """The step counter / bias corrections of this step (once per step, before any optimizer_step_params / optimizer_step_rows)."""
        opt.advance()

    def optimizer_step_params(self, opt, params, grad_scales):
        """AdamW over a subset of opt's parameters (counter advanced already)."""
        opt.grad_scale = dict(grad_scales)
        opt.step_params(params)

    def optimizer_step_rows(self, opt, param, row0, row1, grad_rows):
        """AdamW over rows [row0, row1) of `param` with the gradient rows given in a separate [row1 - row0, d] buffer (the piece of a
        reduce-scattered gradient this rank owns): llmrec_adamw_multi_f32 on the row ranges of parameter and moments."""
        o = self.ops
        if row1 <= row0:
            return
        m, v = opt.moments(param)
        d = param.shape[1]
        arr = (o.AdamwTensor * 1)()
        off = row0 * d * 4
        arr[0].p, arr[0].g, arr[0].m, arr[0].v = param.data_ptr() + off, grad_rows.data_ptr(), m.data_ptr() + off, v.data_ptr() + off
        arr[0].n, arr[0].g_scale = (row1 - row0) * d, 1.0
        o._lib.call("llmrec_adamw_multi_f32", 1, arr, o._p(opt.dev_state), opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.wd, o._stream())

    def zero_(self, tensors):
        o = self.ops
        arr = (o.ZeroTensor * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i].p, arr[i].n = t.data_ptr(), t.numel()
        o._lib.call("llmrec_zero_multi_f32", len(tensors), arr, o._stream())

    def zero_rows(self, ids, dst):
        """dst[ids] = 0 (ids < 0 skipped): row-wise clean-up of a scatter target (llmrec_zero_rows_f32)."""
        o = self.ops
        o._lib.call("llmrec_zero_rows_f32", ids.numel(), o._p(ids), dst.shape[1], o._p(dst), o._ld(dst), o._stream())

    def bpr_bwd_rows(self, Eu, Ei, u, p, n, decay, bsz, saved, grads2, rows3):
        o = self.ops
        o._lib.call("llmrec_bpr_prune_bwd_rows_f32", o._p(Eu), o._ld(Eu), o._p(Ei), o._ld(Ei), Eu.shape[1], o._p(u), o._p(p), o._p(n),
                    u.numel(), None, float(decay), float(bsz), o._p(saved), o._p(grads2), o._p(rows3), o._stream())

    def scatter_rows(self, ids, rows, dst, alpha):
        """dst[ids[j]] += alpha * rows[j], duplicates in ascending j (deterministic)."""
        o = self.ops
        n = ids.numel()
        need = o._lib.query("llmrec_scatter_rows_workspace_bytes", n)
        ws = getattr(self, "_scatter_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._scatter_ws = torch.empty(need, dtype=torch.uint8, device=dst.device)
        o._lib.call("llmrec_scatter_rows_f32", n, o._p(ids), o._p(rows), o._ld(rows), rows.shape[1], float(alpha), o._p(dst), o._ld(dst),
                    o._p(ws), ws.numel(), o._stream())

    def softmax_rows(self, Z):
        return self.ops.softmax_rows(Z)

    def layer_mean(self, terms):
        return self.ops.fuse(list(terms), [], [])

    def linear(self, X, W, b):
        return self.ops.linear(X, W, b)

    def fuse(self, mean_terms, norm_terms, rates):
        return self.ops.fuse(list(mean_terms), list(norm_terms), list(rates))

    def sumsq(self, coef, Xs):
        return self.ops.sumsq(coef, list(Xs))[0]

    def score_topk(self, Eu, Ei, q, train, K):
        return self.ops.score_topk(Eu, Ei, q, train, K)

    def topk_hits(self, idx, q, rowptr, colidx):
        return self.ops.topk_hits(idx, q, rowptr, colidx)

    def topk_metric_sums(self, idx, hits, q, rowptr, Ks):
        """[4 * len(Ks)] float64 device sums over the listed users (precision, recall, ndcg, hit_ratio blocks)."""
        return self.ops.topk_metrics(idx, hits, q, rowptr, Ks).sum(0).reshape(-1)

    def bpr_fwd(self, Eu, Ei, u, p, n, remember, decay, bsz, global_m, global_B, offset, scores_only):
        o = self.ops
        Eu, Ei = o._rowmajor(Eu), o._rowmajor(Ei)
        B = u.numel()
        out = torch.empty(2, dtype=torch.float32, device=Eu.device)
        saved = self._saved if (not scores_only and self._saved is not None and self._saved.numel() == o.bpr_saved_floats(B)) \
            else torch.empty(o.bpr_saved_floats(B), dtype=torch.float32, device=Eu.device)
        self._saved = saved if scores_only else None        # pass 2 reuses pass 1's per-sample scratch
        o._lib.call("llmrec_bpr_prune_fwd_sharded_f32", o._p(Eu), o._ld(Eu), o._p(Ei), o._ld(Ei), Eu.shape[1], o._p(u), o._p(p), o._p(n),
                    B, float(remember), float(decay), float(bsz), o._p(global_m), int(global_B), int(offset), 1 if scores_only else 0,
                    o._p(out), o._p(saved), o._stream())
        return out, saved

    def bpr_bwd(self, Eu, Ei, u, p, n, decay, bsz, saved, grads2):
        o = self.ops
        Eu, Ei = o._rowmajor(Eu), o._rowmajor(Ei)
        dEu, dEi = torch.zeros_like(Eu), torch.zeros_like(Ei)
        plan = o.bpr_scatter_plan(u, p, n)
        o._lib.call("llmrec_bpr_prune_bwd_f32", o._p(Eu), o._ld(Eu), o._p(Ei), o._ld(Ei), Eu.shape[1], o._p(u), o._p(p), o._p(n),
                    u.numel(), None, float(decay), float(bsz), o._p(saved), o._p(grads2.contiguous()), o._p(dEu), o._ld(dEu),
                    o._p(dEi), o._ld(dEi), o._p(plan), o._stream())
        return dEu, dEi

    def sample(self, seed, step, exist_users, n_items, by_user, B):
        return self.ops.sample_bpr(seed, step, exist_users, n_items, by_user, B)

    def optimizer(self, params, lr):
        return self.ops.FusedAdamW(params, lr=lr)


@dataclass
class ShardedGraph:
    """This rank's slice of the bipartite graph, ready for the four SpMM directions."""
    n_users_local: int
    n_items: int
    u0: int
    ui_fwd: object      # A_ui[blk, :]        rows = local users, row_scale = s_u
    ui_bwd: object      # A_ui[blk, :]^T      rows = items, col_scale = s_u        (partial -> all-reduce)
    iu_fwd: object      # A_iu[:, blk]        rows = items, row_scale = s_i(global) (partial -> all-reduce)
    iu_bwd: object      # A_iu[:, blk]^T      rows = local users, col_scale = s_i
    by_user: object
    s_i: torch.Tensor

    @staticmethod
    def build(local_users: torch.Tensor, items: torch.Tensor, n_users_local: int, n_items: int, u0: int,
              comm: Comm, backend) -> "ShardedGraph":
        """local_users: user ids RELATIVE to this rank's block."""
        by_user = backend.pattern_csr(local_users, items, n_users_local, n_items)
        by_item = backend.pattern_csr(items, local_users, n_items, n_users_local)
        deg_u = backend.degrees(by_user)
        deg_i = comm.all_reduce_(backend.degrees(by_item).clone())           # global item degrees
        inv = lambda deg: torch.where(deg > 0, (1.0 / torch.sqrt(deg.double())).float(), torch.zeros_like(deg))
        s_u, s_i = inv(deg_u), inv(deg_i)                                     # reference main.py:115-117
        return ShardedGraph(n_users_local, n_items, u0,
                            backend.with_scales(by_user, s_u, None), backend.with_scales(by_item, None, s_u),
                            backend.with_scales(by_item, s_i, None), backend.with_scales(by_user, None, s_i),
                            by_user, s_i)


def _fn_ui(graph: ShardedGraph, comm: Comm, backend):
    class ShardedUI(torch.autograd.Function):
        @staticmethod
        def forward(ctx, Xi):
            return backend.spmm(graph.ui_fwd, Xi)

        @staticmethod
        def backward(ctx, dYu):
            return comm.all_reduce_(backend.spmm(graph.ui_bwd, dYu))
    return ShardedUI.apply


def _fn_iu(graph: ShardedGraph, comm: Comm, backend):
    class ShardedIU(torch.autograd.Function):
        @staticmethod
        def forward(ctx, Xu):
            return comm.all_reduce_(backend.spmm(graph.iu_fwd, Xu))

        @staticmethod
        def backward(ctx, dYi):
            return backend.spmm(graph.iu_bwd, dYi)
    return ShardedIU.apply


def _fn_replicated(comm: Comm):
    class ReplicatedGrad(torch.autograd.Function):
        """Identity on a replicated tensor whose consumers differ per rank: the true gradient is
        the sum of the per-rank gradients."""
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            return comm.all_reduce_(g.contiguous().clone())
    return ReplicatedGrad.apply


def _fn_bpr(comm: Comm, backend, remember: float, decay: float, bsz: float):
    class ShardedBpr(torch.autograd.Function):
        """[mf, emb] of the GLOBAL batch from this rank's share of the samples."""
        @staticmethod
        def forward(ctx, Eu, Ei, u, p, n):
            B = u.numel()
            _, s1 = backend.bpr_fwd(Eu, Ei, u, p, n, remember, decay, bsz, None, 0, 0, True)
            global_m = comm.all_gather_cat(backend.bpr_local_m(s1, B).contiguous())
            out, saved = backend.bpr_fwd(Eu, Ei, u, p, n, remember, decay, bsz, global_m, global_m.numel(), comm.rank * B, False)
            norms = comm.all_reduce_(saved[B:B + 3].clone())
            saved[B:B + 3] = norms
            mf = comm.all_reduce_(out[:1].clone())
            reg = (1.0 / (2.0 * norms + 1e-8)).sum()
            emb = (decay * (reg / bsz)).reshape(1)
            ctx.save_for_backward(Eu, Ei, u, p, n, saved)
            return torch.cat([mf, emb])

        @staticmethod
        def backward(ctx, g):
            Eu, Ei, u, p, n, saved = ctx.saved_tensors
            dEu, dEi = backend.bpr_bwd(Eu, Ei, u, p, n, decay, bsz, saved, g)
            return dEu, dEi, None, None, None
    return ShardedBpr.apply


class ShardedIDModel(nn.Module):
    """The ID-embedding block of the reference (Models.py:169-186) over a sharded graph:
    user rows local, item rows replicated. E_u[blk], E_i = mean over the L+1 layer outputs, row
    softmax on the last layer, items of layer l+1 from the NEW users."""

    def __init__(self, graph: ShardedGraph, comm: Comm, backend, d: int, n_layers: int, n_users_global: int, seed: int):
        super().__init__()
        self.graph, self.comm, self.backend, self.n_layers = graph, comm, backend, n_layers
        dev = graph.s_i.device
        g = torch.Generator(device="cpu"); g.manual_seed(seed)
        # xavier_uniform over the GLOBAL table shapes (reference Models.py:39-42); the item table is
        # drawn identically on every rank, the user table per rank
        bi = math.sqrt(6.0 / (graph.n_items + d))
        self.item_id_embedding = nn.Parameter(((torch.rand(graph.n_items, d, generator=g) * 2 - 1) * bi).to(dev))
        bu = math.sqrt(6.0 / (n_users_global + d))
        gu = torch.Generator(device="cpu"); gu.manual_seed(seed * 7919 + 1 + comm.rank)
        self.user_id_embedding = nn.Parameter(((torch.rand(graph.n_users_local, d, generator=gu) * 2 - 1) * bu).to(dev))
        self.ui = _fn_ui(graph, comm, backend)
        self.iu = _fn_iu(graph, comm, backend)
        self.replicated = _fn_replicated(comm)

    def forward(self):
        u, i = self.user_id_embedding, self.item_id_embedding
        us, is_ = [u], [i]
        for layer in range(self.n_layers):
            last = layer == self.n_layers - 1
            u = self.ui(i)
            if last:
                u = self.backend.softmax_rows(u)
            i = self.iu(u)
            if last:
                i = self.backend.softmax_rows(i)
            us.append(u); is_.append(i)
        e_u = self.backend.layer_mean(us)
        e_i = self.replicated(self.backend.layer_mean(is_))
        return e_u, e_i


class ShardedMMModel(nn.Module):
    """The full multi-modal model (reference Models.py:127-199) over a sharded graph.

    Layout: user-side tensors hold this rank's user rows only (user_id_embedding, the LLM user
    profile features, every *_u stream); item-side tensors and the four Linear layers are
    replicated. Collectives: one all-reduce per iu-direction SpMM forward and per ui-direction SpMM
    backward (ShardedIU / ShardedUI), one all-reduce of the gradient of each replicated tensor that
    feeds a rank-dependent loss (ReplicatedGrad), and of user_trans' weight gradient (its input rows
    are sharded). Item-feature projections are computed on every rank (replicated compute)."""

    def __init__(self, graph: ShardedGraph, comm: Comm, backend, d: int, n_layers: int, n_users_global: int,
                 item_feats: dict, user_feats_local: torch.Tensor, keys, rates, seed: int):
        super().__init__()
        self.graph, self.comm, self.backend, self.n_layers, self.keys = graph, comm, backend, n_layers, list(keys)
        self.c_m, self.c_u, self.c_a = rates
        self.item_feats, self.user_feats = item_feats, user_feats_local
        dev = graph.s_i.device
        torch.manual_seed(seed)                                # identical replicated parameters on every rank
        self.image_trans = nn.Linear(item_feats["image"].shape[1], d)
        self.text_trans = nn.Linear(item_feats["text"].shape[1], d)
        self.user_trans = nn.Linear(user_feats_local.shape[1], d)
        self.item_trans = nn.Linear(item_feats["attr/" + self.keys[0]].shape[1], d)
        for lin in (self.image_trans, self.text_trans, self.user_trans, self.item_trans):
            nn.init.xavier_uniform_(lin.weight)
        self.item_id_embedding = nn.Parameter(torch.empty(graph.n_items, d))
        nn.init.xavier_uniform_(self.item_id_embedding)
        gu = torch.Generator(); gu.manual_seed(seed * 7919 + 1 + comm.rank)
        bu = math.sqrt(6.0 / (n_users_global + d))
        self.user_id_embedding = nn.Parameter((torch.rand(graph.n_users_local, d, generator=gu) * 2 - 1) * bu)
        self.to(dev)
        self.ui = _fn_ui(graph, comm, backend)
        self.iu = _fn_iu(graph, comm, backend)
        self.replicated = _fn_replicated(comm)

    def forward(self):
        be, rep = self.backend, self.replicated
        p_img = be.linear(self.item_feats["image"], self.image_trans.weight, self.image_trans.bias)
        p_txt = be.linear(self.item_feats["text"], self.text_trans.weight, self.text_trans.bias)
        p_att = {k: be.linear(self.item_feats["attr/" + k], self.item_trans.weight, self.item_trans.bias) for k in self.keys}
        # user_trans is replicated but sees only this rank's rows: its gradient is the sum over ranks
        p_usr = be.linear(self.user_feats, rep(self.user_trans.weight), rep(self.user_trans.bias))
        img_u = self.ui(p_img); img_i = self.iu(img_u)
        txt_u = self.ui(p_txt); txt_i = self.iu(txt_u)
        att_u, att_i = {}, {}
        for k in self.keys:
            att_u[k] = self.ui(p_att[k]); att_i[k] = self.iu(att_u[k])
        prof_i = self.iu(p_usr); prof_u = self.ui(prof_i)
        u, i = self.user_id_embedding, self.item_id_embedding
        us, is_ = [u], [i]
        for layer in range(self.n_layers):
            last = layer == self.n_layers - 1
            u = self.ui(i)
            if last:
                u = be.softmax_rows(u)
            i = self.iu(u)
            if last:
                i = be.softmax_rows(i)
            us.append(u); is_.append(i)
        rates = [self.c_m, self.c_m, self.c_u] + [self.c_a] * len(self.keys)
        e_u = be.fuse(us, [img_u, txt_u, prof_u] + [att_u[k] for k in self.keys], rates)
        e_i = be.fuse(is_, [img_i, txt_i, prof_i] + [att_i[k] for k in self.keys], rates)
        # replicated tensors that feed rank-dependent losses: true gradient = sum over ranks
        return {"E_u": e_u, "E_i": rep(e_i), "img_u": img_u, "txt_u": txt_u, "img_i": rep(img_i), "txt_i": rep(txt_i),
                "prof_u": prof_u, "att_i": {k: rep(att_i[k]) for k in self.keys}}


class ShardedMMTrainer:
    """Step of the full model on a sharded batch (reference main.py:228-278): the 8 BPR + prune losses
    over the GLOBAL batch, the feature regulariser (user rows local, item rows replicated and
    weighted 1/world so their all-reduced gradient counts once), backward, AdamW."""

    def __init__(self, model: ShardedMMModel, hp, batch_local: int, n_items: int):
        self.model, self.comm, self.backend, self.hp = model, model.comm, model.backend, hp
        self.n_items = n_items
        self.opt = self.backend.optimizer(list(model.parameters()), hp_lr(hp))
        bsz_flag = float(hp.batch_size)
        self.bpr = _fn_bpr(self.comm, self.backend, 1 - hp.prune_loss_drop_rate, hp.decay, bsz_flag)

    def step(self, users_local, pos, neg):
        hp, be, m = self.hp, self.backend, self.model
        fw = m()
        main = self.bpr(fw["E_u"], fw["E_i"], users_local, pos, neg)
        img = self.bpr(fw["img_u"], fw["img_i"], users_local, pos, neg)
        txt = self.bpr(fw["txt_u"], fw["txt_i"], users_local, pos, neg)
        aug = 0
        for k in m.keys:
            aug = aug + self.bpr(fw["prof_u"], fw["att_i"][k], users_local, pos, neg)[0]
        coef = hp.feat_reg_decay * 0.5 / self.n_items
        reg_local = be.sumsq(coef, [fw["img_u"], fw["txt_u"]]) + be.sumsq(coef / self.comm.world, [fw["img_i"], fw["txt_i"]])
        loss = main[0] + main[1] + reg_local + hp.aug_mf_rate * aug + hp.mm_mf_rate * (img[0] + txt[0])
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        reg_global = self.comm.all_reduce_(reg_local.detach().clone().reshape(1))[0]
        return (loss.detach() - reg_local.detach() + reg_global), main.detach()


def hp_lr(hp):
    return getattr(hp, "lr", 1e-4)


@torch.no_grad()
def sharded_evaluate(model, comm: Comm, backend, graph: ShardedGraph, test_rowptr, test_colidx, n_test_users_global: int, Ks):
    """Full-rank evaluation, embarrassingly parallel by user (reference utility/batch_test.py:112-169):
    each rank ranks its own users against the replicated item table and the 12 metric sums are
    all-reduced. test_rowptr/test_colidx: CSR of the held-out items of this rank's users."""
    fw = model()
    deg = (test_rowptr[1:] - test_rowptr[:-1])
    q = torch.nonzero(deg > 0).reshape(-1).to(torch.int64)
    kmax = max(Ks)
    sums = torch.zeros(4 * len(Ks), dtype=torch.float64, device=graph.s_i.device)
    if q.numel():
        idx, _ = backend.score_topk(fw["E_u"], fw["E_i"], q, graph.by_user, kmax)
        hits = backend.topk_hits(idx, q, test_rowptr, test_colidx)
        sums += backend.topk_metric_sums(idx, hits, q, test_rowptr, Ks)      # metrics on the device: 12 doubles leave it, not n x K hits
    comm.all_reduce_(sums)
    out = (sums / n_test_users_global).cpu().numpy().reshape(4, len(Ks))
    return {"precision": out[0], "recall": out[1], "ndcg": out[2], "hit_ratio": out[3], "auc": 0.0}


class ShardedTrainer:
    """One synchronous training step of the sharded ID model: sample B_local users from the local
    block on the device, sharded BPR + prune over the global batch, backward, AdamW (user rows
    local; the item table's gradient is already complete on every rank, so its update is
    replicated without a further collective)."""

    def __init__(self, model: ShardedIDModel, lr: float, batch_local: int, drop_rate: float, decay: float, seed: int):
        self.model, self.comm, self.backend = model, model.comm, model.backend
        self.B = batch_local
        self.opt = self.backend.optimizer(list(model.parameters()), lr)
        bsz_flag = float(batch_local * self.comm.world)
        self.bpr = _fn_bpr(self.comm, self.backend, 1 - drop_rate, decay, bsz_flag)
        g = model.graph
        deg = self.backend.degrees(g.by_user)
        self.exist = torch.nonzero(deg > 0).reshape(-1).to(torch.int64)
        self.seed = seed * 1000003 + self.comm.rank
        self.step_id = 0

    def sample(self):
        g = self.model.graph
        return self.backend.sample(self.seed, self.step_id, self.exist, g.n_items, g.by_user, self.B)

    def step(self, triples=None):
        u, p, n = triples if triples is not None else self.sample()
        self.step_id += 1
        e_u, e_i = self.model()
        out = self.bpr(e_u, e_i, u, p, n)
        loss = out[0] + out[1]
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach(), out.detach()


class ShardedBench:
    """bench.py --workload synth: cfg-4-shaped synthetic graph, `users_per_gpu` users and
    `edges_per_gpu` edges per rank (weak scaling), items replicated; at 8 GPUs the defaults give
    10 M users x 1 M items x 200 M edges, d = 64, 2 layers, batch 1024 per GPU."""

    def __init__(self, users_per_gpu, n_items, edges_per_gpu, seed, device, rank, world, d=64, n_layers=2, batch_local=1024):
        from . import synth
        self.comm = Comm()
        self.backend = HipBackend()
        self.device, self.world = device, world
        rows, cols = synth.bipartite_edges_device(users_per_gpu, n_items, edges_per_gpu, seed * 131 + rank, device)
        self.nnz_local = int(rows.numel())
        self.graph = ShardedGraph.build(rows, cols, users_per_gpu, n_items, rank * users_per_gpu, self.comm, self.backend)
        del rows, cols
        self.model = ShardedIDModel(self.graph, self.comm, self.backend, d, n_layers, users_per_gpu * world, seed)
        self.trainer = ShardedTrainer(self.model, 1e-4, batch_local, 0.71, 1e-5, seed)
        self.units_per_step = batch_local * world
        self.cfg = {"workload": "synthetic_user_sharded_id_path_cfg4_shape", "users_per_gpu": users_per_gpu, "n_items": n_items,
                    "edges_per_gpu": self.nnz_local, "embed_size": d, "prop_layers": n_layers, "batch_per_gpu": batch_local,
                    "global_batch": batch_local * world, "prune_loss_drop_rate": 0.71,
                    "parallelism": "user-sharded x%d, item table replicated, 1 all-reduce(I x d) per layer fwd + bwd" % world}

    def step(self):
        return self.trainer.step()

    def config(self):
        return self.cfg

    def extras(self):
        nnz = torch.tensor([self.nnz_local], dtype=torch.float64, device=self.device)
        self.comm.all_reduce_(nnz)
        L = self.model.n_layers
        return {"propagated_edges_per_step": float(nnz.item()) * 4 * L,
                "allreduce_bytes_per_step": 4.0 * self.graph.n_items * self.model.item_id_embedding.shape[1] * (2 * L + 1)}
