"""Row-sharded FUSED step of the ID-embedding path (BASELINE.json configs 4 / 5).

Same partition as llmrec_amd/dist.py - users (rows of A_ui, columns of A_iu) sharded over the ranks, every item-side
tensor replicated - but the step is written out by hand like llmrec_amd/fused.py (no autograd graph, preallocated
buffers, SpMM epilogues for the softmax / softmax-backward / "+ mean term" of reference Models.py:169-186), and the
exchanges are organised for xGMI:

  * per layer and direction ONE I x d message (forward: I^{l+1} = sum_r A_iu[:, blk_r] U^{l+1}[blk_r]; backward:
    dI^l = sum_r A_ui[blk_r, :]^T dU^{l+1}[blk_r]). The SpMM that produces it runs in item-row CHUNKS; each chunk's
    all-reduce is issued (async, on the communicator's stream) as soon as the chunk is computed, so the reduction of
    chunk c overlaps the SpMM of chunk c + 1 - the per-layer message is then bound by max(compute, link time) instead
    of their sum (RCCL picks reduce-scatter + all-gather inside each all-reduce; chunks are sized >= 32 MB so the
    point-to-point xGMI links stay bandwidth- rather than latency-bound);
  * the BPR gradient of the replicated fused item table has at most 2 B non-zero rows per rank: the ranks
    ALL-GATHER those rows (ids + 2 B x d floats, 0.5 MB at B = 1024, d = 64) and every rank scatter-adds the same
    list in the same order (llmrec_scatter_rows_f32: sorted, deterministic) - instead of the dense I x d all-reduce
    of llmrec_amd/dist.py's per-op autograd path (256 MB at cfg 4) - so replicas stay bit-identical;
  * prune threshold over the GLOBAL batch: all-gather of B floats (llmrec_bpr_prune_fwd_sharded_f32, two passes),
    norms / mf share: one all-reduce of 4 floats.
With all-reduces the item table's gradient is complete on every rank, so its AdamW update is replicated; with "rs_ag" the update is
sharded (owner_updates_items): the last backward message is reduce-scattered, each owner updates its rows, updated rows are all-gathered.

``exchange="rs_ag"`` replaces each chunk's all-reduce by the DIRECT form of SURVEY.md 8(e) for a fully connected xGMI
mesh: a reduce-scatter (every rank sends 1/world of the chunk to each peer over its own link and reduces the piece it
owns), the row-local epilogue (the last layer's softmax) on the OWNED piece only, and an all-gather of the finished
pieces - 2 (world - 1) / world of the chunk per link and direction instead of a ring's per-link bound, and the softmax
is computed once per row instead of once per rank. Replicas stay bit-identical (every row is reduced at exactly one
rank and broadcast). No multi-GPU box has been available to time either form; ``all_reduce`` stays the default.

Scatter targets (dE_u, dE_i) are kept ALL-ZERO between steps: the step scatters B gradient rows into them and clears
exactly those rows after their last reader (llmrec_zero_rows_f32) - no dense memset of the [U, d] table per step.

The local kernels come from a ``backend`` (llmrec_amd.dist.HipBackend = the C-ABI HIP library; tests inject the
torch-CPU stand-in to run this file under gloo with two ranks against the single-process oracle).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import _lib
from .dist import Comm, ShardedGraph


class _Done:
    def wait(self):
        return True


class _ShardedExchange:
    """reduce-scatter -> (row-local op on the owned piece) -> all-gather of one item-row chunk; see the module docstring.
    start() queues the reduce-scatter behind the kernel that produced the chunk; finish_local() orders the current stream
    after it, runs the epilogue on the owned rows and queues the all-gather; wait() orders the current stream after that.
    gather = (dst_view [rows of the chunk, d], own_rows [piece, d] = dst_view's piece of this rank): the all-gather then moves THAT tensor's
    owned piece into dst_view in place instead of the reduced piece into `view` - the owner-updates form of the item table's step (the
    reduced gradient piece is consumed by `after`, which updates the owned parameter rows; what travels back are parameters)."""

    def __init__(self, comm: Comm, view: torch.Tensor, shard: torch.Tensor, after, gather=None):
        self.comm, self.view, self.shard, self.after, self.gather = comm, view, shard, after, gather
        self.h = None
        d = comm.dist
        if comm._host_staged():                            # gloo stand-in (CPU tests / two processes on one GPU): reduce each piece at its owner
            host = view.detach().cpu().reshape(comm.world, -1)
            for r in range(comm.world):
                part = host[r].clone()
                d.reduce(part, dst=r)
                if r == comm.rank:
                    shard.copy_(part.reshape(shard.shape))
        else:
            self.h = d.reduce_scatter_tensor(shard.view(-1), view.reshape(-1), async_op=True)

    def finish_local(self):
        if self.h is not None:
            self.h.wait()
        if self.after is not None:
            self.after(self.shard)
        d, comm = self.comm.dist, self.comm
        dst, src = (self.view, self.shard) if self.gather is None else self.gather
        if comm._host_staged():
            parts = [torch.empty(src.numel(), dtype=src.dtype) for _ in range(comm.world)]
            d.all_gather(parts, src.detach().cpu().reshape(-1))
            dst.copy_(torch.cat(parts).reshape(dst.shape))
            self.h = None
        else:
            self.h = d.all_gather_into_tensor(dst.reshape(-1), src.reshape(-1), async_op=True)   # (in place when src is dst's own piece)

    def wait(self):
        if self.h is not None:
            self.h.wait()
        return True


def all_reduce_start(comm: Comm, t: torch.Tensor):
    """Start an in-place sum all-reduce of t; returns a handle whose wait() orders the CURRENT stream after it.
    nccl (= RCCL): asynchronous on the communicator's stream (it first waits for the work already queued on the
    current stream, i.e. for the kernel that produced t); gloo / one rank: done on return."""
    if comm.dist is None or (comm.world == 1 and not comm.force):
        return _Done()
    if comm._host_staged():
        comm.all_reduce_(t)
        return _Done()
    return comm.dist.all_reduce(t, async_op=True)


def plan_chunks(n_items: int, d: int, world: int, n_chunks=None, exchange: str = "all_reduce", force: bool = False):
    """The item-row chunks [(r0, r1)] of an exchanged I x d message (pure arithmetic: bench.py composes an N-rank line's fields from it
    without launching). None = the policy: one chunk in a world of one rank (nothing to overlap), else pieces of >= 32 MB, at most 8;
    rs_ag: every chunk but possibly the last splits evenly over the ranks."""
    if n_chunks is None:
        n_chunks = 1 if (world == 1 and not force) else max(1, min(8, (4 * n_items * d) // (32 << 20)))
    n_chunks = max(1, min(int(n_chunks), n_items))
    per = (n_items + n_chunks - 1) // n_chunks
    if exchange == "rs_ag":
        per = (per + world - 1) // world * world
    return [(r0, min(r0 + per, n_items)) for r0 in range(0, n_items, per)]


def message_plan(n_items: int, d: int, n_layers: int, batch: int, n_chunks: int, exchange: str = "all_reduce", sparse_forward: bool = False) -> dict:
    """What one step of the row-sharded ID path exchanges (one I x d message per layer and direction), from the shapes alone."""
    n_msg = 2 * n_layers - (1 if sparse_forward else 0)
    return {"exchange": exchange, "allreduce_I_x_d_bytes": 4 * n_items * d * n_msg,
            "last_forward_message": "a fixed block of 2 B world rows (the batches' item rows, zeros behind the list's end)" if sparse_forward else "I x d",
            "allreduce_messages": n_msg * n_chunks + (1 if sparse_forward else 0), "chunk_bytes": 4 * d * ((n_items + n_chunks - 1) // n_chunks),
            "bpr_rows_allgather_bytes_per_rank": 2 * batch * (4 * d + 8), "prune_allgather_bytes_per_rank": 4 * batch}


class ShardedFusedID:
    """One rank's part of the fused ID-path training step."""

    def __init__(self, graph: ShardedGraph, comm: Comm, backend, d: int, n_layers: int, n_users_global: int, seed: int,
                 lr: float, batch_local: int, drop_rate: float, decay: float, n_chunks: Optional[int] = None,
                 user_init: Optional[torch.Tensor] = None, item_init: Optional[torch.Tensor] = None,
                 batch_size_flag: Optional[float] = None, exchange: str = "all_reduce", sparse_backward: bool = True,
                 sparse_forward: bool = True, owner_updates_items: Optional[bool] = None):
        """batch_size_flag: the divisor of the BPR regulariser - the reference divides by the --batch_size FLAG, not by the
        number of triples in the batch (main.py:340: augmented triples do not change it); default = batch_local * world.
        exchange: "all_reduce" | "rs_ag" (module docstring). sparse_backward: skip the all-zero operand rows in the two SpMMs of the last
        layer's backward (False: the dense products, for A/B runs). sparse_forward: compute the last layer's two forward products only in
        the rows the step reads (forward(needed=...)). owner_updates_items (default: on with "rs_ag"): the item table's AdamW is SHARDED - the
        last backward message's gradient is reduce-scattered, each rank updates the rows it owns (parameters + both moments: 28 B per
        parameter once per row instead of once per rank) and the all-gather returns updated table rows instead of gradient rows."""
        self.g, self.comm, self.be = graph, comm, backend
        self.d, self.L, self.B = d, n_layers, batch_local
        self.remember, self.decay = 1.0 - drop_rate, decay
        self.bsz_flag = float(batch_local * comm.world) if batch_size_flag is None else float(batch_size_flag)
        if exchange not in ("all_reduce", "rs_ag"):
            raise ValueError("ShardedFusedID: exchange must be all_reduce or rs_ag")
        self.exchange = exchange
        self.owner_updates_items = (exchange == "rs_ag") if owner_updates_items is None else bool(owner_updates_items and exchange == "rs_ag")
        dev = graph.s_i.device
        U, I = graph.n_users_local, graph.n_items
        self.U, self.I = U, I
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        # parameters: xavier_uniform over the GLOBAL table shapes (reference Models.py:39-42); the item table is drawn
        # identically on every rank, the user rows per rank
        def xavier(rows, bound, gen_seed):
            # drawn ON the device (25.6 GB tables at cfg 5); the same seed gives the same table on every rank
            gen = torch.Generator(device=dev); gen.manual_seed(gen_seed)
            t = torch.empty(rows, d, dtype=torch.float32, device=dev)
            t.uniform_(-bound, bound, generator=gen)
            return t
        if item_init is None:
            item_init = xavier(I, math.sqrt(6.0 / (I + d)), seed)
        if user_init is None:
            user_init = xavier(U, math.sqrt(6.0 / (n_users_global + d)), seed * 7919 + 1 + comm.rank)
        self.item_tab = torch.nn.Parameter(item_init.to(dev).float().contiguous())
        self.user_tab = torch.nn.Parameter(user_init.to(dev).float().contiguous())
        # forward / backward buffers. The user side is the big one (25.6 GB per [U, d] tensor at cfg 5): besides the table
        # and its two AdamW moments it holds L layer outputs and TWO work tensors - dE_u, which IS the table's gradient up
        # to the factor 1 / (L + 1) that AdamW applies on the fly (grad_scale), and hU = dU / h in the backward (and the
        # dense E_u of an evaluation). A training step never forms the dense E_u: the BPR loss needs its B batch rows only
        # (llmrec_gather_mean_f32)
        self.Ul = [f(U, d) for _ in range(n_layers)]
        self.Il = [f(I, d) for _ in range(n_layers)]
        self.dE_u, self.dE_i = torch.zeros(U, d, dtype=torch.float32, device=dev), torch.zeros(I, d, dtype=torch.float32, device=dev)
        self._scatter_dirty = False                           # True while rows of dE_u / dE_i may be non-zero (an aborted step)
        self.hU = f(U, d)
        self.E_u, self.E_i = self.hU, f(I, d)
        self.bufI, self.tmpI = f(I, d), f(I, d)
        self.item_tab.grad, self.user_tab.grad = f(I, d), self.dE_u
        self.opt = backend.optimizer([self.user_tab, self.item_tab], lr)
        # the backward runs the PATTERN operands (no per-edge col_scale gather): the tensor an SpMM gathers from is scaled by
        # its rows' s once, where it is produced (epilogue post_scale / llmrec_scale_rows_f32): A_iu^T g = R (s_i . g)
        self.R_user = backend.with_scales(graph.iu_bwd, None, None)       # rows = local users, gathers item rows
        self.s_u, self.s_i = graph.ui_fwd.row_scale, graph.s_i
        self.rows3 = f(3, batch_local, d)
        self.Eu_rows = f(batch_local, d)
        self.Ei_rows = f(2 * batch_local, d)                   # the layer means of the batch's positive, then negative items
        self.pn_ids = torch.empty(2 * batch_local, dtype=torch.int64, device=dev)
        self.arange_b = torch.arange(batch_local, dtype=torch.int64, device=dev)
        self.arange_neg = self.arange_b + batch_local
        self.gat_rows = f(comm.world, 2, batch_local, d)
        self.gat_ids = torch.empty(comm.world, 2, batch_local, dtype=torch.int64, device=dev)
        self.my_ids = torch.empty(2, batch_local, dtype=torch.int64, device=dev)
        # The backward of the LAST propagation layer works on sparse rows: the gradient that enters it (g = softmax_bwd(I_L, inv dE_i))
        # is non-zero only in the rows of the batch's items, its product A_iu^T g only in the rows of their neighbours (a few per cent of
        # the users). One byte per row says so (llmrec_spmm_epilogue_t x_row_mask / y_row_flag; the active value is a per-step stamp, so
        # nothing is ever cleared): those two SpMMs skip the gathers of all-zero rows - 2 of the step's 4 L products.
        self.sparse_backward = sparse_backward
        self.sparse_forward = sparse_backward and sparse_forward
        if self.sparse_forward:                                # the restricted last layer's row list (device) and its fixed-size message block
            cap = comm.world * 2 * batch_local
            if cap > _lib.CONST["LLMREC_SORT_UNIQUE_MAX"]:      #  the one-block id sort's limit: refuse at set-up, not at the first step
                raise RuntimeError("ShardedFusedID: the restricted forward lists world * 2 * batch = %d item ids per step; llmrec_sort_unique_ids_i32 "
                                   "takes at most %d - use sparse_forward=False (the dense forward) for this batch size"
                                   % (cap, _lib.CONST["LLMREC_SORT_UNIQUE_MAX"]))
            # (the layer's message is the FIXED-SIZE block of `cap` rows with a device-side count: with heavily duplicated batch items it
            #  carries more bytes than the distinct rows alone would - the price of never reading the count back to the host)
            self.need_rows = torch.zeros(cap, dtype=torch.int32, device=dev)
            self.need_n = torch.zeros(1, dtype=torch.int32, device=dev)
            self.need_part = torch.zeros(cap, d, dtype=torch.float32, device=dev)
        self.graph_items_to_users = graph.ui_bwd                 # rows = items, columns = this rank's users (the pattern is what matters)
        self.flag_u = torch.zeros(U, dtype=torch.uint8, device=dev)
        self.flag_i = torch.zeros(I, dtype=torch.uint8, device=dev)
        # item-row chunks of the two SpMMs whose output is all-reduced (>= 32 MB per message unless told otherwise)
        self.chunks = plan_chunks(I, d, comm.world, n_chunks, exchange, force=comm.force)
        # rs_ag: one staging piece per chunk (the rows this rank reduces and owns between the two halves of the exchange)
        self.shards = [f((r1 - r0) // comm.world, d) if (exchange == "rs_ag" and (r1 - r0) % comm.world == 0) else None
                       for r0, r1 in self.chunks]
        self.iu_fwd_chunks = [backend.row_chunk(graph.iu_fwd, r0, r1) for r0, r1 in self.chunks]
        self.ui_bwd_chunks = [backend.row_chunk(backend.with_scales(graph.ui_bwd, None, None), r0, r1) for r0, r1 in self.chunks]   # R^T pattern
        deg = backend.degrees(graph.by_user)
        self.exist = torch.nonzero(deg > 0).reshape(-1).to(torch.int64)
        self.seed = seed * 1000003 + comm.rank
        self.step_id = 0
        self._last_stamp = None                               # the stamp of the previous step (marks are cleared when it does not advance by one)
        self.allreduce_bytes = 0

    def parameters(self):
        return [self.user_tab, self.item_tab]

    # -- the chunked, overlapped all-reduce ----------------------------------------------------------
    def _reduced_spmm(self, chunk_ops, X, out, epilogue_for=None, after=None, owner_update=None):
        """out[rows_c] = sum over ranks of chunk_ops[c] @ X, chunk by chunk: the exchange of chunk c is in flight
        while chunk c + 1 is computed. after(view): optional row-local op on reduced rows (the softmax) - on the whole
        chunk after an all-reduce, on the owned piece between the reduce-scatter and the all-gather of "rs_ag".
        owner_update = (table, fn(row0, row1, reduced_rows)) ("rs_ag" only; `out` is the table's gradient): the rank that owns a piece
        of a chunk applies fn to ITS rows of `table` with the reduced gradient piece and the all-gather moves the updated TABLE rows (in
        place) instead of the gradient; chunks that do not split evenly are all-reduced and every rank applies fn to all their rows."""
        comm = self.comm
        collective = comm.dist is not None and (comm.world > 1 or comm.force)
        pending, started = [], []
        for c, (r0, r1) in enumerate(self.chunks):
            view = out[r0:r1]
            self.be.spmm(chunk_ops[c], X, out=view, epilogue=epilogue_for(r0, r1) if epilogue_for else None)
            self.allreduce_bytes += view.numel() * 4
            if collective and self.exchange == "rs_ag" and self.shards[c] is not None:
                if owner_update is not None:
                    table, fn = owner_update
                    piece = (r1 - r0) // comm.world
                    o0 = r0 + comm.rank * piece
                    ex = _ShardedExchange(comm, view, self.shards[c], (lambda sh, a=o0, b=o0 + piece: fn(a, b, sh)),
                                          gather=(table[r0:r1], table[o0:o0 + piece]))
                else:
                    ex = _ShardedExchange(comm, view, self.shards[c], after)
                if started:                                   # one chunk of lag: its reduce-scatter ran beside this chunk's SpMM
                    started.pop().finish_local()
                started.append(ex)
                pending.append((ex, view, False, (r0, r1)))
            else:
                pending.append((all_reduce_start(comm, view), view, True, (r0, r1)))
        while started:
            started.pop().finish_local()
        for h, view, whole, (r0, r1) in pending:
            h.wait()
            if whole and owner_update is not None:
                owner_update[1](r0, r1, view)                 # (replicated update of the rows of a chunk that went through an all-reduce)
            elif whole and after is not None:
                after(view)

    # -- forward ---------------------------------------------------------------------------------------
    def forward(self, dense_users: bool = True, dense_items: bool = True, needed=None):
        """dense_users / dense_items = False (training): E_u / E_i are not materialised (the loss gathers the layer means of its batch rows:
        one pass over B rows instead of one over the whole tables).
        needed = (stamp, item_rows) (training, sparse_forward): the LAST layer's two products are computed for the rows somebody reads -
        U_L for the users flag_u marks with `stamp` (the batch's users and the users the batch's items reach), I_L for the listed item
        rows (the items of every rank's batch) - and the message of that layer is those rows instead of the I x d table."""
        be, L = self.be, self.L
        i_prev = self.item_tab.detach()
        for l in range(L):
            last = l == L - 1
            if last and needed is not None:
                # no host read-back anywhere (round 5): the list of item rows and its length stay on the device, the message is the
                # fixed-size block [2 B world, d] whose slots past the list's end are zeros
                stamp, rows, n_rows = needed
                be.spmm(self.g.ui_fwd, i_prev, out=self.Ul[l], epilogue={"op": "softmax", "y_row_needed": self.flag_u, "x_mask_active": stamp})
                part = self.need_part
                be.spmm_rows_compact(self.g.iu_fwd, self.Ul[l], rows, n_rows, part)   # this rank's share of the listed rows
                self.allreduce_bytes += part.numel() * 4
                if self.comm.dist is not None and (self.comm.world > 1 or self.comm.force):
                    self.comm.all_reduce_(part)
                be.softmax_rows_into(part, part)
                be.scatter_set_rows(rows, n_rows, part, self.Il[l])
                break
            be.spmm(self.g.ui_fwd, i_prev, out=self.Ul[l], epilogue={"op": "softmax"} if last else None)      # local users
            self._reduced_spmm(self.iu_fwd_chunks, self.Ul[l], self.Il[l],
                               after=(lambda view: be.softmax_rows_into(view, view)) if last else None)
            i_prev = self.Il[l]
        if dense_users:
            be.layer_mean_into([self.user_tab.detach()] + self.Ul, self.E_u)
        if dense_items:
            be.layer_mean_into([self.item_tab.detach()] + self.Il, self.E_i)
        return self.E_u, self.E_i

    # -- one training step -------------------------------------------------------------------------------
    def sample(self):
        return self.be.sample(self.seed, self.step_id, self.exist, self.I, self.g.by_user, self.B)

    def step(self, triples=None):
        """triples: (users RELATIVE to this rank's block, pos, neg) int64 device vectors of length batch_local, or None
        to draw them with the device sampler. Returns (loss, [mf, emb]) of the GLOBAL batch as device scalars."""
        be, comm, L, B = self.be, self.comm, self.L, self.B
        u, p, n = triples if triples is not None else self.sample()
        self.step_id += 1
        self.allreduce_bytes = 0
        # the rows this step touches, known before the forward: the items of every rank's batch (all-gather of 2 B ids) and, through
        # their adjacency, the users they reach; byte marks with a per-step stamp (nothing is cleared)
        stamp = self.step_id % 255 + 1
        if self.sparse_backward and (stamp == 1 or self._last_stamp != stamp - 1):
            # the stamps start a new cycle - or step_id was set from outside (a resume): no mark of an earlier cycle may survive - g below
            # is formed in the marked item rows only, a stale mark that equals a later stamp would make the masked product read a row of g
            # left over from an earlier step. The marks are cleared whenever the stamp does not advance by exactly one.
            self.flag_u.zero_(); self.flag_i.zero_()
        self._last_stamp = stamp
        self.my_ids[0].copy_(p); self.my_ids[1].copy_(n)
        comm.all_gather_into(self.gat_ids.view(-1), self.my_ids.view(-1))
        needed = None
        if self.sparse_backward:
            be.mark_rows(u, stamp, self.flag_u)                   # the non-zero rows of dE_u ...
            be.mark_rows(self.gat_ids.view(-1), stamp, self.flag_i)   # ... and of dE_i (hence of g)
            # ... and the local users those items reach: the only rows of A_iu^T g that can be non-zero (a sweep over the adjacency of
            # <= 2 B world items instead of a look at every user's index list)
            be.mark_neighbours(self.gat_ids.view(-1), self.graph_items_to_users, stamp, self.flag_u)
            if self.sparse_forward:                                # the distinct item rows of every rank's batch, ascending: identical on every
                be.sort_unique_ids(self.gat_ids.view(-1), self.need_rows, self.need_n)   # rank; built on the device (padded ids < 0 skipped)
                needed = (stamp, self.need_rows, self.need_n)
        self.forward(dense_users=False, dense_items=False, needed=needed)
        # BPR + prune over the global batch (reference main.py:158-165,330-342): two passes around an all-gather of B floats;
        # the user side of the loss is the B x d block of layer-mean rows, indexed 0..B-1
        be.gather_mean_into([self.user_tab.detach()] + self.Ul, u, self.Eu_rows)
        B0 = self.arange_b.numel()
        self.pn_ids[:B0].copy_(p); self.pn_ids[B0:].copy_(n)
        be.gather_mean_into([self.item_tab.detach()] + self.Il, self.pn_ids, self.Ei_rows)
        ar, E_i = self.arange_b, self.Ei_rows                  # the loss indexes the compact blocks: user b, positive b, negative B + b
        p_loc, n_loc = self.arange_b, self.arange_neg
        _, s1 = be.bpr_fwd(self.Eu_rows, E_i, ar, p_loc, n_loc, self.remember, self.decay, self.bsz_flag, None, 0, 0, True)
        global_m = comm.all_gather_cat(be.bpr_local_m(s1, B).contiguous())
        out, saved = be.bpr_fwd(self.Eu_rows, E_i, ar, p_loc, n_loc, self.remember, self.decay, self.bsz_flag, global_m, global_m.numel(),
                                comm.rank * B, False)
        small = torch.cat([saved[B:B + 3], out[:1]])
        comm.all_reduce_(small)                                   # the three squared norms + the mf shares
        saved[B:B + 3] = small[:3]
        mf = small[3:4]
        emb = (self.decay * ((1.0 / (2.0 * small[:3] + 1e-8)).sum() / self.bsz_flag)).reshape(1)
        # backward: compact gradient rows; users scatter locally, item rows are exchanged (all-gather of 2 B rows)
        ones = torch.ones(2, dtype=torch.float32, device=out.device)
        be.bpr_bwd_rows(self.Eu_rows, E_i, ar, p_loc, n_loc, self.decay, self.bsz_flag, saved, ones, self.rows3)
        if self._scatter_dirty:                               # only after a step that did not reach its clean-up
            be.zero_([self.dE_u, self.dE_i])
        self._scatter_dirty = True
        be.scatter_rows(u, self.rows3[0], self.dE_u, 1.0)
        comm.all_gather_into(self.gat_rows.view(-1), self.rows3[1:3].reshape(-1))
        be.scatter_rows(self.gat_ids.view(-1), self.gat_rows.view(-1, self.d), self.dE_i, 1.0)     # same list, same order on every rank
        inv = 1.0 / (L + 1)
        # dI[L] = inv dE_i -> g = softmax_bwd(I_L, dI[L]); then per layer
        #   dU[l+1] = inv dE_u + A_iu[:, blk]^T g          (local; softmax backward as the epilogue on the last layer)
        #   dI[l]   = inv dE_i + sum_r A_ui[blk_r, :]^T h  (chunked + all-reduced; every rank adds inv / world of the replicated dE_i)
        #   (g and h below are stored PRE-SCALED by s_i / s_u, see __init__)
        g = self.bufI
        if L >= 1 and self.sparse_backward:
            # g = s_i . softmax_bwd(I_L, inv dE_i) in the rows of the batch's items only: the masked product below reads no other row
            # (three dense passes over the I x d tables otherwise)
            be.softmax_bwd_listed_into(self.gat_ids.view(-1), inv, self.Il[L - 1], self.dE_i, self.s_i, self.tmpI)
            g = self.tmpI
        elif L >= 1:
            be.axpy_into(inv, self.dE_i, self.bufI)
            be.softmax_bwd_into(self.Il[L - 1], self.bufI, self.tmpI)
            be.scale_rows_into(self.s_i, self.tmpI, self.tmpI)
            g = self.tmpI
        for l in range(L - 1, -1, -1):
            last = l == L - 1
            epi = {"op": "softmax_bwd" if last else "none", "alpha": inv, "Z": self.dE_u, "post_scale": self.s_u}
            sparse = last and self.sparse_backward                # (g is the softmax backward of the scattered rows only on the last layer)
            if last:
                epi["S"] = self.Ul[l]
            if sparse:                                            # rows of g outside the batch's items are zero: not gathered; flag_u := rows of h that can be non-zero
                epi.update({"x_row_mask": self.flag_i, "x_mask_active": stamp, "z_row_flag": self.flag_u, "y_row_gate": self.flag_u})
            be.spmm(self.R_user, g, out=self.hU, epilogue=epi)                           # h = s_u . dU[l+1] (softmax backward on the last layer)
            dst = self.item_tab.grad if l == 0 else self.bufI
            w = inv / comm.world
            # (h itself is NOT sparse enough to mask: the users two hops from the batch - through its most popular items - are ~40 % of
            #  all users at cfg 4, and the masked product then costs more than the dense one: 9.2 vs 6.2 ms measured)
            sharded_update = l == 0 and self.owner_updates_items
            if sharded_update:                                    # this message IS the item table's gradient: owners update, parameters travel back
                be.optimizer_advance(self.opt)
                upd = (self.item_tab.data, lambda a, b, rows: be.optimizer_step_rows(self.opt, self.item_tab, a, b, rows))
            self._reduced_spmm(self.ui_bwd_chunks, self.hU, dst,
                               epilogue_for=lambda r0, r1: {"op": "none", "alpha": w, "Z": self.dE_i[r0:r1],
                                                            "post_scale": None if l == 0 else self.s_i[r0:r1]},
                               owner_update=upd if sharded_update else None)
            g = self.bufI
        if L == 0:
            be.axpy_into(inv, self.dE_i, self.item_tab.grad)
        # user_tab.grad = inv * dE_u (U^0 only enters the mean): the factor rides in AdamW instead of a pass over the table
        if self.owner_updates_items and L >= 1:
            be.optimizer_step_params(self.opt, [self.user_tab], {self.user_tab: inv})
        else:
            be.optimizer_step(self.opt, {self.user_tab: inv})
        # last readers done (the l = 0 message read dE_i, AdamW read dE_u as the user table's gradient): clear the touched rows
        be.zero_rows(u, self.dE_u)
        be.zero_rows(self.gat_ids.view(-1), self.dE_i)
        self._scatter_dirty = False
        return (mf + emb).reshape(()), torch.cat([mf, emb])

    # -- evaluation ------------------------------------------------------------------------------------------
    def eval_topk(self, query_users_local: torch.Tensor, K: int = 50, forward: bool = True):
        """Full-rank evaluation of this rank's users (reference utility/batch_test.py:112-169): users shard, the item table
        is replicated, so ranks are independent - no-grad forward (with its layer all-reduces) + scoring + masked top-K
        against the local training rows. Returns (idx int32 [n, K], scores)."""
        if forward:
            self.forward()
        return self.be.score_topk(self.E_u, self.E_i, query_users_local, self.g.by_user, K)

    # -- accounting for bench.py -------------------------------------------------------------------------
    def message_bytes_per_step(self) -> dict:
        out = message_plan(self.I, self.d, self.L, self.B, len(self.chunks), self.exchange, self.sparse_forward)
        out["exchanged_bytes_last_step"] = int(self.allreduce_bytes)
        return out
