"""Fused training step: the whole Stage-2 step - sampler included - as a fixed sequence of 28 HIP launches (36 with LLMREC_FOLD=0)
over preallocated buffers, with the backward pass written out by hand (no autograd graph), captured into one
HIP graph; likewise an evaluation (forward + scoring + masked top-K).

Same arithmetic as the modular path (Models.MM_Model.forward + llmrec_amd.engine.train_step,
i.e. reference Models.py:127-199 + main.py:228-278); what changes is the organisation:
  * the seven item-side streams (image, text, 5 attributes) travel as ONE [rows, 7d] operand, so
    the 16 side-feature SpMMs of the reference (Models.py:152-167) become 4 (+4 in backward) - 2 + 2 when the constant
    products A_ui F_k are pre-propagated at set-up - and the adjacency indices are read once per direction instead of seven times;
  * the 8 BPR + prune losses (main.py:232-254) are two launches on the critical path (scores - which also begin the step on the device:
    row stamp, AdamW's counter - and selection + gradient rows) and one beside it (the loss values, the feature regulariser from the
    fusion launch's partial sums, the logged scalars);
  * projections and weight gradients of the constant feature matrices are grouped launches (8 projections in
    one; all four Linears' gradients, row-listed, in one, with their AdamW in the reduction launch) in split-precision bf16x3
    arithmetic (LLMREC_GEMM=f32: exact fp32);
  * independent chains run on four HIP streams (fork/join = graph edges);
  * the embedding tables' AdamW reads its gradient where it is formed (the user table's is inv * dE_u) and the row-wise clean-up of the
    scatter targets rides in the item table's AdamW launch;
  * nothing allocates or synchronises, so the step can be replayed from a HIP graph
    (``capture()``), removing the per-launch host cost that dominates at Netflix scale.
Falls outside its scope (use the modular path): dropout > 0 and the --mask branch.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _lib, ops
from .engine import Hyper

_p, _ld, _c = ops._p, ops._ld, ctypes


def _call(name, *a):
    _lib.call(name, *a, ops._stream())


class _capture_without_gc:
    """torch.cuda.graph(g) with Python's cyclic garbage collector held off for the duration of the capture. torch.cuda.graph collects
    garbage once, on entry; the body of a captured step then creates tens of thousands of container objects (ctypes argument arrays,
    tuples), so generation-0/1/2 collections run INSIDE the capture and may finalise whatever cyclic garbage exists by then - a
    previously imported drop-in module with its device tensors, an older step object with its graph executables - i.e. free device
    memory and destroy executables in the middle of a stream capture. Hygiene, not a fix for anything observed: the host fault round 4
    chased through the GPU suite turned out to be hipGraphLaunch's own (llmrec_amd/__init__.py)."""

    def __init__(self, graph):
        from . import require_graph_replay
        require_graph_replay("llmrec_amd: stream capture")      # the hardware-queue work-around must be in force (llmrec_amd/__init__.py)
        self.ctx = torch.cuda.graph(graph, capture_error_mode="thread_local")   # other threads (the RCCL watchdog) may touch the runtime

    def __enter__(self):
        import gc
        self.was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was_enabled:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_enabled:
                gc.enable()


class FusedStep:
    def __init__(self, model, graph, hp: Hyper, rates, optimizer: ops.FusedAdamW, b_max: int):
        """model: Models.MM_Model (parameters + constant feature tensors); graph: ops.BipartiteGraph
        or any object with .ui/.iu SparseOperands; rates: (model_cat, user_cat, item_cat)."""
        self.m, self.hp, self.opt = model, hp, optimizer
        self.ui, self.iu = graph.ui, graph.iu
        self.U, self.I = model.n_users, model.n_items
        self.d = d = model.embedding_dim
        self.L = model.n_ui_layers
        self.keys = list(model.item_feats.keys())
        self.S = S = 2 + len(self.keys)                      # item-side streams: image, text, attributes
        self.c_m, self.c_u, self.c_a = rates
        self.b_max = b_max
        dev = model.item_id_embedding.weight.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        U, I = self.U, self.I
        # forward buffers
        self.U_cat, self.I_cat = f(U, S * d), f(I, S * d)
        self.P_usr, self.prof_i, self.prof_u = f(U, d), f(I, d), f(U, d)
        self.Ul = [f(U, d) for _ in range(self.L)]
        self.Il = [f(I, d) for _ in range(self.L)]
        self.E_u, self.E_i = f(U, d), f(I, d)
        # backward buffers
        # The loss backward scatters into buffers that are all-zero between steps - dE_u / dE_i and the sc_* sources of the fusion
        # backward - and clears exactly the rows it touched afterwards (llmrec_bpr_multi_zero_rows_f32); the fusion backward
        # writes d*_cat = source + its own term (llmrec_fuse_bwd_src_f32). No dense memset of the six gradient buffers (70 MB
        # per step at the Netflix shape; measured -2.2 % of the step in round 2).
        # LLMREC_CHECK_ZERO=1 (debug; synchronises, so not under capture): assert that invariant before every scatter
        self.check_zero = os.environ.get("LLMREC_CHECK_ZERO", "0") == "1"
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.dE_u, self.dE_i = z(U, d), z(I, d)
        self.sc_U, self.sc_I, self.sc_prof = z(U, 2 * d), z(I, S * d), z(U, d)
        # one byte per row, stamped by the loss launch for the rows of the batch with the step's stamp (a device counter: nothing is ever
        # cleared, no reader races a clean-up): the fusion backward writes the ~90 % of rows no sample touched (zero gradient, zero
        # sources) without reading them. (The same stamps offered to the transposed side product as "rows whose attribute streams are
        # all-zero" bought nothing: at 4 edges per row that SpMM is bound by its per-row latency chain, not by the gathers.)
        self.flag_u = torch.zeros(U, dtype=torch.uint8, device=dev)
        self.flag_i = torch.zeros(I, dtype=torch.uint8, device=dev)
        self.row_stamp = torch.zeros(1, dtype=torch.int32, device=dev)     # advanced by the scores launch (LLMREC_ROW_STAMP)
        # the batch's deterministic scatter plan (llmrec_bpr_scatter_plan: sorted (id, slot) keys + run lengths), built right behind the
        # sampler: the loss backward adds the rows several samples share in one fixed order - no float atomics, same-seed runs agree bit for bit
        self.bpr_plan = torch.zeros(max(ops.bpr_plan_words(b_max), 1), dtype=torch.int64, device=dev)
        # running sums (double) of the logged scalars [loss, mf, emb] over the steps since the caller last cleared them: the epoch line of
        # reference main.py:280-283 without any host-side arithmetic between graph replays
        self.epoch_sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self.dU_cat, self.dI_cat = f(U, S * d), f(I, S * d)
        self.dprof_u, self.dprof_i, self.dP_usr = f(U, d), f(I, d), f(U, d)
        self.bufU, self.bufI, self.tmpU, self.tmpI = f(U, d), f(I, d), f(U, d), f(I, d)
        self.out = f(8, 2)
        self.saved = f(8 * ops.bpr_saved_floats(b_max))
        self.scal = f(4)                                     # [feat_reg, loss, mf, emb]
        self.ws_sumsq = torch.empty(_lib.query("llmrec_sumsq_workspace_bytes", 0, 0), dtype=torch.uint8, device=dev)
        feats = [model.image_feats, model.text_feats, model.user_feats] + [model.item_feats[k] for k in self.keys]
        ws = max(_lib.query("llmrec_linear_wgrad_workspace_bytes", x.shape[0] * (len(self.keys) if i >= 3 else 1), d, x.shape[1])
                 for i, x in enumerate(feats))
        self.ws_wgrad = torch.empty(ws, dtype=torch.uint8, device=dev)
        wsq = lambda x: torch.empty(_lib.query("llmrec_linear_wgrad_workspace_bytes", x.shape[0], d, x.shape[1]), dtype=torch.uint8, device=dev)
        # one workspace per weight gradient that may be in flight at the same time (user / text / image run beside item_trans')
        self.ws_wgrad_b, self.ws_wgrad_c, self.ws_wgrad_d = wsq(model.user_feats), wsq(model.text_feats), wsq(model.image_feats)
        self.ws_wgrad_multi = None                           # sized at the first backward (needs the gradient tensors)
        self._partials = {}
        # Independent chains (7-stream side features / LLM profile / ID embeddings / the two fusion halves / the logged scalars)
        # run on five HIP streams in forward and in backward; under capture the fork / join become graph edges, so the
        # latency-bound Netflix-scale SpMMs overlap the GEMMs. LLMREC_STREAMS=0 keeps everything on one stream (debugging).
        self.multi_stream = os.environ.get("LLMREC_STREAMS", "1") == "1"
        self.s1, self.s2, self.s3 = (torch.cuda.Stream(device=dev) for _ in range(3))
        for p in model.parameters():
            if p.requires_grad and p.grad is None and p is not model.batch_norm.weight and p is not model.batch_norm.bias:
                p.grad = torch.zeros_like(p)
        # loss weights of the 8 BPR problems (main.py:273): main (mf + emb), image/text (mm_mf_rate), attributes (aug_mf_rate)
        self.w_mf = [1.0, hp.mm_mf_rate, hp.mm_mf_rate] + [hp.aug_mf_rate] * len(self.keys)
        self.w_emb = [1.0] + [0.0] * (S)
        self.n_prob = 1 + S
        if self.n_prob > _lib.CONST["LLMREC_BPR_MAX_PROBLEMS"]:
            raise RuntimeError("FusedStep: %d BPR problems exceed LLMREC_BPR_MAX_PROBLEMS" % self.n_prob)
        self.w_mf_dev = torch.tensor(self.w_mf, dtype=torch.float32, device=dev)
        self.graph_exec = None
        self.static = None
        self._eval_graphs = {}
        # measurement only (bench.py `roofline`): a uint64 tensor of STAMP_SLOTS device timestamps written by llmrec_timestamp launches around
        # the step, the projection launch and the weight-gradient launches - the durations INSIDE a replayed graph, which cannot carry events
        self.stamps = None
        self.spmm_edge_units = 0.0                            # edge traversals (nnz x d / 64) of the SpMM launches of the last step built
        # projection / weight-gradient arithmetic: "bf16x3" (default) = exact 3-term bf16 split of both operands, six bf16
        # MFMAs, fp32-roundoff-class error (2e-6 measured); "f32" = the exact fp32 MFMA fma chain
        self.gemm = os.environ.get("LLMREC_GEMM", "bf16x3")
        # PRE-PROPAGATED item-side operands (LLMREC_PREPROPAGATE=0 restores the reference's order of operations). The features F_k
        # and the adjacency are constants of a run, and projection followed by propagation is one product:
        #     A_ui (F_k W^T + 1 b^T) = (A_ui F_k) W^T + (A_ui 1) b^T            (Models.py:145-157: image / text / the 5 attributes)
        # so A_ui F_k [U, K] is formed ONCE here, the step projects IT (llmrec_linear_problem_t.bias_scale = the row sums of A_ui) straight
        # into U_cat, and the backward takes dW = dU_cat^T (A_ui F_k), db = sum_u (A_ui 1)[u] dU_cat[u]. Per step this removes the two
        # [.., 7 d] SpMMs through A_ui (forward and transposed) from the critical path and shrinks both GEMMs from I = 17 366 to
        # U = 13 187 rows; every parameter-dependent product is still computed every step, only the order of the (associative)
        # products changes: results agree with the other order to fp32 rounding (tests/test_gpu_step.py, the bench's parity gate).
        # SHAPE-AWARE (VERDICT r03 next #4c): pre-propagation moves both GEMMs from I rows to U rows and takes the two [., 7 d] SpMMs off the
        # critical path. It wins when the user side is the smaller one (Netflix shape, U / I = 0.76: 0.46 vs 0.62 ms per step) and loses
        # when it is the larger one (MovieLens shape, U / I = 1.21: 0.49 vs 0.48 ms, measured with the row-listed weight gradient on):
        # default = pre-propagate iff U <= I. LLMREC_PREPROPAGATE=1 / 0 forces either order.
        want = os.environ.get("LLMREC_PREPROPAGATE", "auto")
        self.preprop = d <= 64 and (want == "1" or (want not in ("0", "1") and U <= I))
        if not self.preprop:
            self.P_cat, self.dP_cat = f(I, S * d), f(I, S * d)        # the projected features and their gradient (reference order)
        if self.preprop:
            item_feats = [model.image_feats, model.text_feats] + [model.item_feats[k] for k in self.keys]
            self.AX = [self._propagate_constant(self.ui.fwd, x) for x in item_feats]
            self.a_rowsum = ops.spmm_raw(self.ui.fwd, torch.ones(I, 4, dtype=torch.float32, device=dev))[:, 0].contiguous()   # A_ui 1
            self.ws_colsum = torch.empty(_lib.query("llmrec_weighted_colsum_workspace_bytes", S * d), dtype=torch.uint8, device=dev)
        # ROW-LISTED weight gradient (VERDICT r03 next #4a; LLMREC_WGRAD_ROWS=0: the dense launch). The gradient of the five attribute
        # streams' projected features is EXACTLY ZERO outside the rows the batch reaches: an attribute stream meets the loss in the fused
        # embeddings and BPR terms of the batch's rows only (Models.py:160-163,188-197; main.py:249-254), so dU_cat[:, attribute columns] is
        # non-zero in the batch's users (fusion backward) and in the users adjacent to the batch's items (A_iu^T of the items' rows) - 43 %
        # of the users at the Netflix shape. llmrec_batch_reach_rows lists those rows (two small launches beside the forward) and the
        # weight-gradient launch streams the listed rows of dU_cat and A_ui F_k only (llmrec_wgrad_problem_t.row_list): the same sums
        # without the terms that are zero. The image / text streams stay dense (the feature regulariser reads every row, main.py:151-156),
        # and so does user_trans (its gradient is two hops wide: 66 % of the rows).
        self.wgrad_rows = (getattr(type(self), "WGRAD_ROWS", True) and os.environ.get("LLMREC_WGRAD_ROWS", "1") == "1" and self.preprop
                           and len(self.keys) > 0 and self.gemm == "bf16x3")     # (the exact-fp32 launches ignore the list: do not build it)
        # resident blocks the weight-gradient launch is laid out for (llmrec_wgrad_target_t.block_budget; 0 = 256, one per CU). A block of
        # that launch owns its CU's whole register file, so a 256-block round starves whatever runs beside it: the ID chain's last SpMM took
        # 81 us beside it and 12 us alone and, once the row list had shortened the GEMM, had become the step's tail. 224 blocks leave 32 CUs
        # to the other streams: 256 / 240 / 224 / 208 / 192 blocks -> 0.474 / 0.459 / 0.454 / 0.456 / 0.469 ms per step on one box.
        self.wgrad_blocks = int(os.environ.get("LLMREC_WGRAD_BLOCKS", "224"))
        if self.wgrad_rows:
            self.act_flags = torch.zeros(U + 16, dtype=torch.uint8, device=dev)[:U]   # all-zero between calls (readable in 16-byte words)
            self.act_rows = torch.zeros(U + 32, dtype=torch.int32, device=dev)
            self.act_n = torch.zeros(1, dtype=torch.int32, device=dev)
            self.act_expected = None      # rows the launch geometry is laid out for: set from the first step's count (capture() / step_eager())
        # AdamW inside the step, in two launches: the embedding tables as soon as the ID chain's backward has produced their
        # gradients (beside the weight-gradient GEMM), the four Linears right after the slab reduction of that GEMM - the
        # step's tail is then one small launch instead of reduction -> cross-stream join -> a 1.96 M-parameter update.
        # (Batch-sharded replicas all-reduce the gradients first and update afterwards: llmrec_amd/dp.py sets this to False.)
        self.inline_adamw = getattr(type(self), "INLINE_ADAMW", True)
        # FOLDED launches (round 5; LLMREC_FOLD=0 restores the 36-launch step): AdamW's counter advances inside the scores
        # launch ("a new step begins": llmrec_bpr_multi_scores_step_f32), the feature regulariser's sum of squares comes out of the fusion launch
        # as per-block partials (llmrec_fuse_fwd_multi_sumsq_f32: it holds those rows anyway) and is reduced by the ONE launch that also forms
        # the loss values and assembles the logged scalars (llmrec_bpr_multi_losses_assemble_f32), the user table's AdamW reads inv * dE_u
        # directly and stores it as the table's .grad on the way (no axpy), and the row-wise clean-up of the scatter targets rides in the item
        # table's AdamW launch (llmrec_adamw_multi_zero_rows_f32): 36 -> 28 launches, same arithmetic except the regulariser's summation tree.
        self.fold = getattr(type(self), "FOLD", True) and os.environ.get("LLMREC_FOLD", "1") == "1" and self.inline_adamw
        if self.fold:
            self.ss_cap = 2 * 2048                                # >= the fusion launch's grid (two problems, <= 2048 blocks each)
            self.ss_partial = torch.zeros(self.ss_cap, dtype=torch.float32, device=dev)
            self.ss_n = 0
        self._zero_in_forward = False                         # set by step_eager: forward() alone (evaluation) must not advance AdamW
        self._emb_params = [model.user_id_embedding.weight, model.item_id_embedding.weight]
        self._lin_params = [p for p in optimizer.params if p.grad is not None and all(p is not e for e in self._emb_params)]
        # Launch (= capture) order at the fork points decides which branch the graph runs behind its parent without a cross-queue
        # hand-off (~10 us). Measured one at a time in round 2 (profiles/README.md): the forward's side-feature SpMMs are
        # captured before the profile chain, the BPR backward before the regulariser / loss assembly (-3.6 % together); the
        # other fork points are neutral or lose.

    # -- raw kernel helpers -----------------------------------------------------------------------
    @staticmethod
    def _propagate_constant(a: ops.Csr, X: torch.Tensor) -> torch.Tensor:
        """A X for a constant feature matrix, 64 columns at a time (set-up only)."""
        X = X.detach()
        out = torch.empty(a.n_rows, X.shape[1], dtype=torch.float32, device=X.device)
        step = 64 if X.shape[1] % 64 == 0 else X.shape[1]
        for c0 in range(0, X.shape[1], step):
            ops.spmm_raw(a, X[:, c0:c0 + step], out=out[:, c0:c0 + step])
        return out

    def _fork(self, *streams):
        if self.multi_stream:
            cur = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(cur)

    def _mark(self):
        """An event at the current point of the current stream (a fork point that side streams may wait for later)."""
        if not self.multi_stream:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def _fork_from(self, ev, *streams):
        if self.multi_stream:
            for st in streams:
                st.wait_event(ev)

    def _join(self, *streams):
        if self.multi_stream:
            cur = torch.cuda.current_stream()
            for st in streams:
                cur.wait_stream(st)

    def _on(self, st):
        """Context: launch on side stream `st` (or stay on the current stream when disabled)."""
        return torch.cuda.stream(st) if self.multi_stream else torch.cuda.stream(torch.cuda.current_stream())

    STAMP_SLOTS = 6       # step begin, projection begin / end, weight gradient begin / end (reduction included), step end

    def _stamp(self, slot: int):
        if self.stamps is not None:
            _call("llmrec_timestamp", self.stamps.data_ptr() + 8 * slot)

    def _spmm(self, a: ops.Csr, X, Y, accumulate=False, tag=0, epilogue=None):
        """Y = epilogue(A X) (llmrec_spmm_f32); accumulate: Y += A X. tag: one partial-sum scratch per concurrent chain."""
        d = X.shape[1]
        self.spmm_edge_units += a.nnz * (d / 64.0)
        sw, pl = a.plan_for(d, whole_row=epilogue is not None and epilogue.op != ops.EPI_NONE)
        partials = None
        if pl.n_seg:
            key = (id(pl), d, tag)
            partials = self._partials.get(key)
            if partials is None:
                partials = self._partials[key] = torch.empty(pl.n_seg * d, dtype=torch.float32, device=X.device)
        if accumulate:
            epilogue = ops.spmm_epilogue(ops.EPI_NONE, 1.0, Y)
        if epilogue is None and os.environ.get("LLMREC_SPMM_PIPELINE", "1") == "0":      # (A/B switch: one task per lane group)
            epilogue = ops.spmm_epilogue()
        rp, ci = pl.csr_of(a)
        _call("llmrec_spmm_f32", a.n_rows, a.n_cols, _p(rp), _p(ci), _p(a.val), _p(a.row_scale), _p(a.col_scale),
              _p(X), _ld(X), _p(Y), _ld(Y), d, sw, _c.byref(pl.c_struct()), _p(partials),
              _c.byref(epilogue) if epilogue is not None else None)

    def _linear(self, X, lin, out):
        _call("llmrec_linear_fwd_f32", X.shape[0], self.d, X.shape[1], _p(X), _ld(X), _p(lin.weight), _ld(lin.weight), _p(lin.bias),
              _p(out), _ld(out))

    def projection_jobs(self):
        """(X, Linear, out, bias_scale) of the step's 8 projections (Models.py:145-150), longest K first: the launch's tail then
        consists of the short (512 / 768-wide) work units. Pre-propagated: the item-side operands are A_ui F_k and land in U_cat."""
        m = self.m
        if self.preprop:
            lins = [m.image_trans, m.text_trans] + [m.item_trans] * len(self.keys)
            jobs = [(self.AX[s_], lins[s_], self._side(self.U_cat, s_), self.a_rowsum) for s_ in range(self.S)]
        else:
            jobs = [(m.item_feats[key], m.item_trans, self._side(self.P_cat, 2 + k), None) for k, key in enumerate(self.keys)]
            jobs += [(m.text_feats, m.text_trans, self._side(self.P_cat, 1), None), (m.image_feats, m.image_trans, self._side(self.P_cat, 0), None)]
        jobs.append((m.user_feats, m.user_trans, self.P_usr, None))
        jobs.sort(key=lambda j: -j[0].shape[1])
        return jobs

    def wgrad_targets(self, dY_cat, dP_usr):
        """The step's weight gradients as llmrec_linear_wgrad_multi targets [(pairs, dW, db, accumulate)]: item_trans (5 attribute
        streams), user_trans, text_trans, image_trans. dY_cat = the [rows, 7 d] gradient of the projections' outputs (dU_cat when
        the operands are pre-propagated - its pairs carry the row weights of the bias gradient - else dP_cat)."""
        m = self.m
        if self.preprop:
            feats, roww = self.AX, self.a_rowsum
        else:
            feats, roww = [m.image_feats, m.text_feats] + [m.item_feats[key] for key in self.keys], None
        rows = (self.act_rows, self.act_n, self.act_expected) if (self.wgrad_rows and self.gemm == "bf16x3") else None
        item_pairs = [(self._side(dY_cat, 2 + k), feats[2 + k], roww, rows) for k in range(len(self.keys))]
        return [(item_pairs, m.item_trans.weight.grad, m.item_trans.bias.grad, False),
                ([(dP_usr, m.user_feats)], m.user_trans.weight.grad, m.user_trans.bias.grad, False),
                ([(self._side(dY_cat, 1), feats[1], roww)], m.text_trans.weight.grad, m.text_trans.bias.grad, False),
                ([(self._side(dY_cat, 0), feats[0], roww)], m.image_trans.weight.grad, m.image_trans.bias.grad, False)]

    def _project_all(self, which: str = "all"):
        """All 8 projections in one grouped launch (d <= 64), else one by one. which = "user" / "items": only user_trans' projection /
        only the seven item-side ones (LLMREC_SPLIT_PROJ: two launches, so that the profile chain can start behind the first)."""
        jobs = self.projection_jobs()
        if which != "all":
            jobs = [j for j in jobs if (j[2] is self.P_usr) == (which == "user")]
        if self.d > 64 or len(jobs) > _lib.CONST["LLMREC_LINEAR_MAX_PROBLEMS"]:
            if self.preprop:
                raise RuntimeError("FusedStep: the pre-propagated projection needs the grouped launch (d <= 64); set LLMREC_PREPROPAGATE=0")
            for X, lin, out, _ in jobs:
                self._linear(X, lin, out)
            return
        arr = (ops.LinearProblem * len(jobs))()
        for i, (X, lin, out, bscale) in enumerate(jobs):
            arr[i].X, arr[i].ldx, arr[i].M, arr[i].K = X.data_ptr(), _ld(X), X.shape[0], X.shape[1]
            arr[i].W, arr[i].ldw, arr[i].bias = lin.weight.data_ptr(), _ld(lin.weight), lin.bias.data_ptr()
            arr[i].Y, arr[i].ldy = out.data_ptr(), _ld(out)
            arr[i].bias_scale = bscale.data_ptr() if bscale is not None else None
        _call("llmrec_linear_fwd_grouped_bf16x3" if self.gemm == "bf16x3" else "llmrec_linear_fwd_grouped_f32", len(jobs), arr, self.d)

    def _wgrad(self, dY, X, lin, accumulate, ws=None):
        ops.linear_wgrad_grouped([(dY, X)], lin.weight.grad, lin.bias.grad, accumulate, self.ws_wgrad if ws is None else ws,
                                 precision=self.gemm)

    def _softmax(self, Z, Y):
        _call("llmrec_softmax_rows_fwd_f32", Z.shape[0], self.d, _p(Z), _ld(Z), _p(Y), _ld(Y))

    def _softmax_bwd(self, Y, dY, dZ, alpha: float = 1.0):
        _call("llmrec_softmax_rows_bwd_scaled_f32", Y.shape[0], self.d, float(alpha), _p(Y), _ld(Y), _p(dY), _ld(dY), _p(dZ), _ld(dZ))

    def _axpy(self, alpha, X, Y, accumulate, rows=None, cols=None):
        rows = X.shape[0] if rows is None else rows
        cols = X.shape[1] if cols is None else cols
        _call("llmrec_axpy_f32", rows, cols, float(alpha), None, _p(X), _ld(X), _p(Y), _ld(Y), 1 if accumulate else 0)

    @staticmethod
    def _tables(ts):
        return (_c.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), (_c.c_int64 * len(ts))(*[_ld(t) for t in ts])

    def _side(self, cat, s):
        return cat[:, s * self.d:(s + 1) * self.d]

    def _norm_terms(self, cat, prof):
        """[image, text, profile, attributes...] in the reference's order of addition (Models.py:188-197)."""
        return [self._side(cat, 0), self._side(cat, 1), prof] + [self._side(cat, 2 + k) for k in range(len(self.keys))]

    def _rates(self):
        r = [self.c_m, self.c_m, self.c_u] + [self.c_a] * len(self.keys)
        return (_c.c_float * len(r))(*r)

    # -- forward ----------------------------------------------------------------------------------
    def forward(self, sampler=None, after_chain=None, before_fusion=None):
        """sampler / after_chain: launches for the ID chain's stream, ahead of / behind its SpMMs (the batch and what is derived from it);
        before_fusion: called behind the join of the side chains, ahead of the fusion launch."""
        m, d = self.m, self.d
        self._fork(self.s2)
        # LLMREC_ID_FIRST=1 (experiment, profiles/experiments/r05_step_chain.md): the ID chain's SpMMs BEFORE the sampler and the row list on
        # this stream - behind them (one-block kernels that wait 50 - 100 us for a CU beside the projection) the chain ended as late as
        # the two other chains the fusion joins
        id_first = os.environ.get("LLMREC_ID_FIRST", "0") == "1"
        with self._on(self.s2):                                          # ID chain: needs no projection
            if sampler is not None and self.multi_stream and not id_first:   # the batch is first read by the losses, after the join below:
                sampler()                                                # sampling rides beside the projection instead of ahead of it
            if self._zero_in_forward and not self.fold:
                self.opt.advance()                                       # AdamW's step counter / bias corrections, off the critical path
            i_prev = m.item_id_embedding.weight
            for l in range(self.L):
                last = l == self.L - 1
                if last:                                                 # row softmax of the last layer = the SpMM's epilogue
                    sm = ops.spmm_epilogue(ops.EPI_SOFTMAX)
                    self._spmm(self.ui.fwd, i_prev, self.Ul[l], tag=2, epilogue=sm)
                    self._spmm(self.iu.fwd, self.Ul[l], self.Il[l], tag=2, epilogue=sm)
                else:
                    self._spmm(self.ui.fwd, i_prev, self.Ul[l], tag=2)
                    self._spmm(self.iu.fwd, self.Ul[l], self.Il[l], tag=2)
                i_prev = self.Il[l]
            if sampler is not None and self.multi_stream and id_first:
                sampler()
            self._ev_chain = self._mark() if after_chain is not None else None   # the chain's SpMMs are done; what follows on this stream
            if after_chain is not None:                                          # is not needed by the fusion
                after_chain()
        self._stamp(1)
        split_proj = os.environ.get("LLMREC_SPLIT_PROJ", "0") == "1" and self.multi_stream and self.d <= 64
        if split_proj:                                                   # user_trans first: the profile chain runs BESIDE the item-side projection
            self._project_all("user")                                    # and is long done when the fusion joins it (no late second parent)
            ev1 = self._mark()
            self._fork_from(ev1, self.s1)
            with self._on(self.s1):
                self._spmm(self.iu.fwd, self.P_usr, self.prof_i, tag=1)
                self._spmm(self.ui.fwd, self.prof_i, self.prof_u, tag=1)
            self._project_all("items")
        else:
            self._project_all()
        self._stamp(2)
        if not split_proj:
            ev1 = self._mark()
        if not self.preprop:
            self._spmm(self.ui.fwd, self.P_cat, self.U_cat)              # 7 streams, one adjacency pass
        self._spmm(self.iu.fwd, self.U_cat, self.I_cat)
        if not split_proj:
            if getattr(self, "profile_on_main", False):                  # (evaluation graphs: two branches instead of three, see eval_topk)
                self._spmm(self.iu.fwd, self.P_usr, self.prof_i, tag=1)
                self._spmm(self.ui.fwd, self.prof_i, self.prof_u, tag=1)
            else:
                self._fork_from(ev1, self.s1)
                with self._on(self.s1):                                  # profile stream: items first
                    self._spmm(self.iu.fwd, self.P_usr, self.prof_i, tag=1)
                    self._spmm(self.ui.fwd, self.prof_i, self.prof_u, tag=1)
        if self._zero_in_forward and not self.fold:                      # training: the feature regulariser's value needs only U_cat / I_cat -
            self._fork(self.s3)                                          # captured HERE it runs beside the fusion and the BPR launches (captured
            with self._on(self.s3):                                      # after the BPR backward, round 2, the graph ran it last: the step's tail)
                self._feat_reg()
        if getattr(self, "_ev_chain", None) is not None and self.multi_stream:
            self._join(self.s1)
            torch.cuda.current_stream().wait_event(self._ev_chain)               # (not the stream's tail: after_chain's launches are joined later)
        else:
            self._join(self.s1, self.s2)
        if before_fusion is not None:
            before_fusion()

        # E_u and E_i (Models.py:185-197) in ONE launch: llmrec_fuse_fwd_multi_f32 (two independent row ranges)
        keep = []

        def problem(pr, out, base, layers, cat, prof):
            means, norms = [base] + layers, self._norm_terms(cat, prof)
            mp, ml = self._tables(means)
            npt, nl = self._tables(norms)
            rates = self._rates()
            keep.extend((mp, ml, npt, nl, rates))
            pr.rows, pr.mean_scale, pr.n_mean, pr.n_norm = out.shape[0], 1.0 / len(means), len(means), len(norms)
            pr.mean_terms, pr.mean_ld = _c.cast(mp, _c.c_void_p), _c.cast(ml, _c.c_void_p)
            pr.norm_terms, pr.norm_ld, pr.rates = _c.cast(npt, _c.c_void_p), _c.cast(nl, _c.c_void_p), _c.cast(rates, _c.c_void_p)
            pr.out, pr.ldo = out.data_ptr(), _ld(out)
        arr = (ops.FuseFwdProblem * 2)()
        problem(arr[0], self.E_i, m.item_id_embedding.weight, self.Il, self.I_cat, self.prof_i)
        problem(arr[1], self.E_u, m.user_id_embedding.weight, self.Ul, self.U_cat, self.prof_u)
        if self.fold and self._zero_in_forward:                          # + the regulariser's sum of squares over the image / text streams (terms 0, 1)
            n_part = _c.c_int32(0)
            _call("llmrec_fuse_fwd_multi_sumsq_f32", 2, arr, d, 2, _p(self.ss_partial), self.ss_cap, _c.byref(n_part))
            self.ss_n = int(n_part.value)
        else:
            _call("llmrec_fuse_fwd_multi_f32", 2, arr, d)

    def outputs(self):
        """The reference's 14-tuple as views of the forward buffers (Models.py:199)."""
        att_i = {k: self._side(self.I_cat, 2 + j) for j, k in enumerate(self.keys)}
        att_u = {k: self._side(self.U_cat, 2 + j) for j, k in enumerate(self.keys)}
        return (self.E_u, self.E_i, self._side(self.I_cat, 0), self._side(self.I_cat, 1), self._side(self.U_cat, 0),
                self._side(self.U_cat, 1), self.P_usr, att_i, self.prof_u, self.prof_i, att_u, att_i, None, None)

    # -- losses + backward ------------------------------------------------------------------------
    def _problems(self):
        arr = (ops.BprProblem * self.n_prob)()
        tabs = [(self.E_u, self.E_i, self.dE_u, self.dE_i)]
        tU, tI, tP = self.sc_U, self.sc_I, self.sc_prof
        for s in range(2):
            tabs.append((self._side(self.U_cat, s), self._side(self.I_cat, s), self._side(tU, s), self._side(tI, s)))
        for k in range(len(self.keys)):
            tabs.append((self.prof_u, self._side(self.I_cat, 2 + k), tP, self._side(tI, 2 + k)))
        for i, (eu, ei, deu, dei) in enumerate(tabs):
            arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = eu.data_ptr(), _ld(eu), ei.data_ptr(), _ld(ei)
            arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = deu.data_ptr(), _ld(deu), dei.data_ptr(), _ld(dei)
            arr[i].g_mf, arr[i].g_emb = self.w_mf[i], self.w_emb[i]
        return arr

    def _bpr_launches(self, users, pos, neg, n_valid, lo: int = 0, hi: Optional[int] = None):
        """scores -> selection -> backward rows of the problems [lo, hi) (all: the step's three loss launches). The launch that contains
        problem 0 begins the step (row stamp, AdamW's counter) and stamps the batch's rows."""
        hp, B = self.hp, users.numel()
        hi = self.n_prob if hi is None else hi
        full = self._problems()
        n = hi - lo
        probs = full if (lo == 0 and hi == self.n_prob) else (ops.BprProblem * n)(*[full[i] for i in range(lo, hi)])
        saved = self.saved if lo == 0 else self.saved[lo * ops.bpr_saved_floats(B):]
        remember = float(1 - hp.prune_loss_drop_rate)
        first = lo == 0
        if self.fold and first:                                          # the scores launch begins the step: row stamp + AdamW's counter
            o = self.opt
            if o.dev_state is None:
                o.dev_state = torch.zeros(3, dtype=torch.float32, device=self.E_u.device)
            _call("llmrec_bpr_multi_scores_step_f32", n, probs, self.d, _p(users), _p(pos), _p(neg), B, _p(n_valid), _p(saved),
                  _p(self.row_stamp), _p(o.dev_state), o.lr, o.betas[0], o.betas[1])
        else:
            _call("llmrec_bpr_multi_scores_f32", n, probs, self.d, _p(users), _p(pos), _p(neg), B, _p(n_valid), _p(saved),
                  _p(self.row_stamp) if first else None)
        _call("llmrec_bpr_multi_select_bwd_f32", n, probs, self.d, _p(users), _p(pos), _p(neg), B, _p(n_valid), remember,
              float(hp.decay), float(hp.batch_size), _p(saved), _p(self.flag_u) if first else None, _p(self.flag_i) if first else None,
              _p(self.row_stamp), _p(self.bpr_plan))

    def loss_backward(self, users, pos, neg, n_valid=None):
        B = users.numel()
        if B > self.b_max:
            raise RuntimeError("FusedStep: batch of %d exceeds b_max %d" % (B, self.b_max))
        probs = self._problems()
        hp = self.hp
        remember = float(1 - hp.prune_loss_drop_rate)
        # critical path: scores -> [selection + gradient rows] (two launches); the loss VALUES (one more launch) and their assembly for
        # the log line ride on the ID chain's stream (a branch of their own right behind the BPR launches: the graph ran it as the step's tail)
        self._check_scatter_targets()
        if getattr(self, "_ev_plan", None) is not None:                  # (LLMREC_PLAN_STREAM=late: the plan was built behind the ID chain)
            torch.cuda.current_stream().wait_event(self._ev_plan)
        if getattr(self, "_bpr_split_done", False):                      # (LLMREC_BPR_SPLIT: problems 1.. were launched beside the fusion)
            self._bpr_launches(users, pos, neg, n_valid, lo=0, hi=1)
            self._join(self.s1)
            self._bpr_split_done = False
        else:
            self._bpr_launches(users, pos, neg, n_valid)

        def side():                              # (runs on the ID chain's stream, _backward places it)
            if self.fold:                        # loss values + regulariser (the fusion launch's partial sums) + assembly: one launch
                w = (_c.c_float * self.n_prob)(*self.w_mf)
                _call("llmrec_bpr_multi_losses_assemble_f32", self.n_prob, B, _p(n_valid), remember, float(hp.decay), float(hp.batch_size),
                      _p(self.out), _p(self.saved), w, _p(self.ss_partial), self.ss_n, float(hp.feat_reg_decay * 0.5 / self.I),
                      _p(self.scal), _p(self.epoch_sums))
                return
            _call("llmrec_bpr_multi_losses_f32", self.n_prob, B, _p(n_valid), remember, float(hp.decay), float(hp.batch_size),
                  _p(self.out), _p(self.saved))
            self._join(self.s3)                  # the regulariser's value (s3, during the forward)
            self._assemble_loss(0)               # loss = sum_p w_mf[p] * mf_p + emb_0 + feat_reg
        self._backward(probs, users, pos, neg, n_valid, side_work=side, bpr_bwd_done=True)
        if not self.fold:
            self._join(self.s3)

    def _assemble_loss(self, mode: int, tail=None, inv_world: float = 1.0):
        """Logged scalars (main.py:273,280-283) from the 8 BPR results + the regulariser: one single-wave launch."""
        w = (_c.c_float * self.n_prob)(*self.w_mf)
        _call("llmrec_loss_assemble_f32", mode, self.n_prob, _p(self.out), w, _p(self.scal), _p(tail), float(inv_world),
              _p(self.epoch_sums) if mode != 1 else None)

    def _feat_reg(self):
        """Feature regulariser (main.py:151-156) over the image/text columns of both cat buffers -> scal[0]."""
        coef = self.hp.feat_reg_decay * 0.5 / self.I
        for k, blk in enumerate((self.I_cat, self.U_cat)):
            _call("llmrec_sumsq_f32", blk.shape[0], 2 * self.d, _p(blk), _ld(blk), float(coef), k, _p(self.scal), _p(self.ws_sumsq),
                  self.ws_sumsq.numel())

    def _check_scatter_targets(self):
        if self.check_zero and not torch.cuda.is_current_stream_capturing():
            dirty = [n for n, t in (("dE_u", self.dE_u), ("dE_i", self.dE_i), ("sc_U", self.sc_U), ("sc_I", self.sc_I), ("sc_prof", self.sc_prof))
                     if float(t.abs().max()) != 0.0]
            if dirty:
                raise RuntimeError("FusedStep: scatter targets not all-zero before the loss backward: %s (an aborted step? call reset_scatter_targets())" % dirty)

    def _backward(self, probs, users, pos, neg, n_valid, replicated_scale: float = 1.0, after_first=None, bpr_bwd_done: bool = False,
                  side_work=None):
        """Hand-written backward from the saved BPR state to the parameter gradients (and, with inline_adamw, the update).
        replicated_scale weights the batch-independent loss terms (1 / world on batch-sharded replicas, whose
        gradients are summed over ranks afterwards). bpr_bwd_done: the loss launch already scattered the gradient rows.
        after_first: side work captured right after the BPR backward; side_work: launches for the ID chain's stream, ahead of its last SpMM."""
        hp, d, L, S = self.hp, self.d, self.L, self.S
        B = users.numel()
        coef = hp.feat_reg_decay * 0.5 / self.I * replicated_scale
        if not bpr_bwd_done:
            self._check_scatter_targets()
            _call("llmrec_bpr_multi_bwd_f32", self.n_prob, probs, d, _p(users), _p(pos), _p(neg), B, _p(n_valid), float(hp.decay),
                  float(hp.batch_size), _p(self.saved), _p(self.bpr_plan))
        ev_rows = self._mark()                                           # dE_u / dE_i hold the scattered rows: all the ID chain needs
        if after_first is not None:
            after_first()

        # the backward of both fusions in ONE launch (llmrec_fuse_bwd_src_multi_f32)
        keep = []

        def problem(pr, dout, cat, prof, dcat, dprof, srcs, flags):
            norms, dnorms = self._norm_terms(cat, prof), self._norm_terms(dcat, dprof)
            npt, nl = self._tables(norms)
            dp, dl = self._tables(dnorms)
            # sources = what the loss backward scattered for this side (terms in _norm_terms order: image, text, profile, attributes);
            # the feature regulariser's gradient on the image / text streams (terms 0, 1) rides along: 2 coef x
            sp = (_c.c_void_p * len(srcs))(*[t.data_ptr() if t is not None else None for t in srcs])
            sl = (_c.c_int64 * len(srcs))(*[_ld(t) if t is not None else 0 for t in srcs])
            rates = self._rates()
            keep.extend((npt, nl, dp, dl, sp, sl, rates))
            pr.rows, pr.dOut, pr.lddo, pr.n_norm = dout.shape[0], dout.data_ptr(), _ld(dout), len(norms)
            pr.norm_terms, pr.norm_ld, pr.rates = _c.cast(npt, _c.c_void_p), _c.cast(nl, _c.c_void_p), _c.cast(rates, _c.c_void_p)
            pr.d_terms, pr.d_ld = _c.cast(dp, _c.c_void_p), _c.cast(dl, _c.c_void_p)
            pr.src_terms, pr.src_ld = _c.cast(sp, _c.c_void_p), _c.cast(sl, _c.c_void_p)
            pr.n_reg_terms, pr.reg_two_coef = 2, float(2.0 * coef)
            if flags is not None and bpr_bwd_done:                      # (only the fused loss launch stamps the rows)
                pr.row_flags, pr.row_stamp = flags.data_ptr(), self.row_stamp.data_ptr()
        arr = (ops.FuseBwdProblem * 2)()
        problem(arr[0], self.dE_i, self.I_cat, self.prof_i, self.dI_cat, self.dprof_i,
                [self._side(self.sc_I, 0), self._side(self.sc_I, 1), None] + [self._side(self.sc_I, 2 + k) for k in range(len(self.keys))], self.flag_i)
        problem(arr[1], self.dE_u, self.U_cat, self.prof_u, self.dU_cat, self.dprof_u,
                [self._side(self.sc_U, 0), self._side(self.sc_U, 1), self.sc_prof] + [None] * len(self.keys), self.flag_u)
        _call("llmrec_fuse_bwd_src_multi_f32", 2, arr, d)
        ev_fuse = self._mark()                                           # the fusion backward has read dE_u / dE_i
        m = self.m
        inv = 1.0 / (L + 1)
        # capture order at this fork (experiment knobs, profiles/experiments/r05_step_chain.md): LLMREC_BWD_MAIN_FIRST=1 captures the critical
        # transposed side product BEFORE the two side branches; LLMREC_LOSS_STREAM=s3 puts the logged-scalar launch on its own stream
        main_first = os.environ.get("LLMREC_BWD_MAIN_FIRST", "1" if self.fold else "0") == "1" and self.multi_stream
        loss_s3 = os.environ.get("LLMREC_LOSS_STREAM", "s3" if self.fold else "s2") == "s3" and self.multi_stream and side_work is not None
        if main_first:
            self._spmm(self.iu.bwd, self.dI_cat, self.dU_cat, accumulate=True)
            if not self.preprop:
                self._spmm(self.ui.bwd, self.dU_cat, self.dP_cat)
            self._fork_from(ev_fuse, self.s1)
        else:
            self._fork(self.s1)
        if loss_s3:
            self._fork_from(ev_rows, self.s3)
            with self._on(self.s3):
                side_work()
            side_work = None
        with self._on(self.s1):
            # profile chain: prof_u = ui(prof_i), prof_i = iu(P_usr); user_trans' weight gradient joins the item-side ones below
            self._spmm(self.ui.bwd, self.dprof_u, self.dprof_i, accumulate=True, tag=1)
            self._spmm(self.iu.bwd, self.dprof_i, self.dP_usr, tag=1)
            if self.gemm != "bf16x3":
                self._wgrad(self.dP_usr, m.user_feats, m.user_trans, False, ws=self.ws_wgrad_b)

        def id_chain():
            # ID chain (items of layer l+1 from the new users; softmax on the last layer)
            # every "+ mean term" and every softmax backward below is an epilogue of the SpMM that produces the tensor:
            #   dI[L] = inv dE_i                      -> g = softmax_bwd(I_L, dI[L])            (one row kernel, no SpMM feeds it)
            #   dU[l+1] = inv dE_u + A_iu^T g         -> h = softmax_bwd(U_L, dU[l+1]) on the last layer
            #   dI[l]   = inv dE_i + A_ui^T h         -> (l > 0) feeds the next round as g; (l = 0) IS the item table's gradient
            # U^0 only enters the mean: the user table's gradient needs nothing else. It and the logged loss values go first: whatever
            # this stream still has queued when the weight-gradient GEMM takes every CU (about when the chain's last SpMM starts)
            # waits for the GEMM's blocks to retire and becomes the step's tail.
            if self.fold:                                                         # AdamW reads inv * dE_u and stores it as the table's .grad
                self.opt.step_params(self._emb_params[:1], sources={self._emb_params[0]: (self.dE_u, inv)})
            else:
                self._axpy(inv, self.dE_u, m.user_id_embedding.weight.grad, False)
                if self.inline_adamw:                                             # U^0 only enters the mean: the user table's gradient is final here
                    self.opt.step_params(self._emb_params[:1])
            if side_work is not None:
                side_work()
            g = self.bufI
            if L >= 1:                                                                    # dI[L] = inv dE_i (mean part), g = softmax_bwd(I_L, dI[L])
                self._softmax_bwd(self.Il[L - 1], self.dE_i, self.tmpI, alpha=inv); g = self.tmpI
            for l in range(L - 1, -1, -1):
                last = l == L - 1
                if last:
                    self._spmm(self.iu.bwd, g, self.tmpU, tag=2,
                               epilogue=ops.spmm_epilogue(ops.EPI_SOFTMAX_BWD, inv, self.dE_u, self.Ul[l]))   # h
                    h = self.tmpU
                else:
                    self._spmm(self.iu.bwd, g, self.bufU, tag=2, epilogue=ops.spmm_epilogue(ops.EPI_NONE, inv, self.dE_u))
                    h = self.bufU
                dst = m.item_id_embedding.weight.grad if l == 0 else self.bufI
                self._spmm(self.ui.bwd, h, dst, tag=2, epilogue=ops.spmm_epilogue(ops.EPI_NONE, inv, self.dE_i))
                g = self.bufI
            if L == 0:
                self._axpy(inv, self.dE_i, m.item_id_embedding.weight.grad, False)

            # after the last reader of dE_u / dE_i (this chain and the fusion backward): clear the touched rows. (The wait below is for
            # an EARLY event. A wait for a late one anywhere in this stream's chain - e.g. for the transposed side product, had the row
            # stamps needed a clean-up - makes the graph runtime start the WHOLE chain late: profiles/experiments/r03_wgrad.md, last table.)
            if ev_fuse is not None:
                torch.cuda.current_stream().wait_event(ev_fuse)
            if self.fold and L > 0:                                               # clean-up + the item table's AdamW in ONE launch
                jobs = [(users, self.dE_u), (pos, self.dE_i), (neg, self.dE_i), (users, self.sc_U), (pos, self.sc_I), (neg, self.sc_I),
                        (users, self.sc_prof)]
                self.opt.step_params(self._emb_params[1:], zero_rows=(jobs, B, n_valid))
            else:
                _call("llmrec_bpr_multi_zero_rows_f32", self.n_prob, probs, d, _p(users), _p(pos), _p(neg), B, _p(n_valid))
                if self.inline_adamw:                                             # the item table's gradient is final: update it here,
                    self.opt.step_params(self._emb_params[1:])                    # beside the weight-gradient GEMM

        # the ID chain depends on the BPR rows only, not on the fusion backward: it starts beside it (captured after it, so that the
        # fusion backward stays the graph's same-queue successor of the BPR launch) and has most of its SpMMs behind it when the
        # weight-gradient GEMM takes every CU (one 512-register wave per SIMD leaves no room for a second kernel)
        if ev_rows is not None:
            self._fork_from(ev_rows, self.s2)
        with self._on(self.s2):
            id_chain()
        # side chain: I_cat = iu(U_cat), U_cat = ui(P_cat) (or the pre-propagated projection); then the item-side weight gradients
        if not main_first:
            self._spmm(self.iu.bwd, self.dI_cat, self.dU_cat, accumulate=True)
            if not self.preprop:
                self._spmm(self.ui.bwd, self.dU_cat, self.dP_cat)
        dY_cat = self.dU_cat if self.preprop else self.dP_cat
        targets = self.wgrad_targets(dY_cat, self.dP_usr)
        item_pairs, text_pairs, image_pairs = targets[0][0], targets[2][0], targets[3][0]
        done, updated = False, []
        if self.gemm == "bf16x3":
            # ONE launch for all four Linears: equal slabs over all of them, so the launch is whole rounds of equal blocks and the
            # short gradients pay no ramp-up / ragged last round of their own (round 2, same box: three launches back to back 0.661 ms
            # per step, one launch 0.637 ms). user_trans' gradient is in the same launch since round 3: with the pre-propagated
            # operands nothing separates the two launches in time any more, and side by side each ran at half speed (the chip is
            # power-bound here: profiles/experiments/r03_wgrad.md). The bias gradients (row-weighted when pre-propagated) come out of it too.
            split_wgrad = os.environ.get("LLMREC_SPLIT_WGRAD", "0") == "1" and self.multi_stream and self.inline_adamw
            if not split_wgrad:
                self._join(self.s1)                                      # dP_usr (the profile chain is long done by now)
            if self.ws_wgrad_multi is None:
                need = ops.linear_wgrad_multi_workspace(targets, self.wgrad_blocks)
                self.ws_wgrad_multi = torch.empty(max(need, 0), dtype=torch.uint8, device=dY_cat.device) if need >= 0 else False
            if self.ws_wgrad_multi is not False:
                self._stamp(3)
                if getattr(self, "_ev_reach", None) is not None:                 # (LLMREC_REACH_LATE: the row list was built behind the ID chain)
                    torch.cuda.current_stream().wait_event(self._ev_reach)
                lins = (m.item_trans, m.user_trans, m.text_trans, m.image_trans)              # (wgrad_targets' order)
                if split_wgrad:
                    # the three item-side Linears first - their only late parent is the transposed side product on THIS stream - then,
                    # behind the join with the profile chain, user_trans' own launch (two GEMM + two reduction launches instead of one + one)
                    sel = [0, 2, 3]
                    ops.linear_wgrad_multi([targets[i] for i in sel], self.ws_wgrad_multi, update=(self.opt, [(lins[i].weight, lins[i].bias) for i in sel]),
                                           block_budget=self.wgrad_blocks)
                    self._join(self.s1)
                    if getattr(self, "ws_wgrad_user", None) is None:
                        self.ws_wgrad_user = torch.empty(max(ops.linear_wgrad_multi_workspace(targets[1:2], self.wgrad_blocks), 16), dtype=torch.uint8, device=dY_cat.device)
                    ops.linear_wgrad_multi(targets[1:2], self.ws_wgrad_user, update=(self.opt, [(lins[1].weight, lins[1].bias)]), block_budget=self.wgrad_blocks)
                    updated = [p_ for l in lins for p_ in (l.weight, l.bias)]
                elif self.inline_adamw:                                  # the four Linears' AdamW rides in the slab-reduction launch
                    ops.linear_wgrad_multi(targets, self.ws_wgrad_multi, update=(self.opt, [(l.weight, l.bias) for l in lins]), block_budget=self.wgrad_blocks)
                    updated = [p_ for l in lins for p_ in (l.weight, l.bias)]
                else:
                    ops.linear_wgrad_multi(targets, self.ws_wgrad_multi, block_budget=self.wgrad_blocks)
                self._stamp(4)
                done = True
            else:                                                        # outside the multi-target fast path: user_trans' on its own
                self._wgrad(self.dP_usr, m.user_feats, m.user_trans, False, ws=self.ws_wgrad_b)
        if not done:
            # exact-fp32 GEMMs, or shapes outside the multi-target fast path (N != 64, K % 128 != 0): one launch per Linear; their
            # kernels sum dY unweighted, so a pre-propagated step takes its bias gradients from llmrec_weighted_colsum_f32
            strip = lambda pairs: [p_[:2] for p_ in pairs]
            bias = (lambda lin: None) if self.preprop else (lambda lin: lin.bias.grad)
            if self.preprop:
                outs = [m.image_trans.bias.grad, m.text_trans.bias.grad] + [m.item_trans.bias.grad] * len(self.keys)
                gp = (_c.c_void_p * S)(*[t.data_ptr() for t in outs])
                _call("llmrec_weighted_colsum_f32", self.U, S, d, _p(dY_cat), _ld(dY_cat), _p(self.a_rowsum), gp, 0, _p(self.ws_colsum), self.ws_colsum.numel())
            ops.linear_wgrad_grouped(strip(item_pairs), m.item_trans.weight.grad, bias(m.item_trans), False, self.ws_wgrad, precision=self.gemm)
            ops.linear_wgrad_grouped(strip(text_pairs), m.text_trans.weight.grad, bias(m.text_trans), False, self.ws_wgrad_c, precision=self.gemm)
            ops.linear_wgrad_grouped(strip(image_pairs), m.image_trans.weight.grad, bias(m.image_trans), False, self.ws_wgrad_d, precision=self.gemm)
        self._join(self.s1)                                              # user_trans' gradient
        if self.inline_adamw:                                            # whatever the reduction launch did not update (all four Linears on
            self.opt.step_params([p_ for p_ in self._lin_params if all(p_ is not q for q in updated)])   # the per-target fallback paths)
        self._join(self.s2)
        if loss_s3:
            self._join(self.s3)

    def _train_forward(self, sampler=None, after_chain=None, before_fusion=None):
        """forward() of a training step: also advances AdamW's counters (and samples the batch) on a side stream."""
        self._zero_in_forward = True
        try:
            self.forward(sampler, after_chain, before_fusion)
        finally:
            self._zero_in_forward = False

    def build_scatter_plan(self, users, pos, neg, n_valid=None):
        """llmrec_bpr_scatter_plan of this batch into the step's plan buffer (one launch; step_eager does it behind the sampler - callers
        that drive forward() / loss_backward() themselves call it once per batch, any time before the loss backward)."""
        B = users.numel()
        if B > self.b_max:
            raise RuntimeError("FusedStep: batch of %d exceeds b_max %d" % (B, self.b_max))
        _call("llmrec_bpr_scatter_plan", _p(users), _p(pos), _p(neg), B, _p(n_valid), _p(self.bpr_plan))

    def reset_scatter_targets(self):
        """Dense clear of the buffers the sparse-zero scheme keeps all-zero between steps (set-up, (re)capture, and after a
        step that raised between the loss backward's scatter and its row-wise clean-up)."""
        for t in (self.dE_u, self.dE_i, self.sc_U, self.sc_I, self.sc_prof):
            t.zero_()

    def step_eager(self, users, pos, neg, n_valid=None, sampler=None):
        """sampler: optional callable that fills (users, pos, neg, n_valid) on the current stream first (inside the same
        graph when captured; running it on a side stream beside the forward measured no faster)."""
        side = self.multi_stream                                 # the sampler rides beside the projection (forward())
        # (Built on the regulariser's stream instead - forked from the main stream, waiting for the sampler's event of the ID chain's stream,
        #  its own event awaited by the weight gradient - hipGraphInstantiate of this image recursed until the stack ran out: not kept.)
        # The scatter plan of the loss backward (needed ~200 us later) and the row list (needed by the weight gradient only, at the far end of
        # the step) ride on the ID chain's stream BEHIND its SpMMs; the fusion waits for the chain's event (self._ev_chain), the loss launches
        # for the plan's (self._ev_plan), the weight gradient for the list's (self._ev_reach). Measured on one box, 300 steps each, twice
        # (round 6, profiles/experiments/r06_step_chain.md): both right behind the sampler, ahead of the SpMMs (LLMREC_PLAN_STREAM=early
        # LLMREC_REACH_LATE=0, the state until then) 0.4539 / 0.4539 ms per step; here 0.4470 / 0.4481. With the plan behind the chain but the
        # fusion joining the whole stream (the first form of "late") 0.460 / 0.461; the plan on a branch of its own forked from the sampler
        # 0.534 / 0.534 - the graph runtime then runs the PROJECTION behind the ID chain (not kept).
        fill = sampler
        plan_late = self.multi_stream and os.environ.get("LLMREC_PLAN_STREAM", "late") == "late"
        reach_late = self.multi_stream and self.wgrad_rows and os.environ.get("LLMREC_REACH_LATE", "1") == "1"
        self._ev_reach = None

        def reach():
            ops.batch_reach_rows(users, pos, neg, n_valid, self.iu.fwd, self.act_flags, self.act_rows, self.act_n)

        def sampler():
            if fill is not None:
                fill()
            if not plan_late:
                self.build_scatter_plan(users, pos, neg, n_valid)
            if self.wgrad_rows and not reach_late:
                reach()

        self._ev_plan = None

        def late_work():
            if plan_late:
                self.build_scatter_plan(users, pos, neg, n_valid)
                self._ev_plan = self._mark()                             # (the loss launches wait for this event, the fusion only for the chain's)
            if reach_late:
                reach()
                self._ev_reach = self._mark()
        late = late_work if (plan_late or reach_late) else None
        self.spmm_edge_units = 0.0
        calls0 = _lib.n_calls
        try:
            self._stamp(0)
            if sampler is not None and not side:
                sampler()
            # LLMREC_BPR_SPLIT=1 (experiment, profiles/experiments/r06_step_chain.md): the seven side problems' loss launches (their tables are
            # complete BEFORE the fusion) on the profile stream beside the fusion; only problem 0 (E_u / E_i) stays on the critical path
            split = (os.environ.get("LLMREC_BPR_SPLIT", "0") == "1" and self.fold and self.multi_stream and self.n_prob > 1)
            self._bpr_split_done = False

            def side_problems():
                self._fork_from(self._mark(), self.s1)
                with self._on(self.s1):
                    self._bpr_launches(users, pos, neg, n_valid, lo=1)
                self._bpr_split_done = True
            self._train_forward(sampler if side else None, late, side_problems if split else None)
            self.loss_backward(users, pos, neg, n_valid)
            if not self.inline_adamw:
                self.opt.step(advanced=True)
            self._stamp(5)
            self.entry_point_calls_per_step = _lib.n_calls - calls0 - (2 if self.stamps is not None else 0) * 3
            if self.wgrad_rows and self.act_expected is None and not torch.cuda.is_current_stream_capturing():
                self._size_wgrad_for_rows()                      # first step: one read-back of the list length
        except Exception:
            if not torch.cuda.is_current_stream_capturing():   # the invariant of LLMREC_SPARSE_ZERO may be broken: restore it
                try:
                    self.reset_scatter_targets()
                except Exception:
                    pass
            raise
        return self.scal[1], self.scal[2], self.scal[3]

    def _size_wgrad_for_rows(self):
        """Lay the weight-gradient launch out for the list length the last step saw (+ 10 %): the kernel cuts a listed problem's slabs
        into equal pieces of the ACTUAL length every step, this only decides how many blocks each target gets. One host read-back;
        the workspace is re-sized here (outside any capture)."""
        n = int(self.act_n.item())
        self.act_expected = max(64, min(self.U, int(n * 1.10) + 32))
        need = ops.linear_wgrad_multi_workspace(self.wgrad_targets(self.dU_cat if self.preprop else self.dP_cat, self.dP_usr), self.wgrad_blocks)
        self.ws_wgrad_multi = torch.empty(max(need, 0), dtype=torch.uint8, device=self.dP_usr.device) if need >= 0 else False

    def check_wgrad_geometry(self, factor: float = 1.5) -> bool:
        """the row-listed weight gradient's launch geometry is laid out for the list length of the FIRST step (+ 10 %) and frozen
        into the captured graph; the kernel re-cuts its slabs for the actual length, so results never depend on it, but a workload whose
        batches later reach many more (or far fewer) rows would run in a badly balanced launch with no signal. Call this where the host
        synchronises anyway (main.py: once per epoch, behind the epoch sums): ONE read-back of the last step's list length; when it left
        [expected / factor, expected * factor] the launch is laid out again and the step graph RE-captured (no warm-up step: nothing is
        trained here). Returns whether it re-captured."""
        if not self.wgrad_rows or self.act_expected is None or torch.cuda.is_current_stream_capturing():
            return False
        want = max(64, min(self.U, int(int(self.act_n.item()) * 1.10) + 32))
        if self.act_expected / factor <= want <= self.act_expected * factor:
            return False
        self._size_wgrad_for_rows()
        if self.graph_exec is not None and getattr(self, "_capture_args", None) is not None:
            self.capture(warm=False, **self._capture_args)
        return True

    def flush(self):
        """Nothing is deferred in the single-graph step (DataParallelStep defers its AdamW)."""

    # -- evaluation -------------------------------------------------------------------------------
    def eval_topk(self, query_users: torch.Tensor, train: Optional[ops.Csr], K: int, use_graph: bool = False, held=None, Ks=None):
        """Reference Trainer.test up to the ranked lists (main.py:297-303, batch_test.py:83-109): no-grad forward +
        scoring + masked top-K for the listed users -> (idx int32 [n, K], scores). With use_graph the whole
        evaluation (~40 launches) is one HIP graph per (query set, K), replayed at every epoch end.
        held = (rowptr, colidx) of the held-out CSR + Ks (use_graph only): the evaluation also ends with llmrec_topk_eval_sums - hits, per-user
        precision / recall / ndcg / hit-ratio and their sums over the users, two launches at the graph's tail writing the 4 x len(Ks)
        doubles into PINNED host memory (self.eval_sums(...) after a stream synchronisation; batch_test.py:160-165 divided by the
        number of users) - nothing per user leaves the device and no copy follows the replay."""
        if not use_graph:
            self.forward()
            return ops.score_topk(self.E_u, self.E_i, query_users, train, K)
        Ks = tuple(int(k) for k in Ks) if held is not None else None
        key = (query_users.data_ptr(), query_users.numel(), K, id(train), None if held is None else (held[0].data_ptr(), Ks))
        ev = self._eval_graphs.get(key)
        if ev is None:
            q = query_users.to(torch.int64).contiguous()
            n = q.numel()
            idx = torch.empty(n, K, dtype=torch.int32, device=q.device)
            sc = torch.empty(n, K, dtype=torch.float32, device=q.device)
            ws = ops.topk_workspace(n, self.I, q.device, self.d)
            sums = sums_ws = None
            if held is not None:
                sums = torch.zeros(4, len(Ks), dtype=torch.float64).pin_memory()
                sums_ws = torch.empty(_lib.query("llmrec_topk_eval_sums_workspace_bytes", n, len(Ks)), dtype=torch.uint8, device=q.device)

            def run():
                # the evaluation's graph has TWO branches (the ID chain beside the projection; the profile chain stays on the main stream):
                # 0.572 ms per replay back to back / 0.599 one at a time, against 0.73 - 0.76 / 0.595 with the training forward's three
                # side branches - successive launches of a many-branch graph pay for their cross-queue joins (tools/eval_probe.py, round 6)
                branches, ms = os.environ.get("LLMREC_EVAL_BRANCHES", "2"), self.multi_stream
                self.profile_on_main = branches == "2"
                self.multi_stream = ms and branches != "1"
                try:
                    self.forward()
                finally:
                    self.profile_on_main, self.multi_stream = False, ms
                _call("llmrec_score_topk_mode_f32", n, _p(q), _p(self.E_u), _ld(self.E_u), _p(self.E_i), _ld(self.E_i), self.I, self.d,
                      _p(train.rowptr) if train is not None else None, _p(train.colidx) if train is not None else None,
                      K, _p(idx), _p(sc), _p(ws), ws.numel() if ws is not None else 0, ops.topk_mode(None, self.I, self.d, K))
                if held is not None:
                    ops.topk_eval_sums(idx, q, held[0], held[1], Ks, out=sums, ws=sums_ws)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with _capture_without_gc(g):
                run()
            ev = self._eval_graphs[key] = (g, idx, sc, q, train, ws, sums, sums_ws, held)     # keeps the captured operands alive
        ev[0].replay()
        self._last_eval = ev
        return ev[1], ev[2]

    def eval_sums(self):
        """The pinned [4, len(Ks)] sums of the LAST eval_topk(..., held=...) replay; synchronises the current stream first."""
        ev = getattr(self, "_last_eval", None)
        if ev is None or ev[6] is None:
            raise RuntimeError("FusedStep.eval_sums: the last evaluation was not captured with a held-out set")
        torch.cuda.current_stream().synchronize()
        return ev[6]

    def drop_eval_graph(self, query_users: torch.Tensor):
        """Release every captured evaluation (graph, result lists, top-K workspace) of this query tensor."""
        for key in [k for k in self._eval_graphs if k[0] == query_users.data_ptr() and k[1] == query_users.numel()]:
            del self._eval_graphs[key]

    # -- HIP graph --------------------------------------------------------------------------------
    def _make_static(self):
        dev = self.E_u.device
        # ONE int64 block [users | pos | neg | n_valid] so that a host-sampled batch can arrive in a single asynchronous H2D copy
        # (step_packed): b_max ids each, then one slot whose LOW 32 bits are the int32 n_valid the kernels read (little-endian)
        b = self.b_max
        blk = torch.zeros(3 * b + 1, dtype=torch.int64, device=dev)
        self.static_block = blk
        self.static = {"users": blk[0:b], "pos": blk[b:2 * b], "neg": blk[2 * b:3 * b], "n_valid": blk[3 * b:3 * b + 1].view(torch.int32)[:1]}
        return self.static

    def capture(self, warm_users=None, warm_pos=None, warm_neg=None, warm_n_valid=None, batcher=None, unroll: int = 1, warm: bool = True):
        """Capture one step (fixed batch capacity b_max, actual size on the device in n_valid).
        With `batcher` (engine.DeviceBatcher, capacity == b_max) the sampler is part of the graph: a
        training step is then ``step()`` with no arguments = one graph replay, nothing else on the stream.
        warm=False (a RE-capture, check_wgrad_geometry): no warm-up step is run - it would be an extra optimiser step - and the static
        batch buffers are kept."""
        st = self._make_static() if (warm or self.static is None) else self.static
        self.batcher = batcher
        self._capture_args = {"batcher": batcher, "unroll": unroll}
        if batcher is not None:
            if batcher.capacity != self.b_max:
                raise RuntimeError("FusedStep.capture: batcher capacity %d != b_max %d" % (batcher.capacity, self.b_max))
        elif warm:
            self._load(warm_users, warm_pos, warm_neg, warm_n_valid)

        def one_step():
            fill = (lambda: batcher.fill(st["users"], st["pos"], st["neg"], st["n_valid"])) if batcher is not None else None
            self.step_eager(st["users"], st["pos"], st["neg"], st["n_valid"], sampler=fill)
        self.reset_scatter_targets()                           # (re)capture starts from the invariant, whatever ran before
        if warm:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                         # warm-up on a side stream (allocations, plan caches)
                one_step()
            torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with _capture_without_gc(g):
            one_step()
        self.graph_exec = g
        # with the sampler inside the graph nothing host-side separates two steps: run_steps() replays a graph of `unroll` steps (the
        # device front end takes ~10 us to start the next graph behind the previous one's last kernel - paid once per `unroll` steps)
        self.graph_multi, self.graph_unroll = None, 0
        if batcher is not None and unroll > 1:
            g2 = torch.cuda.CUDAGraph()
            with _capture_without_gc(g2):
                for _ in range(unroll):
                    one_step()
            self.graph_multi, self.graph_unroll = g2, unroll

    def run_steps(self, n: int):
        """n training steps of the captured graph(s) (in-graph sampler only): as many replays of the `unroll`-step graph as fit, single
        steps for the remainder. Returns the logged scalars of the last step."""
        if self.graph_exec is None or getattr(self, "batcher", None) is None:
            raise RuntimeError("FusedStep.run_steps: capture(batcher=...) first")
        k = self.graph_unroll if self.graph_multi is not None else 0
        while k and n >= k:
            self.graph_multi.replay()
            n -= k
        for _ in range(n):
            self.graph_exec.replay()
        return self.scal[1], self.scal[2], self.scal[3]

    def _load(self, users, pos, neg, n_valid):
        st, B = self.static, users.numel()
        if B > self.b_max:
            raise RuntimeError("FusedStep: batch of %d exceeds b_max %d" % (B, self.b_max))
        st["users"][:B].copy_(users); st["pos"][:B].copy_(pos); st["neg"][:B].copy_(neg)
        if n_valid is None:
            st["n_valid"].fill_(B)
        else:
            st["n_valid"].copy_(n_valid)

    def packed_layout(self):
        """(slots, b_max): a batch for step_packed is an int64 block of `slots` = 3 b_max + 1 entries - users at [0, B'), positives at
        [b_max, b_max + B'), negatives at [2 b_max, 2 b_max + B'), the number of triples B' in the last slot."""
        return 3 * self.b_max + 1, self.b_max

    def step_packed(self, host_block: torch.Tensor):
        """One training step from a host-sampled batch already laid out as packed_layout() says (pinned memory): ONE asynchronous H2D copy
        into the captured step's static buffers, then the graph replay - no per-tensor device copies (step(users, pos, neg) issues four)."""
        if self.graph_exec is None or getattr(self, "batcher", None) is not None:
            raise RuntimeError("FusedStep.step_packed: capture(users, pos, neg) first (a graph without an in-graph sampler)")
        self.static_block.copy_(host_block, non_blocking=True)
        self.graph_exec.replay()
        return self.scal[1], self.scal[2], self.scal[3]

    def step(self, users=None, pos=None, neg=None, n_valid=None):
        """One training step; replays the captured graph when there is one (no arguments when the
        sampler was captured with it)."""
        if self.graph_exec is None:
            return self.step_eager(users, pos, neg, n_valid)
        if users is not None:
            if getattr(self, "batcher", None) is not None:
                raise RuntimeError("FusedStep.step: this graph samples its own batch; call step() without arguments")
            self._load(users, pos, neg, n_valid)
        elif getattr(self, "batcher", None) is None:
            raise RuntimeError("FusedStep.step: a batch is needed (the graph was captured without a sampler)")
        self.graph_exec.replay()
        return self.scal[1], self.scal[2], self.scal[3]
