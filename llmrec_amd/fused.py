"""Fused training step: the whole Stage-2 step - sampler included - as a fixed sequence of ~80 HIP launches
over preallocated buffers, with the backward pass written out by hand (no autograd graph), captured into one
HIP graph; likewise an evaluation (forward + scoring + masked top-K).

Same arithmetic as the modular path (Models.MM_Model.forward + llmrec_amd.engine.train_step,
i.e. reference Models.py:127-199 + main.py:228-278); what changes is the organisation:
  * the seven item-side streams (image, text, 5 attributes) travel as ONE [rows, 7d] operand, so
    the 16 side-feature SpMMs of the reference (Models.py:152-167) become 4 (+4 in backward) and the
    adjacency indices are read once per direction instead of seven times;
  * the 8 BPR + prune losses (main.py:232-254) are three launches (llmrec_bpr_multi_fwd_f32: scores, rank
    selection, reduction) and their backward one (llmrec_bpr_multi_bwd_f32), scattering straight into the
    gradient buffers;
  * projections and weight gradients of the constant feature matrices are grouped launches (8 projections in
    one; item_trans' five streams in one) in split-precision bf16x3 arithmetic (LLMREC_GEMM=f32: exact fp32);
  * independent chains run on five HIP streams (fork/join = graph edges);
  * gradients are written into preallocated .grad tensors; AdamW reads them in place;
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
        self.P_cat, self.U_cat, self.I_cat = f(I, S * d), f(U, S * d), f(I, S * d)
        self.P_usr, self.prof_i, self.prof_u = f(U, d), f(I, d), f(U, d)
        self.Ul = [f(U, d) for _ in range(self.L)]
        self.Il = [f(I, d) for _ in range(self.L)]
        self.E_u, self.E_i = f(U, d), f(I, d)
        # backward buffers
        # LLMREC_SPARSE_ZERO=1 (default): the loss backward scatters into buffers that are all-zero between steps - dE_u / dE_i and
        # the sc_* sources of the fusion backward - and clears exactly the rows it touched afterwards
        # (llmrec_bpr_multi_zero_rows_f32); the fusion backward writes d*_cat = source + its own term
        # (llmrec_fuse_bwd_src_f32). No dense memset of the six gradient buffers (70 MB per step at the Netflix shape).
        self.sparse_zero = os.environ.get("LLMREC_SPARSE_ZERO", "1") == "1"
        # LLMREC_CHECK_ZERO=1 (debug; synchronises, so not under capture): assert that invariant before every scatter
        self.check_zero = os.environ.get("LLMREC_CHECK_ZERO", "0") == "1"
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.dE_u, self.dE_i = z(U, d), z(I, d)
        if self.sparse_zero:
            self.sc_U, self.sc_I, self.sc_prof = z(U, 2 * d), z(I, S * d), z(U, d)
        self.dU_cat, self.dI_cat, self.dP_cat = f(U, S * d), f(I, S * d), f(I, S * d)
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
        # Three independent chains (7-stream side features / LLM profile / ID embeddings) run on three
        # HIP streams in forward and in backward; under capture the fork/join becomes graph edges, so
        # the latency-bound Netflix-scale SpMMs and the weight-gradient GEMMs overlap.
        import os as _os
        self.multi_stream = _os.environ.get("LLMREC_STREAMS", "1") == "1"
        self.s1, self.s2, self.s3, self.s4 = (torch.cuda.Stream(device=dev) for _ in range(4))
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
        self._bwd_accumulators = (self.dE_u, self.dE_i, self.dU_cat, self.dI_cat, self.dprof_u, self.dprof_i)
        self._zeroed = False
        self._zero_in_forward = False                         # set by step_eager: forward() alone (evaluation) must not pay for it
        # projection / weight-gradient arithmetic: "bf16x3" (default) = exact 3-term bf16 split of both operands, six bf16
        # MFMAs, fp32-roundoff-class error (2e-6 measured), HBM-bound; "f32" = the exact fp32 MFMA fma chain

        self.gemm = os.environ.get("LLMREC_GEMM", "bf16x3")
        self.wgrad_serial = os.environ.get("LLMREC_WGRAD_SERIAL", "1") == "1"
        # LLMREC_WGRAD_MULTI=1 (default): item_trans', text's and image's weight gradients as ONE launch (bf16x3 only)
        self.wgrad_multi = os.environ.get("LLMREC_WGRAD_MULTI", "1") == "1" and self.gemm == "bf16x3"
        self.id_chain_late = os.environ.get("LLMREC_ID_CHAIN_LATE", "0") == "1"
        # Launch (= capture) order at the fork points. Where a bit is set, the critical path's next launch is issued BEFORE the
        # side stream's work and the side stream waits for an event recorded at the fork point; the order decides which branch
        # the graph runs behind its parent without a cross-queue hand-off. Bits: 1 projection before the ID chain / sampler,
        # 2 the forward's side-feature SpMMs before the profile chain, 4 fuse(user) before fuse(item), 8 the BPR backward
        # before the feature regulariser / loss assembly, 16 fuse_bwd(item) before fuse_bwd(user), 32 the backward's side
        # chain before the profile / ID chains. Measured one at a time and interleaved with the baseline on one box
        # (0.6552-0.6576 ms per step): 2 -> 0.6435, 8 -> 0.6471, 2 + 8 -> 0.6297-0.6344; 1 and 32 lose 2-7 %, 4 and 16 are neutral.
        self.critical_first = int(os.environ.get("LLMREC_CRITICAL_FIRST", "10"))

    # -- raw kernel helpers -----------------------------------------------------------------------
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

    def _spmm(self, a: ops.Csr, X, Y, accumulate=False, tag=0, epilogue=None):
        """Y = epilogue(A X) (llmrec_spmm_f32); accumulate: Y += A X. tag: one partial-sum scratch per concurrent chain."""
        d = X.shape[1]
        sw, pl = a.plan_for(d, whole_row=epilogue is not None and epilogue.op != ops.EPI_NONE)
        partials = None
        if pl.n_seg:
            key = (id(pl), d, tag)
            partials = self._partials.get(key)
            if partials is None:
                partials = self._partials[key] = torch.empty(pl.n_seg * d, dtype=torch.float32, device=X.device)
        if accumulate:
            epilogue = ops.spmm_epilogue(ops.EPI_NONE, 1.0, Y)
        _call("llmrec_spmm_f32", a.n_rows, a.n_cols, _p(a.rowptr), _p(a.colidx), _p(a.val), _p(a.row_scale), _p(a.col_scale),
              _p(X), _ld(X), _p(Y), _ld(Y), d, sw, _c.byref(pl.c_struct()), _p(partials),
              _c.byref(epilogue) if epilogue is not None else None)

    def _linear(self, X, lin, out):
        _call("llmrec_linear_fwd_f32", X.shape[0], self.d, X.shape[1], _p(X), _ld(X), _p(lin.weight), _ld(lin.weight), _p(lin.bias),
              _p(out), _ld(out))

    def _project_all(self):
        """All 8 projections of Models.py:145-150 in one grouped launch (d <= 64), else one by one."""
        m = self.m
        # longest K first: the launch's tail then consists of the short (512/768-wide) work units
        jobs = [(m.item_feats[key], m.item_trans, self._side(self.P_cat, 2 + k)) for k, key in enumerate(self.keys)]
        jobs.append((m.user_feats, m.user_trans, self.P_usr))
        jobs += [(m.text_feats, m.text_trans, self._side(self.P_cat, 1)), (m.image_feats, m.image_trans, self._side(self.P_cat, 0))]
        jobs.sort(key=lambda j: -j[0].shape[1])
        if self.d > 64 or len(jobs) > _lib.CONST["LLMREC_LINEAR_MAX_PROBLEMS"]:
            for X, lin, out in jobs:
                self._linear(X, lin, out)
            return
        arr = (ops.LinearProblem * len(jobs))()
        for i, (X, lin, out) in enumerate(jobs):
            arr[i].X, arr[i].ldx, arr[i].M, arr[i].K = X.data_ptr(), _ld(X), X.shape[0], X.shape[1]
            arr[i].W, arr[i].ldw, arr[i].bias = lin.weight.data_ptr(), _ld(lin.weight), lin.bias.data_ptr()
            arr[i].Y, arr[i].ldy = out.data_ptr(), _ld(out)
        _call("llmrec_linear_fwd_grouped_bf16x3" if self.gemm == "bf16x3" else "llmrec_linear_fwd_grouped_f32", len(jobs), arr, self.d)

    def _wgrad(self, dY, X, lin, accumulate, ws=None):
        ops.linear_wgrad_grouped([(dY, X)], lin.weight.grad, lin.bias.grad, accumulate, self.ws_wgrad if ws is None else ws,
                                 precision=self.gemm)

    def _softmax(self, Z, Y):
        _call("llmrec_softmax_rows_fwd_f32", Z.shape[0], self.d, _p(Z), _ld(Z), _p(Y), _ld(Y))

    def _softmax_bwd(self, Y, dY, dZ):
        _call("llmrec_softmax_rows_bwd_f32", Y.shape[0], self.d, _p(Y), _ld(Y), _p(dY), _ld(dY), _p(dZ), _ld(dZ))

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
    def forward(self, sampler=None):
        m, d = self.m, self.d
        order = self.critical_first                                      # capture order at the fork points (see __init__)
        proj_first, side_first, fuse_user_first = order & 1, order & 2, order & 4
        if proj_first:
            ev0 = self._mark()
            self._project_all()
            self._fork_from(ev0, self.s2)
        else:
            self._fork(self.s2)
        with self._on(self.s2):                                          # ID chain: needs no projection
            if sampler is not None and self.multi_stream:                # the batch is first read by the losses, after the join below:
                sampler()                                                # sampling rides beside the projection instead of ahead of it
            if self._zero_in_forward:                                    # the backward's scatter targets, off the critical path
                if not self.sparse_zero:
                    self._zero_accumulators()
                self.opt.advance()                                       # AdamW's step counter / bias corrections, likewise
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
        if not proj_first:
            self._project_all()
        if not side_first:
            self._fork(self.s1)
        else:
            ev1 = self._mark()
            self._spmm(self.ui.fwd, self.P_cat, self.U_cat)
            self._spmm(self.iu.fwd, self.U_cat, self.I_cat)
            self._fork_from(ev1, self.s1)
        with self._on(self.s1):                                          # profile stream: items first
            self._spmm(self.iu.fwd, self.P_usr, self.prof_i, tag=1)
            self._spmm(self.ui.fwd, self.prof_i, self.prof_u, tag=1)
        if not side_first:
            self._spmm(self.ui.fwd, self.P_cat, self.U_cat)              # 7 streams, one adjacency pass
            self._spmm(self.iu.fwd, self.U_cat, self.I_cat)
        self._join(self.s1, self.s2)

        def fuse(out, base, layers, cat, prof):
            means = [base] + layers
            norms = self._norm_terms(cat, prof)
            mp, ml = self._tables(means)
            npt, nl = self._tables(norms)
            _call("llmrec_fuse_fwd_f32", out.shape[0], d, 1.0 / len(means), len(means), mp, ml, len(norms), npt, nl, self._rates(),
                  _p(out), _ld(out))
        if fuse_user_first:
            ev2 = self._mark()
            fuse(self.E_u, m.user_id_embedding.weight, self.Ul, self.U_cat, self.prof_u)
            self._fork_from(ev2, self.s3)
        else:
            self._fork(self.s3)
        with self._on(self.s3):                                          # the item table beside the user table
            fuse(self.E_i, m.item_id_embedding.weight, self.Il, self.I_cat, self.prof_i)
        if not fuse_user_first:
            fuse(self.E_u, m.user_id_embedding.weight, self.Ul, self.U_cat, self.prof_u)
        self._join(self.s3)

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
        tU, tI, tP = (self.sc_U, self.sc_I, self.sc_prof) if self.sparse_zero else (self.dU_cat, self.dI_cat, self.dprof_u)
        for s in range(2):
            tabs.append((self._side(self.U_cat, s), self._side(self.I_cat, s), self._side(tU, s), self._side(tI, s)))
        for k in range(len(self.keys)):
            tabs.append((self.prof_u, self._side(self.I_cat, 2 + k), tP, self._side(tI, 2 + k)))
        for i, (eu, ei, deu, dei) in enumerate(tabs):
            arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = eu.data_ptr(), _ld(eu), ei.data_ptr(), _ld(ei)
            arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = deu.data_ptr(), _ld(deu), dei.data_ptr(), _ld(dei)
            arr[i].g_mf, arr[i].g_emb = self.w_mf[i], self.w_emb[i]
        return arr

    def loss_backward(self, users, pos, neg, n_valid=None):
        B = users.numel()
        if B > self.b_max:
            raise RuntimeError("FusedStep: batch of %d exceeds b_max %d" % (B, self.b_max))
        probs = self._problems()
        hp = self.hp
        _call("llmrec_bpr_multi_fwd_f32", self.n_prob, probs, self.d, _p(users), _p(pos), _p(neg), B, _p(n_valid),
              float(1 - hp.prune_loss_drop_rate), float(hp.decay), float(hp.batch_size), _p(self.out), _p(self.saved))
        # feature regulariser value + loss values for logging, off the critical path: loss = sum_p w_mf[p] * mf_p + emb_0 + feat_reg
        def side():
            with self._on(self.s3):
                self._feat_reg()
                self._assemble_loss(0)
        if self.critical_first & 8:
            ev = self._mark()
            self._backward(probs, users, pos, neg, n_valid, after_first=lambda: (self._fork_from(ev, self.s3), side()))
        else:
            self._fork(self.s3)
            side()
            self._backward(probs, users, pos, neg, n_valid)
        self._join(self.s3)

    def _zero_accumulators(self):
        """The six scatter targets of the backward, cleared by ONE launch (llmrec_zero_multi_f32)."""
        arr = (ops.ZeroTensor * len(self._bwd_accumulators))()
        for i, t in enumerate(self._bwd_accumulators):
            arr[i].p, arr[i].n = t.data_ptr(), t.numel()
        _call("llmrec_zero_multi_f32", len(self._bwd_accumulators), arr)

    def _assemble_loss(self, mode: int, tail=None, inv_world: float = 1.0):
        """Logged scalars (main.py:273,280-283) from the 8 BPR results + the regulariser: one single-wave launch."""
        w = (_c.c_float * self.n_prob)(*self.w_mf)
        _call("llmrec_loss_assemble_f32", mode, self.n_prob, _p(self.out), w, _p(self.scal), _p(tail), float(inv_world))

    def _feat_reg(self):
        """Feature regulariser (main.py:151-156) over the image/text columns of both cat buffers -> scal[0]."""
        coef = self.hp.feat_reg_decay * 0.5 / self.I
        for k, blk in enumerate((self.I_cat, self.U_cat)):
            _call("llmrec_sumsq_f32", blk.shape[0], 2 * self.d, _p(blk), _ld(blk), float(coef), k, _p(self.scal), _p(self.ws_sumsq),
                  self.ws_sumsq.numel())

    def _backward(self, probs, users, pos, neg, n_valid, replicated_scale: float = 1.0, after_first=None):
        """Hand-written backward from the saved BPR state to the parameter gradients. replicated_scale
        weights the batch-independent loss terms (1 / world on batch-sharded replicas, whose
        gradients are summed over ranks afterwards)."""
        hp, d, L, S = self.hp, self.d, self.L, self.S
        B = users.numel()
        coef = hp.feat_reg_decay * 0.5 / self.I * replicated_scale
        if not self._zeroed and not self.sparse_zero:
            self._zero_accumulators()
        self._zeroed = False
        if self.sparse_zero and self.check_zero and not torch.cuda.is_current_stream_capturing():
            dirty = [n for n, t in (("dE_u", self.dE_u), ("dE_i", self.dE_i), ("sc_U", self.sc_U), ("sc_I", self.sc_I), ("sc_prof", self.sc_prof))
                     if float(t.abs().max()) != 0.0]
            if dirty:
                raise RuntimeError("FusedStep: scatter targets not all-zero before the loss backward: %s (an aborted step? call reset_scatter_targets())" % dirty)
        _call("llmrec_bpr_multi_bwd_f32", self.n_prob, probs, d, _p(users), _p(pos), _p(neg), B, _p(n_valid), float(hp.decay),
              float(hp.batch_size), _p(self.saved))
        if after_first is not None:
            after_first()
        fuse_item_first, side_chain_first = self.critical_first & 16, self.critical_first & 32
        def fuse_bwd(dout, cat, prof, dcat, dprof):
            norms, dnorms = self._norm_terms(cat, prof), self._norm_terms(dcat, dprof)
            npt, nl = self._tables(norms)
            dp, dl = self._tables(dnorms)
            # the feature regulariser's gradient on the image / text streams (terms 0, 1) rides along: 2 coef x
            if not self.sparse_zero:
                _call("llmrec_fuse_bwd_f32", dout.shape[0], d, _p(dout), _ld(dout), len(norms), npt, nl, self._rates(), dp, dl, 1,
                      2, float(2.0 * coef))
                return
            # sources = what the loss backward scattered for this side (terms in _norm_terms order: image, text, profile, attributes)
            if dcat is self.dI_cat:
                srcs = [self._side(self.sc_I, 0), self._side(self.sc_I, 1), None] + [self._side(self.sc_I, 2 + k) for k in range(len(self.keys))]
            else:
                srcs = [self._side(self.sc_U, 0), self._side(self.sc_U, 1), self.sc_prof] + [None] * len(self.keys)
            sp = (_c.c_void_p * len(srcs))(*[t.data_ptr() if t is not None else None for t in srcs])
            sl = (_c.c_int64 * len(srcs))(*[_ld(t) if t is not None else 0 for t in srcs])
            _call("llmrec_fuse_bwd_src_f32", dout.shape[0], d, _p(dout), _ld(dout), len(norms), npt, nl, self._rates(), dp, dl, sp, sl,
                  2, float(2.0 * coef))
        if fuse_item_first:                                              # the item side feeds the side chain = the critical path
            ev4 = self._mark()
            fuse_bwd(self.dE_i, self.I_cat, self.prof_i, self.dI_cat, self.dprof_i)
            self._fork_from(ev4, self.s4)
            with self._on(self.s4):
                fuse_bwd(self.dE_u, self.U_cat, self.prof_u, self.dU_cat, self.dprof_u)
        else:
            self._fork(self.s4)
            with self._on(self.s4):                                      # item side beside the user side
                fuse_bwd(self.dE_i, self.I_cat, self.prof_i, self.dI_cat, self.dprof_i)
            fuse_bwd(self.dE_u, self.U_cat, self.prof_u, self.dU_cat, self.dprof_u)
        self._join(self.s4)
        m = self.m
        inv = 1.0 / (L + 1)
        side_done = False
        if side_chain_first:                                             # the side chain's two products first, then the side streams
            ev5 = self._mark()
            self._spmm(self.iu.bwd, self.dI_cat, self.dU_cat, accumulate=True)
            self._spmm(self.ui.bwd, self.dU_cat, self.dP_cat)
            side_done = True
            self._fork_from(ev5, self.s1)
        else:
            self._fork(self.s1)
        with self._on(self.s1):
            # profile chain: prof_u = ui(prof_i), prof_i = iu(P_usr); then user_trans' weight gradient
            self._spmm(self.ui.bwd, self.dprof_u, self.dprof_i, accumulate=True, tag=1)
            self._spmm(self.iu.bwd, self.dprof_i, self.dP_usr, tag=1)
            self._wgrad(self.dP_usr, m.user_feats, m.user_trans, False, ws=self.ws_wgrad_b)

        def id_chain():
            # ID chain (items of layer l+1 from the new users; softmax on the last layer)
            # every "+ mean term" and every softmax backward below is an epilogue of the SpMM that produces the tensor:
            #   dI[L] = inv dE_i                      -> g = softmax_bwd(I_L, dI[L])            (one row kernel, no SpMM feeds it)
            #   dU[l+1] = inv dE_u + A_iu^T g         -> h = softmax_bwd(U_L, dU[l+1]) on the last layer
            #   dI[l]   = inv dE_i + A_ui^T h         -> (l > 0) feeds the next round as g; (l = 0) IS the item table's gradient
            g = self.bufI
            if L >= 1:
                self._axpy(inv, self.dE_i, self.bufI, False)                              # dI[L] = mean part
                self._softmax_bwd(self.Il[L - 1], self.bufI, self.tmpI); g = self.tmpI
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
            self._axpy(inv, self.dE_u, m.user_id_embedding.weight.grad, False)    # U^0 only enters the mean
            if self.sparse_zero:                                                  # last reader of dE_u / dE_i: clear the touched rows
                _call("llmrec_bpr_multi_zero_rows_f32", self.n_prob, probs, d, _p(users), _p(pos), _p(neg), B, _p(n_valid))

        if not self.id_chain_late:
            if side_chain_first:
                self._fork_from(ev5, self.s2)
            else:
                self._fork(self.s2)
            with self._on(self.s2):
                id_chain()
        # side chain: I_cat = iu(U_cat), U_cat = ui(P_cat); then the item-side weight gradients
        if not side_done:
            self._spmm(self.iu.bwd, self.dI_cat, self.dU_cat, accumulate=True)
            self._spmm(self.ui.bwd, self.dU_cat, self.dP_cat)
        if self.id_chain_late:
            # LLMREC_ID_CHAIN_LATE=1: the ID chain's six small launches only have to be done before AdamW; started here they
            # run beside the weight gradients (346 of a SIMD's 512 registers: an SpMM wave fits next to a weight-gradient
            # wave) instead of competing with the side chain's two products. Worth 1-2 % with three weight-gradient
            # launches back to back; with the single multi-target launch the chain then ends AFTER it (its SpMMs stretch
            # to 50-70 us beside the gradient) and the step time is the same either way, so the default starts it early.
            self._fork(self.s2)
            with self._on(self.s2):
                id_chain()
        item_pairs = [(self._side(self.dP_cat, 2 + k), m.item_feats[key]) for k, key in enumerate(self.keys)]
        if self.wgrad_multi:
            # One launch for the three item-side Linears: equal slabs over all of them, so the launch is whole rounds of
            # equal blocks and the two short gradients pay no ramp-up / ragged last round of their own (same box: three launches
            # back to back 0.661 ms per step, one launch 0.637 ms; folding user_trans' in as well - it then no longer runs
            # beside the side chain's SpMMs - is 1 % slower).
            targets = [(item_pairs, m.item_trans.weight.grad, m.item_trans.bias.grad, False),
                       ([(self._side(self.dP_cat, 1), m.text_feats)], m.text_trans.weight.grad, m.text_trans.bias.grad, False),
                       ([(self._side(self.dP_cat, 0), m.image_feats)], m.image_trans.weight.grad, m.image_trans.bias.grad, False)]
            if self.ws_wgrad_multi is None:
                need = ops.linear_wgrad_multi_workspace(targets)
                self.ws_wgrad_multi = torch.empty(max(need, 0), dtype=torch.uint8, device=self.dP_cat.device) if need >= 0 else False
            if self.ws_wgrad_multi is not False:
                ops.linear_wgrad_multi(targets, self.ws_wgrad_multi)
                self._join(self.s1, self.s2)
                return
        if self.wgrad_serial:
            # Default (LLMREC_WGRAD_SERIAL=1): item_trans', text's and image's weight gradients back to back on this stream. Each
            # fills the chip (one wave per SIMD, HBM-bound); side by side (LLMREC_WGRAD_SERIAL=0, below) they interleave their
            # blocks: every launch then lasts 2-3x longer (47 / 125 / 141 us against 17 / 19 / 130 us) while the step gains
            # 2.7 % (0.652 vs 0.670 ms) from the short launches' ramp-up and tail hiding under the long one. The serial form
            # stays the default so that a launch's duration - what the bench's roofline and a rocprof summary report - is the
            # kernel's own.
            ops.linear_wgrad_grouped(item_pairs, m.item_trans.weight.grad, m.item_trans.bias.grad, False, self.ws_wgrad, precision=self.gemm)
            self._wgrad(self._side(self.dP_cat, 1), m.text_feats, m.text_trans, False, ws=self.ws_wgrad_c)
            self._wgrad(self._side(self.dP_cat, 0), m.image_feats, m.image_trans, False, ws=self.ws_wgrad_d)
            self._join(self.s1, self.s2)
            return
        # text / image weight gradients (and their partial-slab reductions) run beside item_trans' on their own streams
        self._fork(self.s3, self.s4)
        with self._on(self.s3):
            self._wgrad(self._side(self.dP_cat, 1), m.text_feats, m.text_trans, False, ws=self.ws_wgrad_c)
        with self._on(self.s4):
            self._wgrad(self._side(self.dP_cat, 0), m.image_feats, m.image_trans, False, ws=self.ws_wgrad_d)
        # the shared item_trans receives all attribute streams in one grouped launch (features are constants: no dX)
        ops.linear_wgrad_grouped(item_pairs, m.item_trans.weight.grad, m.item_trans.bias.grad, False, self.ws_wgrad, precision=self.gemm)
        self._join(self.s1, self.s2, self.s3, self.s4)

    def _train_forward(self, sampler=None):
        """forward() of a training step: also clears the backward's scatter targets (and samples the batch) on a side stream."""
        self._zero_in_forward = True
        try:
            self.forward(sampler)
        finally:
            self._zero_in_forward = False
        self._zeroed = True

    def reset_scatter_targets(self):
        """Dense clear of the buffers the sparse-zero scheme keeps all-zero between steps (set-up, (re)capture, and after a
        step that raised between the loss backward's scatter and its row-wise clean-up)."""
        for t in (self.dE_u, self.dE_i) + ((self.sc_U, self.sc_I, self.sc_prof) if self.sparse_zero else ()):
            t.zero_()

    def step_eager(self, users, pos, neg, n_valid=None, sampler=None):
        """sampler: optional callable that fills (users, pos, neg, n_valid) on the current stream first (inside the same
        graph when captured; running it on a side stream beside the forward measured no faster)."""
        side = os.environ.get("LLMREC_SAMPLER_SIDE", "1") == "1" and self.multi_stream
        try:
            if sampler is not None and not side:
                sampler()
            self._train_forward(sampler if side else None)
            self.loss_backward(users, pos, neg, n_valid)
            self.opt.step(advanced=True)
        except Exception:
            if not torch.cuda.is_current_stream_capturing():   # the invariant of LLMREC_SPARSE_ZERO may be broken: restore it
                try:
                    self.reset_scatter_targets()
                except Exception:
                    pass
            raise
        return self.scal[1], self.scal[2], self.scal[3]

    def flush(self):
        """Nothing is deferred in the single-graph step (DataParallelStep defers its AdamW)."""

    # -- evaluation -------------------------------------------------------------------------------
    def eval_topk(self, query_users: torch.Tensor, train: Optional[ops.Csr], K: int, use_graph: bool = False):
        """Reference Trainer.test up to the ranked lists (main.py:297-303, batch_test.py:83-109): no-grad forward +
        scoring + masked top-K for the listed users -> (idx int32 [n, K], scores). With use_graph the whole
        evaluation (~40 launches) is one HIP graph per (query set, K), replayed at every epoch end."""
        if not use_graph:
            self.forward()
            return ops.score_topk(self.E_u, self.E_i, query_users, train, K)
        key = (query_users.data_ptr(), query_users.numel(), K, id(train))
        ev = self._eval_graphs.get(key)
        if ev is None:
            q = query_users.to(torch.int64).contiguous()
            n = q.numel()
            idx = torch.empty(n, K, dtype=torch.int32, device=q.device)
            sc = torch.empty(n, K, dtype=torch.float32, device=q.device)
            ws = ops.topk_workspace(n, self.I, q.device)

            def run():
                self.forward()
                _call("llmrec_score_topk_ws_f32", n, _p(q), _p(self.E_u), _ld(self.E_u), _p(self.E_i), _ld(self.E_i), self.I, self.d,
                      _p(train.rowptr) if train is not None else None, _p(train.colidx) if train is not None else None,
                      K, _p(idx), _p(sc), _p(ws), ws.numel() if ws is not None else 0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (the RCCL watchdog) may touch the runtime
                run()
            ev = self._eval_graphs[key] = (g, idx, sc, q, train, ws)     # keeps the captured operands alive
        ev[0].replay()
        return ev[1], ev[2]

    def drop_eval_graph(self, query_users: torch.Tensor):
        """Release every captured evaluation (graph, result lists, top-K workspace) of this query tensor."""
        for key in [k for k in self._eval_graphs if k[0] == query_users.data_ptr() and k[1] == query_users.numel()]:
            del self._eval_graphs[key]

    # -- HIP graph --------------------------------------------------------------------------------
    def _make_static(self):
        dev = self.E_u.device
        self.static = {"users": torch.zeros(self.b_max, dtype=torch.int64, device=dev), "pos": torch.zeros(self.b_max, dtype=torch.int64, device=dev),
                       "neg": torch.zeros(self.b_max, dtype=torch.int64, device=dev), "n_valid": torch.zeros(1, dtype=torch.int32, device=dev)}
        return self.static

    def capture(self, warm_users=None, warm_pos=None, warm_neg=None, warm_n_valid=None, batcher=None):
        """Capture one step (fixed batch capacity b_max, actual size on the device in n_valid).
        With `batcher` (engine.DeviceBatcher, capacity == b_max) the sampler is part of the graph: a
        training step is then ``step()`` with no arguments = one graph replay, nothing else on the stream."""
        st = self._make_static()
        self.batcher = batcher
        if batcher is not None:
            if batcher.capacity != self.b_max:
                raise RuntimeError("FusedStep.capture: batcher capacity %d != b_max %d" % (batcher.capacity, self.b_max))
        else:
            self._load(warm_users, warm_pos, warm_neg, warm_n_valid)

        def one_step():
            fill = (lambda: batcher.fill(st["users"], st["pos"], st["neg"], st["n_valid"])) if batcher is not None else None
            self.step_eager(st["users"], st["pos"], st["neg"], st["n_valid"], sampler=fill)
        self.reset_scatter_targets()                           # (re)capture starts from the invariant, whatever ran before
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                             # warm-up on a side stream (allocations, plan caches)
            one_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (the RCCL watchdog) may touch the runtime
            one_step()
        self.graph_exec = g

    def _load(self, users, pos, neg, n_valid):
        st, B = self.static, users.numel()
        if B > self.b_max:
            raise RuntimeError("FusedStep: batch of %d exceeds b_max %d" % (B, self.b_max))
        st["users"][:B].copy_(users); st["pos"][:B].copy_(pos); st["neg"][:B].copy_(neg)
        if n_valid is None:
            st["n_valid"].fill_(B)
        else:
            st["n_valid"].copy_(n_valid)

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
